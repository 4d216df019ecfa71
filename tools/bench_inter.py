#!/usr/bin/env python3
"""Auxiliary line (not bench.py's metric): the frame-batch step for INTER frames
(odhip_pipe_config.inter) - 16 pictures of 1080p, every plane through the with-reference
band stage against the pyramid of its prediction picture, the choice priced on the device -
timed like bench.py times the keyframe step, with frame 0 as the timed pipe reconstructed it
verified against the reference's C functions run the same way, and their single-core rate.

    python tools/bench_inter.py [--frames 16] [--steps 10] [--warmup 2]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402


def prediction(planes, seed):
    """A plausible motion-compensated prediction: the picture one sample to the right, with a
    little noise."""
    rng = np.random.RandomState(seed)
    return [np.clip(np.roll(p.astype(np.int32), 1, axis=1) + rng.randint(-6, 7, size=p.shape), 0, 255).astype(np.uint8)
            for p in planes]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    args = ap.parse_args()
    import torch
    import daala_amd as D
    import _pipeline_check as C
    from _libs import ref
    D.init(0)
    qt = D.QuantTables.load()
    F = args.frames
    cur = [bench.picture_planes(bench.natural_like_frame_np(i, 1234)) for i in range(F)]
    pred = [prediction(c, 50 + i) for i, c in enumerate(cur)]

    def stack(fr):
        return (np.stack([f[0] for f in fr]), np.concatenate([np.stack([f[1] for f in fr]), np.stack([f[2] for f in fr])]))

    pipe = D.Pipe(qt, F, bench.PIC_W, bench.PIC_H, price=True, inter=True)
    pipe.set_pictures(*stack(cur))
    pipe.set_reference_pictures(*stack(pred))
    for _ in range(args.warmup):
        pipe.step()
    pipe.flush()
    pipe.sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pipe.step()
    pipe.flush()
    pipe.sync()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    bpf = bench.blocks_per_frame()
    line = {"metric": "1080p INTER-frame transform blocks/s (filter+DCT+PVQ with reference), auxiliary",
            "value": F * args.steps * bpf / dt, "unit": "blocks/s", "ms_per_step": dt / args.steps * 1e3,
            "frames_per_step": F, "steps": args.steps, "content": "natural-like pictures, prediction = the picture "
            "shifted by one sample + noise", "theta_margin_reruns": pipe.theta_reruns(),
            "price_margin_reruns": pipe.price_reruns()}
    if ref() is not None:
        H, W = bench.H, bench.W
        gpu = [[pipe.read(D.BUF_RECON, 0, bs).reshape(F, H, W) for bs in range(5)],
               [pipe.read(D.BUF_RECON, 1, bs).reshape(2 * F, H // 2, W // 2) for bs in range(4)]]
        cpu, blocks, busy = C.cpu_frame(qt, cur[0], bench.PIC_W, bench.PIC_H, inter_pred=pred[0])
        line["verified"] = C.compare_frame(gpu, cpu, frame=0, frames=F) == []
        line["cpu_baseline"] = {"value": blocks / busy, "unit": "blocks/s", "cores": 1, "kind": "reference",
                                "sample": "one picture, %.1f s" % busy}
        line["speedup_vs_cpu_baseline"] = line["value"] / line["cpu_baseline"]["value"]
    pipe.destroy()
    print(json.dumps(line))


if __name__ == "__main__":
    main()
