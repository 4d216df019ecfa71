#!/usr/bin/env python3
"""The priced 16-frame step at other operating points, timed only (no verification): ms per step for -v values and
content types.  usage: tools/quality_time.py [-v 5,10,20] [--content checker,natural] [--steps 8]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as B  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("-v", default="5,10,20")
    ap.add_argument("--content", default="checker")
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--frames", type=int, default=16)
    args = ap.parse_args()
    import torch
    import daala_amd as D
    D.init(0)
    for content in args.content.split(","):
        B.GENERATOR = B.CONTENT[content]
        luma, chroma = B.synth_pictures(args.frames, 1234)
        for q in [int(x) for x in args.v.split(",")]:
            qt = D.QuantTables.for_quality(q) if q != 20 else D.QuantTables.load()
            pipe = D.Pipe(qt, args.frames, B.PIC_W, B.PIC_H, chroma_cfl=True, device=0, price=True)
            pipe.set_pictures(luma, chroma)
            for _ in range(2):
                pipe.step()
            pipe.flush()
            pipe.sync()
            pipe.record(True)
            t0 = time.perf_counter()
            for _ in range(args.steps):
                pipe.step()
            pipe.flush()
            pipe.sync()
            dt = (time.perf_counter() - t0) / args.steps
            kms = pipe.timings()
            print("%s -v %d: %.3f ms per step  bands luma %.3f chroma %.3f  reruns %d %d" % (
                content, q, dt * 1e3, kms["pvq_noref_bands"][0], kms["pvq_ref_bands"][0], pipe.theta_reruns(),
                pipe.price_reruns()), flush=True)
            pipe.destroy()


if __name__ == "__main__":
    main()
