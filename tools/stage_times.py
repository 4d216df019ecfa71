#!/usr/bin/env python3
"""Filter + DCT stages of the bench step timed alone (odhip_pipe_time_stage), with a digest of every
reconstructed plane so that kernel variants selected by environment knobs (ODHIP_INVERSE_OLD,
ODHIP_INVERSE_SEG, ODHIP_PYR_VARIANT ...) can be compared for speed AND equality in one GPU call.

    python tools/stage_times.py [--frames 16] [--n 10] [--tag name]
"""
import argparse
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as B  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--n", type=int, default=10)
    ap.add_argument("--tag", default="")
    ap.add_argument("--chroma-noref", action="store_true")
    args = ap.parse_args()
    import daala_amd as D
    D.init(0)
    qt = D.QuantTables.load()
    luma, chroma = B.synth_pictures(args.frames, 1234)
    pipe = D.Pipe(qt, args.frames, B.PIC_W, B.PIC_H, chroma_cfl=not args.chroma_noref, device=0, price=True)
    pipe.set_pictures(luma, chroma)
    for _ in range(3):
        pipe.step()
    pipe.flush()
    pipe.sync()
    ab = B.algorithmic_bytes(args.frames)
    out = {"tag": args.tag, "frames": args.frames, "stages": {}}
    tot_b = 0
    tot_ms = 0.0
    for st in ("image_copy_pad_luma", "image_copy_pad_chroma", "forward_pyramid_luma", "forward_pyramid_chroma",
               "dequant_inverse_luma", "dequant_inverse_chroma"):
        ms = min(pipe.time_stage(st, args.n) for _ in range(3))
        by = ab[st]
        out["stages"][st] = {"ms": round(ms, 4), "GBs": round(by / ms / 1e6, 1), "frac": round(by / ms / 1e6 / B.HBM_PEAK_GBS, 4)}
        tot_b += by
        tot_ms += ms
    out["whole_stage"] = {"ms": round(tot_ms, 4), "GBs": round(tot_b / tot_ms / 1e6, 1),
                          "frac": round(tot_b / tot_ms / 1e6 / B.HBM_PEAK_GBS, 4)}
    h = hashlib.sha256()
    for set_ in (0, 1):
        for bs in range(5 - set_):
            h.update(pipe.read(D.BUF_RECON, set_, bs).tobytes())
    out["recon_digest"] = h.hexdigest()[:16]
    hl = hashlib.sha256()
    for set_ in (0, 1):
        for bs in range(5 - set_):
            hl.update(pipe.read(D.BUF_LEVEL, set_, bs).tobytes())
    out["levels_digest"] = hl.hexdigest()[:16]
    print(json.dumps(out))
    pipe.destroy()


if __name__ == "__main__":
    main()
