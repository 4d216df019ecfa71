#!/usr/bin/env python3
"""Generate tests/golden/image_pad.npz from the REAL reference (oracle/_ref): the
padded input planes the encoder keeps (daala_image_copy_pad ->
od_img_plane_copy_pad, src/encode.c:752-837) for a few picture sizes.
Dev-container only; deterministic."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from _libs import GOLDEN, P, ref  # noqa: E402

SIZES = [(70, 50), (33, 17), (65, 129), (64, 64), (176, 120)]


def main():
    r = ref()
    assert r is not None, "build oracle/_ref first: make -C oracle ref"
    rng = np.random.RandomState(752)
    d = {}
    for i, (w, h) in enumerate(SIZES):
        cw, ch = (w + 1) // 2, (h + 1) // 2
        fr = rng.randint(0, 256, size=w * h + 2 * cw * ch).astype(np.uint8)
        fw, fh = (w + 63) // 64 * 64, (h + 63) // 64 * 64
        outs = [np.zeros((fh >> s, fw >> s), np.uint8) for s in (0, 1, 1)]
        arr = (ctypes.c_void_p * 3)(*[a.ctypes.data for a in outs])
        dims = (ctypes.c_int * 6)()
        assert r.ref_image_copy_pad(P(fr), w, h, arr, dims) == 0
        d["frame%d" % i] = fr
        d["size%d" % i] = np.array([w, h], np.int32)
        for pli in range(3):
            d["pad%d_%d" % (i, pli)] = outs[pli]
    np.savez_compressed(os.path.join(GOLDEN, "image_pad.npz"), **d)
    print(sorted(d)[:6], "...")


if __name__ == "__main__":
    main()
