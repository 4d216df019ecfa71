#!/usr/bin/env python3
"""Golden vectors for the deringing filter, produced by the COMPILED REFERENCE
(od_dering through ref_dering, oracle/_ref/libdaalaref.so) superblock by
superblock on two small planes.  Writes tests/golden/dering.npz."""
import ctypes
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from _libs import P, ref  # noqa: E402


def dering_input(h, w, seed):
    """Reconstruction-like int16 plane in coefficient scale: smooth ramps, hard
    edges (where ringing lives) and noise."""
    rng = np.random.RandomState(seed)
    base = np.cumsum(rng.randint(-40, 41, size=(h, w)), axis=1) + np.cumsum(rng.randint(-40, 41, size=(h, w)), axis=0)
    edges = ((np.arange(w)[None, :] // 23 + np.arange(h)[:, None] // 17) % 2) * 600
    return np.clip(base + edges + rng.randint(-30, 31, size=(h, w)), -2048, 2047).astype(np.int16)


def ref_plane(r, x, xdec, pli, dirs, bskip, thresholds, overlap):
    h, w = x.shape
    n = 64 >> xdec
    nhsb, nvsb = w // n, h // n
    y = np.zeros_like(x)
    ss = bskip.shape[1]
    for sby in range(nvsb):
        for sbx in range(nhsb):
            d = (ctypes.c_int * 64)(*dirs[sby * 8:(sby + 1) * 8, sbx * 8:(sbx + 1) * 8].ravel().tolist())
            ysb = np.zeros((n, n), np.int16)
            xp = ctypes.c_void_p(x.ctypes.data + 2 * (sby * n * w + sbx * n))
            bp = ctypes.c_void_p(bskip.ctypes.data + (sby << (4 - xdec)) * ss + (sbx << (4 - xdec)))
            r.ref_dering(P(ysb), n, xp, w, 8, 8, sbx, sby, nhsb, nvsb, xdec, d, pli, bp, ss,
                         int(thresholds[sby * nhsb + sbx]), overlap, 4)
            y[sby * n:(sby + 1) * n, sbx * n:(sbx + 1) * n] = ysb
            dirs[sby * 8:(sby + 1) * 8, sbx * 8:(sbx + 1) * 8] = np.array(list(d)).reshape(8, 8)
    return y


def main():
    r = ref()
    assert r is not None
    rng = np.random.RandomState(11)
    out = {}
    nhsb, nvsb = 3, 2
    xl = dering_input(nvsb * 64, nhsb * 64, 1)
    xc = dering_input(nvsb * 32, nhsb * 32, 2)
    bskip_l = (rng.rand(nvsb * 16, nhsb * 16 + 5) < 0.3).astype(np.uint8)
    bskip_c = (rng.rand(nvsb * 8, nhsb * 8 + 5) < 0.3).astype(np.uint8)
    thr = rng.choice([0, 7, 23, 60, 150], size=nhsb * nvsb).astype(np.int32)
    dirs = np.zeros((nvsb * 8, nhsb * 8), np.int32)
    yl = ref_plane(r, xl, 0, 0, dirs, bskip_l, thr, 1)
    yc = ref_plane(r, xc, 1, 1, dirs.copy(), bskip_c, (thr * 6 // 10).astype(np.int32), 1)
    out.update(xl=xl, xc=xc, bskip_l=bskip_l, bskip_c=bskip_c, thr=thr, dirs=dirs, yl=yl, yc=yc)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "dering.npz"), **out)
    print(hashlib.sha256(yl.tobytes()).hexdigest()[:16], hashlib.sha256(yc.tobytes()).hexdigest()[:16],
          "changed luma px:", int((yl != xl).sum()), "of", xl.size)


if __name__ == "__main__":
    main()
