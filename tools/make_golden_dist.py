#!/usr/bin/env python3
"""tests/golden/dist.npz: od_compute_dist of the REAL reference (oracle/_ref/
libdaalaref_dist.so, src/encode.c:1202-1226 reached by textual inclusion) on seeded block
pairs of every size, masking on / off, flat / HVS matrices, coded quantisers on both sides
of the interpolation range.  Pins oracle/od_oracle.c's odo_compute_dist on boxes without the
reference.  Dev-container only."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from _libs import GOLDEN, P  # noqa: E402


def main():
    r = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "libdaalaref_dist.so"))
    r.ref_compute_dist.restype = ctypes.c_double
    rng = np.random.RandomState(1202)
    d = {}
    for n in (8, 16, 32, 64):
        xs, ys, meta, out = [], [], [], []
        for _ in range(48):
            amp = rng.choice([50, 400, 3000, 20000])
            x = (rng.laplace(size=(n, n)) * amp).astype(np.int32)
            y = (x + rng.laplace(size=(n, n)) * amp * rng.choice([0.01, 0.1, 0.5])).astype(np.int32)
            flat, masking, cq = int(rng.randint(2)), int(rng.randint(2)), int(rng.choice([10, 36, 41, 47, 60]))
            xs.append(x)
            ys.append(y)
            meta.append((flat, masking, cq))
            out.append(r.ref_compute_dist(P(x), P(y), n, flat, masking, cq))
        d["x%d" % n] = np.stack(xs)
        d["y%d" % n] = np.stack(ys)
        d["meta%d" % n] = np.array(meta, np.int32)
        d["dist%d" % n] = np.array(out, np.float64)
    np.savez_compressed(os.path.join(GOLDEN, "dist.npz"), **d)
    print("ok")


if __name__ == "__main__":
    main()
