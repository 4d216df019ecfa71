#!/usr/bin/env python3
"""Generate tests/golden/cfl_flip.npz from the REAL reference (oracle/_ref):
keyframe chroma blocks through od_pvq_encode (src/pvq_encoder.c:789-979) on a
fresh encoder context; the fixture keeps the inputs and whether the reference
negated its `ref` argument (the chroma-from-luma sign flip, :846-872).
Dev-container only; deterministic."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from _libs import GOLDEN, P, ref  # noqa: E402
from daala_amd.quant import QuantTables  # noqa: E402


def blocks(rng, bs, count):
    n = 4 << bs
    amp = rng.choice([30, 300, 3000], size=(count, 1))
    x = (rng.laplace(size=(count, n * n)) * amp).astype(np.int32)
    mix = rng.choice([0.3, 1, 5], size=(count, 1))
    scale = rng.choice([0.01, 0.2, 1.0], size=(count, 1)) * rng.choice([1, -1], size=(count, 1))
    r = (x * scale + rng.laplace(size=(count, n * n)) * amp * mix).astype(np.int32)
    return x, r


def main():
    r = ref()
    assert r is not None, "build oracle/_ref first: make -C oracle ref"
    qt = QuantTables.load()
    rng = np.random.RandomState(846)
    d = {}
    for bs in range(4):
        qm, qmi = qt.qm_slices(1, bs)
        beta = np.array(list(qt.beta_band(1, bs)) + [4096] * 12, np.int32)[:12]
        x, rr = blocks(rng, bs, 64)
        flips = np.zeros(len(x), np.int32)
        for i in range(len(x)):
            r1 = rr[i].copy()
            out = np.zeros_like(r1)
            r.ref_pvq_encode_block(P(r1), P(x[i]), P(out), 37, 1, bs, P(beta), 1, P(qm), P(qmi), 1)
            if np.array_equal(r1, rr[i]):
                flips[i] = 0
            else:
                off0 = 1
                assert np.array_equal(r1[off0:min(len(r1), 512)], -rr[i][off0:min(len(r1), 512)])
                flips[i] = 1
        d["x%d" % bs], d["r%d" % bs], d["flip%d" % bs] = x, rr, flips
    np.savez_compressed(os.path.join(GOLDEN, "cfl_flip.npz"), **d)
    print({k: (v.shape, int(v.sum()) if k.startswith("flip") else None) for k, v in d.items()})


if __name__ == "__main__":
    main()
