#!/usr/bin/env python3
"""Golden vectors for the 8/16/32-point lapping filters, produced by the
COMPILED REFERENCE (oracle/_ref/libdaalaref.so: od_pre_filterN / od_post_filterN
through ref_pre_filter / ref_post_filter).  Writes tests/golden/filters.npz."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from _libs import P, ref  # noqa: E402


def main():
    r = ref()
    assert r is not None, "build oracle/_ref first (make -C oracle all)"
    rng = np.random.RandomState(77)
    d = {}
    for f in (1, 2, 3):
        n = 4 << f
        x = np.concatenate([rng.randint(-a, a + 1, size=(64, n)) for a in (2, 255, 4096, 1 << 19)])
        x = x.astype(np.int32)
        pre = np.zeros_like(x)
        post = np.zeros_like(x)
        for i in range(len(x)):
            r.ref_pre_filter(f, P(pre[i]), P(x[i]))
            r.ref_post_filter(f, P(post[i]), P(x[i]))
        d["x%d" % n] = x
        d["pre%d" % n] = pre
        d["post%d" % n] = post
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "filters.npz"), **d)
    print({k: v.shape for k, v in d.items()})


if __name__ == "__main__":
    main()
