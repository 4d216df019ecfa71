"""od_compute_dist (src/encode.c:1082-1226): the oracle restatement
against golden values from the compiled reference and, when it is present, live; the GPU
split (device parts + host pow) against the oracle, bit for bit."""
import ctypes
import os

import numpy as np
import pytest

from _libs import GOLDEN, P, ROOT, oracle

DIST_SO = os.path.join(ROOT, "oracle", "_ref", "libdaalaref_dist.so")


def _o():
    o = oracle()
    o.odo_compute_dist.restype = ctypes.c_double
    return o


def _bits(a):
    return np.asarray(a, np.float64).view(np.int64)


def test_oracle_dist_matches_reference_golden():
    g = np.load(os.path.join(GOLDEN, "dist.npz"))
    o = _o()
    for n in (8, 16, 32, 64):
        for x, y, (flat, masking, cq), want in zip(g["x%d" % n], g["y%d" % n], g["meta%d" % n],
                                                    g["dist%d" % n]):
            x, y = np.ascontiguousarray(x), np.ascontiguousarray(y)
            got = o.odo_compute_dist(P(x), P(y), n, int(flat), int(masking), int(cq))
            assert _bits(got) == _bits(want), (n, flat, masking, cq)
        assert len(set(np.round(g["dist%d" % n], 3))) > 40


@pytest.mark.skipif(not os.path.exists(DIST_SO), reason="oracle/_ref not built here")
def test_oracle_dist_matches_reference_live():
    r = ctypes.CDLL(DIST_SO)
    r.ref_compute_dist.restype = ctypes.c_double
    o = _o()
    rng = np.random.RandomState(5)
    for n in (8, 16, 32, 64):
        for _ in range(60):
            amp = rng.choice([20, 300, 5000, 60000])
            x = (rng.laplace(size=(n, n)) * amp).astype(np.int32)
            y = (x * rng.choice([0, 1]) + rng.laplace(size=(n, n)) * amp * rng.choice([0.02, 0.3, 2])) \
                .astype(np.int32)
            for flat in (0, 1):
                masking, cq = int(rng.randint(2)), int(rng.randint(1, 64))
                a = o.odo_compute_dist(P(x), P(y), n, flat, masking, cq)
                b = r.ref_compute_dist(P(x), P(y), n, flat, masking, cq)
                assert _bits(a) == _bits(b)


@pytest.mark.gpu
@pytest.mark.parametrize("bs", [1, 2, 3, 4])
def test_gpu_dist_matches_oracle(bs):
    import torch
    import daala_amd as D
    D.init(0)
    o = _o()
    n = 4 << bs
    rng = np.random.RandomState(40 + bs)
    nplanes, h, w = 2, 192, 320            # not multiples of the 64-wide tile for n = 64: 192 = 3 x 64 ok
    if n == 64:
        h, w = 128, 192
    amp = np.kron(rng.choice([30, 300, 4000], size=(nplanes, h // 8, w // 8)), np.ones((8, 8)))
    x = (rng.laplace(size=(nplanes, h, w)) * amp).astype(np.int32)
    y = (x + rng.laplace(size=(nplanes, h, w)) * amp * 0.2).astype(np.int32)
    y[0, :16] = x[0, :16]                  # identical blocks: zero error
    tx, ty = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    for (masking, flat, cq) in ((1, 0, 41), (0, 0, 30), (1, 1, 50)):
        got, _ = D.compute_dist(tx, ty, bs, masking, flat, cq)
        for p in range(nplanes):
            for by in range(h // n):
                for bx in range(w // n):
                    xb = np.ascontiguousarray(x[p, by * n:(by + 1) * n, bx * n:(bx + 1) * n])
                    yb = np.ascontiguousarray(y[p, by * n:(by + 1) * n, bx * n:(bx + 1) * n])
                    want = o.odo_compute_dist(P(xb), P(yb), n, flat, masking, cq)
                    assert _bits(got[p, by, bx]) == _bits(want), (bs, masking, flat, p, by, bx)
