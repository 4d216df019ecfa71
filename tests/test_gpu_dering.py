"""GPU parity of the deringing filter (odhip_dering_planes, od_dering_hip)
against the CPU oracle, bit-exact."""
import ctypes
import os
import sys

import numpy as np
import pytest

from _libs import P, oracle

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def hip():
    import torch
    import daala_amd
    assert torch.cuda.is_available()
    daala_amd.init(0)
    return daala_amd


def _cuda(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("xdec,pli,shape", [(0, 0, (3, 5)), (1, 1, (2, 3)), (0, 0, (1, 1)), (1, 2, (1, 4))])
def test_dering_planes_match_oracle(hip, xdec, pli, shape):
    """Every superblock, several candidate thresholds and planes in one launch."""
    from make_golden_dering import dering_input
    o = oracle()
    rng = np.random.RandomState(8 + xdec)
    nvsb, nhsb = shape
    n = 64 >> xdec
    nplanes, ncand = 3, 4
    h, w = nvsb * n, nhsb * n
    x = np.stack([dering_input(h, w, 50 + p) for p in range(nplanes)])
    ss = nhsb * (16 >> xdec) + 3
    rows = nvsb * (16 >> xdec)
    bskip = (rng.rand(nplanes, rows, ss) < 0.35).astype(np.uint8)
    thr = rng.choice([0, 4, 19, 77, 300, 2500], size=(nplanes, ncand, nvsb * nhsb)).astype(np.int32)
    dirs_in = rng.randint(0, 8, size=(nplanes, nvsb * 8, nhsb * 8)).astype(np.int32)
    tdirs = _cuda(dirs_in)
    y = hip.dering_planes(_cuda(x), xdec, tdirs, pli, _cuda(bskip), _cuda(thr))
    y = y.cpu().numpy()
    dirs_out = tdirs.cpu().numpy()
    for p in range(nplanes):
        for c in range(ncand):
            dirs = dirs_in[p].copy()
            want = np.zeros((h, w), np.int16)
            o.odo_dering_plane(P(want), P(x[p]), w, nhsb, nvsb, xdec, P(dirs), pli, P(bskip[p]), ss,
                               P(np.ascontiguousarray(thr[p, c])), 1, 4)
            assert np.array_equal(y[p, c], want), (p, c)
            assert np.array_equal(dirs_out[p], dirs), p
    if pli == 0:
        assert not np.array_equal(dirs_out, dirs_in)   # luma writes the directions
    else:
        assert np.array_equal(dirs_out, dirs_in)       # chroma only reads them


def test_dering_per_call_surface_matches_oracle(hip):
    """od_dering_hip: od_dering's argument list, host pointers, one superblock."""
    from make_golden_dering import dering_input
    o = oracle()
    L = hip.lib()
    rng = np.random.RandomState(3)
    for xdec, pli in ((0, 0), (1, 1)):
        nhsb, nvsb = 3, 3
        n = 64 >> xdec
        x = dering_input(nvsb * n, nhsb * n, 70 + xdec)
        ss = nhsb * 16 + 2
        bskip = (rng.rand(nvsb * 16, ss) < 0.3).astype(np.uint8)
        for (sbx, sby) in ((0, 0), (1, 1), (2, 2), (2, 0)):
            for overlap in (0, 1):
                thr = int(rng.choice([9, 55, 400]))
                d1 = (ctypes.c_int * 64)(*rng.randint(0, 8, size=64).tolist())
                d2 = (ctypes.c_int * 64)(*list(d1))
                y1 = np.zeros((n, n), np.int16)
                y2 = np.zeros((n, n), np.int16)
                xp = ctypes.c_void_p(x.ctypes.data + 2 * (sby * n * x.shape[1] + sbx * n))
                bp = ctypes.c_void_p(bskip.ctypes.data + (sby << (4 - xdec)) * ss + (sbx << (4 - xdec)))
                o.odo_dering(P(y1), n, xp, x.shape[1], 8, 8, sbx, sby, nhsb, nvsb, xdec, d1, pli, bp, ss, thr,
                             overlap, 4)
                L.od_dering_hip(P(y2), n, xp, x.shape[1], 8, 8, sbx, sby, nhsb, nvsb, xdec, d2, pli, bp, ss,
                                thr, overlap, 4)
                assert np.array_equal(y1, y2), (xdec, sbx, sby, overlap)
                assert list(d1) == list(d2)


def test_dering_1080p_properties(hip):
    """Full-size: a 1080p luma + chroma frame set, five candidate levels; a zero
    threshold is the identity, all-skipped superblocks are untouched, and
    running twice gives identical output."""
    import torch
    from make_golden_dering import dering_input
    nhsb, nvsb = 30, 17
    x = _cuda(np.stack([dering_input(nvsb * 64, nhsb * 64, 5), dering_input(nvsb * 64, nhsb * 64, 6)]))
    g = torch.Generator(device="cuda").manual_seed(1)
    bskip = (torch.rand((2, nvsb * 16, nhsb * 16), device="cuda", generator=g) < 0.3).to(torch.uint8)
    bskip[:, :16, :16] = 1      # superblock (0, 0) and the flags around it: all skipped
    bskip[:, :17, :17] = 1
    base = 37
    thr = torch.tensor([int(gain * base) for gain in (0, 0.5, 0.707, 1, 1.41, 2)], dtype=torch.int32, device="cuda")
    thr = thr[None, :, None].expand(2, 6, nhsb * nvsb).contiguous()
    dirs = torch.zeros((2, nvsb * 8, nhsb * 8), dtype=torch.int32, device="cuda")
    y = hip.dering_planes(x, 0, dirs, 0, bskip, thr)
    y2 = hip.dering_planes(x, 0, dirs, 0, bskip, thr)
    assert torch.equal(y, y2)
    assert torch.equal(y[:, 0], x)                              # level 0
    assert torch.equal(y[:, 5, :64, :64], x[:, :64, :64])       # skipped superblock
    assert not torch.equal(y[:, 5], x)
    d = (y[:, 5].int() - x.int()).abs()
    assert int(d.max()) <= 6 * 2 * base * 3                      # bounded by the taps times the threshold
    assert int(dirs.min()) >= 0 and int(dirs.max()) <= 7
