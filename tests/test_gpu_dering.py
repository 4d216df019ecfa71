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


def _ref_or_oracle_plane(x, xdec, pli, dirs, bskip, thr, overlap=1):
    """One plane, one threshold set, through the COMPILED REFERENCE's od_dering when
    oracle/_ref is here (superblock by superblock), through the oracle otherwise."""
    from _libs import ref
    h, w = x.shape
    n = 64 >> xdec
    nhsb, nvsb = w // n, h // n
    if ref() is not None:
        from make_golden_dering import ref_plane
        return ref_plane(ref(), x, xdec, pli, dirs, bskip, thr, overlap)
    want = np.zeros((h, w), np.int16)
    oracle().odo_dering_plane(P(want), P(x), w, nhsb, nvsb, xdec, P(dirs), pli, P(bskip), bskip.shape[1],
                              P(np.ascontiguousarray(thr)), overlap, 4)
    return want


@pytest.mark.parametrize("xdec,pli", [(0, 0), (1, 1)])
def test_dering_whole_int16_domain(hip, xdec, pli):
    """The packed int16 path must keep the reference's casts for EVERY input: samples
    over the whole int16 range (differences wrap, |p| reaches 32768), thresholds 0, tiny,
    32767 and beyond it / negative (the per-sample path)."""
    rng = np.random.RandomState(77 + xdec)
    nvsb, nhsb = 2, 3
    n = 64 >> xdec
    h, w = nvsb * n, nhsb * n
    x = rng.randint(-32768, 32768, size=(2, h, w)).astype(np.int16)
    x[0, ::7, ::5] = -32768
    x[0, 1::7, 2::5] = 32767
    x[1] = (rng.randint(-3, 4, size=(h, w)) + np.where(rng.rand(h, w) < 0.02, 30000, 0)).astype(np.int16)
    ss = nhsb * (16 >> xdec) + 1
    bskip = (rng.rand(2, nvsb * (16 >> xdec), ss) < 0.2).astype(np.uint8)
    cands = [0, 1, 2, 3, 5, 300, 10922, 32767, 32768, 40000, 70000, 200000, -5]
    thr = np.array([[np.roll(cands, c + 3 * p)[:nvsb * nhsb] for c in range(len(cands))]
                    for p in range(2)], np.int32)
    dirs_in = rng.randint(0, 8, size=(2, nvsb * 8, nhsb * 8)).astype(np.int32)
    tdirs = _cuda(dirs_in)
    y = hip.dering_planes(_cuda(x), xdec, tdirs, pli, _cuda(bskip), _cuda(thr)).cpu().numpy()
    for p in range(2):
        for c in range(len(cands)):
            dirs = dirs_in[p].copy()
            want = _ref_or_oracle_plane(x[p], xdec, pli, dirs, bskip[p], thr[p, c])
            assert np.array_equal(y[p, c], want), (p, c, thr[p, c])


def test_dering_unaligned_planes(hip):
    """A row stride that is not a multiple of 8 samples and a plane that does not start on
    16 bytes take the scalar load / store path: same results."""
    import torch
    from make_golden_dering import dering_input
    L = hip.lib()
    rng = np.random.RandomState(5)
    for xdec, pli in ((0, 0), (1, 2)):
        nvsb, nhsb = 2, 2
        n = 64 >> xdec
        h, w = nvsb * n, nhsb * n
        stride = w + 4
        x = np.zeros((h, stride), np.int16)
        x[:, :w] = dering_input(h, w, 9 + xdec)
        ss = nhsb * (16 >> xdec)
        bskip = (rng.rand(nvsb * (16 >> xdec), ss) < 0.3).astype(np.uint8)
        thr = rng.choice([7, 90, 700], size=(1, 2, nvsb * nhsb)).astype(np.int32)
        dirs_in = rng.randint(0, 8, size=(nvsb * 8, nhsb * 8)).astype(np.int32)
        pad = 3                                     # samples: the planes start 6 bytes into the tensors
        tx = torch.zeros(h * stride + 8, dtype=torch.int16, device="cuda")
        tx[pad:pad + h * stride] = _cuda(x).flatten()
        ty = torch.zeros(2 * h * stride + 8, dtype=torch.int16, device="cuda")
        tdirs, tskip, tthr = _cuda(dirs_in), _cuda(bskip), _cuda(thr)
        rc = L.odhip_dering_planes(ctypes.c_void_p(ty.data_ptr() + 2 * pad), ctypes.c_void_p(tx.data_ptr() + 2 * pad),
                                   stride, nhsb, nvsb, xdec, 1, ctypes.c_void_p(tdirs.data_ptr()), pli,
                                   ctypes.c_void_p(tskip.data_ptr()), ss, ctypes.c_long(bskip.size),
                                   ctypes.c_void_p(tthr.data_ptr()), 2, 1, 4, None)
        assert rc == 0
        torch.cuda.synchronize()
        y = ty[pad:pad + 2 * h * stride].cpu().numpy().reshape(2, h, stride)[:, :, :w]
        for c in range(2):
            dirs = dirs_in.copy()
            want = np.zeros((h, w), np.int16)
            oracle().odo_dering_plane(P(want), P(np.ascontiguousarray(x[:, :w])), w, nhsb, nvsb, xdec, P(dirs), pli,
                                      P(bskip), ss, P(np.ascontiguousarray(thr[0, c])), 1, 4)
            assert np.array_equal(y[c], want), (xdec, c)


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
