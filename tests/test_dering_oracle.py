"""CPU: the oracle's deringing filter (src/dering.c restated) against golden
vectors produced by the compiled reference and, when oracle/_ref is present,
the reference itself superblock by superblock."""
import ctypes
import os

import numpy as np

from _libs import GOLDEN, P, oracle, ref


def _load():
    return np.load(os.path.join(GOLDEN, "dering.npz"))


def test_dering_plane_matches_golden():
    g = _load()
    o = oracle()
    dirs = np.zeros_like(g["dirs"])
    xl = np.ascontiguousarray(g["xl"])
    yl = np.zeros_like(xl)
    bl = np.ascontiguousarray(g["bskip_l"])
    thr = np.ascontiguousarray(g["thr"])
    o.odo_dering_plane(P(yl), P(xl), xl.shape[1], 3, 2, 0, P(dirs), 0, P(bl), bl.shape[1], P(thr), 1, 4)
    assert np.array_equal(yl, g["yl"])
    assert np.array_equal(dirs, g["dirs"])
    xc = np.ascontiguousarray(g["xc"])
    yc = np.zeros_like(xc)
    bc = np.ascontiguousarray(g["bskip_c"])
    thrc = (thr * 6 // 10).astype(np.int32)
    o.odo_dering_plane(P(yc), P(xc), xc.shape[1], 3, 2, 1, P(dirs), 1, P(bc), bc.shape[1], P(thrc), 1, 4)
    assert np.array_equal(yc, g["yc"])
    # a threshold of 0 leaves the plane unchanged (what "level 0" means)
    zero = np.zeros(6, np.int32)
    o.odo_dering_plane(P(yl), P(xl), xl.shape[1], 3, 2, 0, P(dirs), 0, P(bl), bl.shape[1], P(zero), 1, 4)
    assert np.array_equal(yl, xl)


def test_dering_direction_known_answers():
    """od_dir_find8 on pure gratings: the direction index follows the pattern
    (0 = 45 degrees up-right, 2 = horizontal, 4 = 45 degrees down-right, 6 = vertical)."""
    o = oracle()
    ii, jj = np.meshgrid(np.arange(8), np.arange(8), indexing="ij")
    cases = {2: ii, 6: jj, 0: ii + jj, 4: ii - jj}
    for want, phase in cases.items():
        img = (np.sin(phase * 1.3) * 100).astype(np.int16) << 4  # 8-bit pixel scale: the int32 costs do not overflow
        var = ctypes.c_int32()
        got = o.odo_dir_find8(P(np.ascontiguousarray(img)), 8, ctypes.byref(var), 4)
        assert got == want, (want, got)
        assert var.value > 0
    flat = np.full((8, 8), 77 << 4, np.int16)
    var = ctypes.c_int32()
    assert o.odo_dir_find8(P(flat), 8, ctypes.byref(var), 4) == 0 and var.value == 0


def test_dering_matches_reference_live():
    r = ref()
    if r is None:
        return
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from make_golden_dering import dering_input
    o = oracle()
    rng = np.random.RandomState(21)
    n_calls = 0
    for xdec, pli in ((0, 0), (1, 1), (1, 2)):
        nhsb, nvsb = 2, 3
        n = 64 >> xdec
        x = dering_input(nvsb * n, nhsb * n, 30 + xdec)
        ss = nhsb * 16 + 1
        bskip = (rng.rand(nvsb * 16, ss) < 0.4).astype(np.uint8)
        for thr in (3, 40, 333, 5000):
            for overlap in (0, 1):
                for sby in range(nvsb):
                    for sbx in range(nhsb):
                        d1 = (ctypes.c_int * 64)(*rng.randint(0, 8, size=64).tolist())
                        d2 = (ctypes.c_int * 64)(*list(d1))
                        y1 = np.zeros((n, n), np.int16)
                        y2 = np.zeros((n, n), np.int16)
                        xp = ctypes.c_void_p(x.ctypes.data + 2 * (sby * n * x.shape[1] + sbx * n))
                        bp = ctypes.c_void_p(bskip.ctypes.data + (sby << (4 - xdec)) * ss + (sbx << (4 - xdec)))
                        o.odo_dering(P(y1), n, xp, x.shape[1], 8, 8, sbx, sby, nhsb, nvsb, xdec, d1, pli, bp,
                                     ss, thr, overlap, 4)
                        r.ref_dering(P(y2), n, xp, x.shape[1], 8, 8, sbx, sby, nhsb, nvsb, xdec, d2, pli, bp,
                                     ss, thr, overlap, 4)
                        assert np.array_equal(y1, y2), (xdec, thr, overlap, sbx, sby)
                        assert list(d1) == list(d2)
                        n_calls += 1
    assert n_calls == 144
