"""CPU: pin the oracle (the checker of the HIP kernels) to the reference.

* against the committed golden vectors under tests/golden/ (generated from the
  compiled reference by tools/make_golden.py) - always;
* against the compiled reference itself (oracle/_ref/libdaalaref.so) on fresh
  random inputs - whenever that library is present (the dev container; it also
  ships to the GPU box as a prebuilt file).
"""
import ctypes
import hashlib
import os

import numpy as np
import pytest

from _libs import GOLDEN, P, oracle, ref, synth_frame

cd = ctypes.c_double


def load(name):
    return np.load(os.path.join(GOLDEN, name))


def sha(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), np.uint8)


def test_dct_golden():
    g = load("dct.npz")
    o = oracle()
    for ln in range(5):
        n = 4 << ln
        x, y = g["x%d" % n], g["y%d" % n]
        got = np.zeros_like(x)
        o.odo_fdct_2d_batch(ln, P(got), P(x), ctypes.c_long(len(x)))
        assert np.array_equal(got, y)
        back = np.zeros_like(x)
        o.odo_idct_2d_batch(ln, P(back), P(y), ctypes.c_long(len(x)))
        assert np.array_equal(back, x), "exact invertibility (reference dct.c:8302-8305)"
        v, w = g["v%d" % n], g["o%d" % n]
        for i in range(len(v)):
            out = np.zeros(n, np.int32)
            o.odo_fdct_1d(ln, P(out), P(v[i]), 1)
            assert np.array_equal(out, w[i])
            inv = np.zeros(n, np.int32)
            o.odo_idct_1d(ln, P(inv), 1, P(out))
            assert np.array_equal(inv, v[i])


def test_filter4_golden_and_known_answers():
    g = load("filter4.npz")
    o = oracle()
    for i in range(len(g["x"])):
        a = np.zeros(4, np.int32)
        o.odo_pre_filter4(P(a), P(g["x"][i]))
        assert np.array_equal(a, g["pre"][i])
        b = np.zeros(4, np.int32)
        o.odo_post_filter4(P(b), P(g["x"][i]))
        assert np.array_equal(b, g["post"][i])
        c = np.zeros(4, np.int32)
        o.odo_post_filter4(P(c), P(a))
        assert np.array_equal(c, g["x"][i]), "pre -> post is the identity"
    # the reference's own exhaustive +-676 test (src/filter.c:1645-1695)
    ys = np.zeros((16, 4), np.int32)
    for i in range(16):
        o.odo_pre_filter4(P(ys[i]), P(g["sweep"][i]))
    assert np.array_equal(ys, g["sweep_pre"])
    assert ys.min(0).tolist() == [-1003, -1198, -1198, -1003]
    assert ys.max(0).tolist() == [1003, 1198, 1198, 1003]


def test_filters_8_16_32_golden_and_reference():
    """od_pre_filter8/16/32, od_post_filter8/16/32 (unused by the codec, exercised by
    the reference's own tests): the oracle's generic network against vectors from
    the compiled reference, perfect reconstruction, and - when oracle/_ref is
    present - the reference itself on fresh inputs."""
    g = load("filters.npz")
    o = oracle()
    for n in (8, 16, 32):
        x = g["x%d" % n]
        for i in range(len(x)):
            a = np.zeros(n, np.int32)
            assert o.odo_pre_filter(n, P(a), P(x[i])) == 0
            assert np.array_equal(a, g["pre%d" % n][i])
            b = np.zeros(n, np.int32)
            assert o.odo_post_filter(n, P(b), P(x[i])) == 0
            assert np.array_equal(b, g["post%d" % n][i])
            c = np.zeros(n, np.int32)
            o.odo_post_filter(n, P(c), P(a))
            assert np.array_equal(c, x[i]), "pre -> post is the identity"
    t = np.zeros(4, np.int32)
    assert o.odo_pre_filter(12, P(t), P(t)) == -1
    # the generic network at n = 4 is od_pre_filter4 / od_post_filter4
    g4 = load("filter4.npz")
    for i in range(len(g4["x"])):
        a = np.zeros(4, np.int32)
        o.odo_pre_filter(4, P(a), P(g4["x"][i]))
        assert np.array_equal(a, g4["pre"][i])
        o.odo_post_filter(4, P(a), P(g4["x"][i]))
        assert np.array_equal(a, g4["post"][i])
    r = ref()
    if r is not None:
        rng = np.random.RandomState(5)
        for f in (1, 2, 3):
            n = 4 << f
            for _ in range(300):
                x = rng.randint(-(1 << 20), 1 << 20, size=n).astype(np.int32)
                a = np.zeros(n, np.int32)
                b = np.zeros(n, np.int32)
                o.odo_pre_filter(n, P(a), P(x))
                r.ref_pre_filter(f, P(b), P(x))
                assert np.array_equal(a, b)
                o.odo_post_filter(n, P(a), P(x))
                r.ref_post_filter(f, P(b), P(x))
                assert np.array_equal(a, b)


def _pyramid(lib, prefix, px, dec, pic):
    h, w = px.shape
    top = 4 - dec
    lv = [np.zeros((h, w), np.int32) for _ in range(top + 1)]
    arr = (ctypes.c_void_p * 5)(*[l.ctypes.data for l in lv])
    c = np.zeros((h, w), np.int32)
    getattr(lib, prefix + "forward_pyramid_plane")(arr, P(c), P(px), w, w, h, dec, pic[0], pic[1])
    return lv, c


def test_pyramid_golden():
    g = load("pyramid.npz")
    W, H = 192, 128
    planes = synth_frame(W, H, seed=2024)
    for dec, idx in ((0, 0), (1, 1)):
        for pic in ((W, H), (W - 8, H - 24)):
            lv, c = _pyramid(oracle(), "odo_", planes[idx], dec, pic)
            tag = "d%d_%dx%d" % (dec, pic[0], pic[1])
            for bs, l in enumerate(lv):
                assert np.array_equal(sha(l), g["%s_L%d" % (tag, bs)]), (tag, bs)
            assert np.array_equal(sha(c), g["%s_c" % tag])
            if pic == (W, H):
                assert np.array_equal(lv[1], g["%s_L1_full" % tag])
            # lossless round trip at every partition level
            h, w = planes[idx].shape
            for leaf, l in enumerate(lv):
                px = np.zeros((h, w), np.uint8)
                cc = np.zeros((h, w), np.int32)
                oracle().odo_inverse_level_plane(P(px), w, P(cc), P(l), w, h, dec, leaf,
                                                 pic[0], pic[1])
                assert np.array_equal(px, planes[idx])


def test_pvq_search_golden():
    g = load("pvq_search.npz")
    o = oracle()
    for n in (8, 15, 16, 32, 128):
        x, k, g2 = g["x_n%d" % n], g["k_n%d" % n], g["g2_n%d" % n]
        nb = len(x)
        y = np.zeros((nb, n), np.int32)
        cos = np.zeros(nb)
        o.odo_pvq_search_batch(P(x), n, P(k), P(y), P(g2), cd(0.147), None, P(cos),
                               ctypes.c_long(nb))
        assert np.array_equal(y, g["y_n%d" % n])
        assert np.array_equal(cos.view(np.int64), g["cos_n%d" % n].view(np.int64))
        k2 = g["k2_n%d" % n]
        cos2 = np.zeros(nb)
        o.odo_pvq_search_batch(P(x), n, P(k2), P(y), P(g2), cd(0.147), P(k), P(cos2),
                               ctypes.c_long(nb))
        assert np.array_equal(y, g["y2_n%d" % n])
        assert np.array_equal(cos2.view(np.int64), g["cos2_n%d" % n].view(np.int64))


def test_pvq_theta_golden():
    g = load("pvq_theta.npz")
    qt = load("quant_v20.npz")
    o = oracle()
    pos = 0
    for row, sd_want in zip(g["meta"], g["skip_diff"]):
        bs, pli, kf, band, n, q0, bt, gain, ith, mth, vk = [int(v) for v in row]
        x0 = np.ascontiguousarray(g["x0"][pos:pos + n])
        r0 = np.ascontiguousarray(g["r0"][pos:pos + n])
        a = [1, 16, 24, 32, 64, 96, 128, 256, 384, 512][band]
        qoff = int(qt["qm_offset"][bs, 1 if pli else 0])
        qq = np.ascontiguousarray(qt["qm"][qoff + a:qoff + a + n])
        qi = np.ascontiguousarray(qt["qm_inv"][qoff + a:qoff + a + n])
        out = np.zeros(n, np.int32)
        y = np.zeros(n, np.int32)
        i1, i2, i3 = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        sd = cd(0.25)
        got = o.odo_pvq_theta(P(out), P(x0), P(r0), n, q0, P(y), ctypes.byref(i1),
                              ctypes.byref(i2), ctypes.byref(i3), bt, ctypes.byref(sd), 1, kf,
                              pli, P(qq), P(qi), cd(0.147), 1, None)
        assert (got, i1.value, i2.value, i3.value) == (gain, ith, mth, vk)
        assert np.array_equal(out, g["out"][pos:pos + n])
        assert np.array_equal(y, g["y"][pos:pos + n])
        assert sd.value == sd_want
        pos += n


def test_quant_tables_fixture_sane():
    qt = load("quant_v20.npz")
    assert int(qt["quantizer"]) == 243
    assert qt["pvq_qm_q4"].shape == (3, 30)
    assert qt["qm_offset"][:, 0].tolist() == [0, 16, 80, 336, 1360]
    assert qt["beta"][1, 0, 1, 0] == 6144 and qt["beta"][1, 0, 0, 0] == 4096
    assert qt["beta"][1, 1, 2, 3] == 4096


# ---- live cross-check against the compiled reference (when available) ---------
needs_ref = pytest.mark.skipif(ref() is None, reason="oracle/_ref not built here")


@needs_ref
def test_oracle_vs_reference_transforms_random():
    o, r = oracle(), ref()
    rng = np.random.RandomState(77)
    for ln in range(5):
        n = 4 << ln
        x = rng.randint(-70000, 70000, size=(40, n, n)).astype(np.int32)
        a = np.zeros_like(x)
        b = np.zeros_like(x)
        o.odo_fdct_2d_batch(ln, P(a), P(x), ctypes.c_long(len(x)))
        r.ref_fdct_2d_batch(ln, P(b), P(x), ctypes.c_long(len(x)))
        assert np.array_equal(a, b)
        o.odo_idct_2d_batch(ln, P(a), P(x), ctypes.c_long(len(x)))
        r.ref_idct_2d_batch(ln, P(b), P(x), ctypes.c_long(len(x)))
        assert np.array_equal(a, b)


@needs_ref
def test_oracle_vs_reference_pyramid_random():
    o, r = oracle(), ref()
    rng = np.random.RandomState(78)
    px = rng.randint(0, 256, size=(128, 192)).astype(np.uint8)
    for dec in (0, 1):
        p = px if dec == 0 else np.ascontiguousarray(px[:64, :96])
        for pic in ((192, 128), (180, 100)):
            la, ca = _pyramid(o, "odo_", p, dec, pic)
            lb, cb = _pyramid(r, "ref_", p, dec, pic)
            assert all(np.array_equal(x, y) for x, y in zip(la, lb))
            assert np.array_equal(ca, cb)


@needs_ref
def test_oracle_vs_reference_scan_and_helpers():
    o, r = oracle(), ref()
    rng = np.random.RandomState(79)
    for n in (4, 8, 16, 32, 64):
        src = rng.randint(-1000, 1000, size=(n, n)).astype(np.int32)
        a = np.zeros(n * n, np.int32)
        b = np.zeros(n * n, np.int32)
        o.odo_raster_to_coding_order(P(a), n, P(src), n)
        r.ref_raster_to_coding_order(P(b), n, P(src), n)
        ln = min(n * n, 512)
        assert np.array_equal(a[:ln], b[:ln])
        ra = np.zeros((n, n), np.int32)
        rb = np.zeros((n, n), np.int32)
        o.odo_coding_order_to_raster(P(ra), n, P(a), n)
        r.ref_coding_order_to_raster(P(rb), n, P(b), n)
        assert np.array_equal(ra, rb)
    for _ in range(3000):
        n = int(rng.choice([8, 15, 32, 128]))
        x = (rng.randint(-30000, 30000, size=n) >> rng.randint(0, 12)).astype(np.int16)
        q0 = int(rng.randint(1, 600))
        beta = int(rng.choice([4096, 6144]))
        bs = int(rng.randint(0, 4))
        g1, g2 = ctypes.c_int32(), ctypes.c_int32()
        assert o.odo_pvq_compute_gain(P(x), n, q0, ctypes.byref(g1), beta, bs) == \
            r.ref_pvq_compute_gain(P(x), n, q0, ctypes.byref(g2), beta, bs)
        assert g1.value == g2.value
        cg = int(rng.randint(0, 20000))
        if cg * q0 < 2 ** 30:
            assert o.odo_gain_expand(cg, q0, beta) == r.ref_gain_expand(cg, q0, beta)
        assert o.odo_pvq_compute_k(cg, -1, -1, 1, n, beta, 1) == \
            r.ref_pvq_compute_k(cg, -1, -1, 1, n, beta, 1)
        th = int(rng.randint(-70000, 140000))
        assert o.odo_pvq_cos(th) == r.ref_pvq_cos(th)
        xi = x.astype(np.int32) << 3
        assert o.odo_vector_log_mag(P(xi), n) == r.ref_vector_log_mag(P(xi), n)


def test_cfl_flip_golden():
    """odo_cfl_flip against the fixture made from the reference's od_pvq_encode
    (tools/make_golden_cfl.py): the decision and the negated range."""
    g = load("cfl_flip.npz")
    o = oracle()
    import daala_amd.quant as Q
    qt = Q.QuantTables.load()
    for bs in range(4):
        qm, _ = qt.qm_slices(1, bs)
        x, r, flips = g["x%d" % bs], g["r%d" % bs], g["flip%d" % bs]
        assert 0 < flips.sum() < len(flips)
        for i in range(len(x)):
            r1 = r[i].copy()
            f = o.odo_cfl_flip(P(r1), P(x[i]), P(qm), bs)
            assert f == flips[i], (bs, i)
            n = min(r1.size, 512)
            assert np.array_equal(r1[1:n], -r[i][1:n] if f else r[i][1:n])
            assert r1[0] == r[i][0] and np.array_equal(r1[n:], r[i][n:])


@pytest.mark.skipif(ref() is None, reason="oracle/_ref not present")
def test_cfl_flip_vs_reference_random():
    """Live: the reference's od_pvq_encode mutates `ref` exactly as odo_cfl_flip."""
    o, r = oracle(), ref()
    import daala_amd.quant as Q
    qt = Q.QuantTables.load()
    rng = np.random.RandomState(99)
    for bs in range(4):
        n = 4 << bs
        qm, qmi = qt.qm_slices(1, bs)
        beta = np.array(list(qt.beta_band(1, bs)) + [4096] * 12, np.int32)[:12]
        for _ in range(40):
            amp = rng.choice([30, 300, 3000, 30000])
            x = (rng.laplace(size=n * n) * amp).astype(np.int32)
            rr = (rng.choice([1, -1]) * x * rng.choice([0.01, 0.2, 1.0])
                  + rng.laplace(size=n * n) * amp * rng.choice([0.3, 1, 5])).astype(np.int32)
            r1, r2, out = rr.copy(), rr.copy(), np.zeros(n * n, np.int32)
            r.ref_pvq_encode_block(P(r1), P(x), P(out), 37, 1, bs, P(beta), 1, P(qm), P(qmi), 1)
            o.odo_cfl_flip(P(r2), P(x), P(qm), bs)
            assert np.array_equal(r1, r2)


def _pad_planes(o, fr, w, h):
    cw, ch = (w + 1) // 2, (h + 1) // 2
    fw, fh = (w + 63) // 64 * 64, (h + 63) // 64 * 64
    off = 0
    out = []
    for pli, (pw, ph) in enumerate(((w, h), (cw, ch), (cw, ch))):
        src = np.ascontiguousarray(fr[off:off + pw * ph].reshape(ph, pw))
        off += pw * ph
        s = 1 if pli else 0
        dst = np.zeros((fh >> s, fw >> s), np.uint8)
        o.odo_img_plane_copy_pad(P(dst), fw >> s, fw >> s, fh >> s, P(src), pw, pw, ph)
        out.append(dst)
    return out


def test_image_pad_golden():
    """odo_img_plane_copy_pad against the padded input planes of the reference
    encoder (tools/make_golden_pad.py)."""
    g = load("image_pad.npz")
    o = oracle()
    i = 0
    while "frame%d" % i in g:
        w, h = (int(v) for v in g["size%d" % i])
        got = _pad_planes(o, g["frame%d" % i], w, h)
        for pli in range(3):
            assert np.array_equal(got[pli], g["pad%d_%d" % (i, pli)]), (w, h, pli)
        i += 1
    assert i >= 5


@pytest.mark.skipif(ref() is None, reason="oracle/_ref not present")
def test_image_pad_vs_reference_random():
    o, r = oracle(), ref()
    rng = np.random.RandomState(17)
    for (w, h) in [(1, 1), (2, 3), (63, 65), (100, 37), (129, 64), (320, 180)]:
        cw, ch = (w + 1) // 2, (h + 1) // 2
        fr = rng.randint(0, 256, size=w * h + 2 * cw * ch).astype(np.uint8)
        fw, fh = (w + 63) // 64 * 64, (h + 63) // 64 * 64
        outs = [np.zeros((fh >> s, fw >> s), np.uint8) for s in (0, 1, 1)]
        arr = (ctypes.c_void_p * 3)(*[a.ctypes.data for a in outs])
        dims = (ctypes.c_int * 6)()
        assert r.ref_image_copy_pad(P(fr), w, h, arr, dims) == 0
        got = _pad_planes(o, fr, w, h)
        for pli in range(3):
            assert np.array_equal(got[pli], outs[pli]), (w, h, pli)


@pytest.mark.skipif(ref() is None, reason="oracle/_ref not present")
def test_resample_luma_coeffs_vs_reference_random():
    """odo_resample_luma_coeffs (4:2:0: the TF branch for 4x4 luma blocks and the
    quarter copy) against the reference's od_resample_luma_coeffs (src/intra.c:72)."""
    o, r = oracle(), ref()
    rng = np.random.RandomState(72)
    for bs, luma_bs in [(0, 0), (0, 1), (1, 2), (2, 3), (3, 4)]:
        n = 4 << bs
        for _ in range(40):
            luma = (rng.laplace(size=(2 * n, 2 * n)) * rng.choice([10, 300, 20000])).astype(np.int32)
            a = np.full((n, n + 3), -7, np.int32)
            b = a.copy()
            r.od_resample_luma_coeffs(P(a), n + 3, P(luma), 2 * n, 1, 1, bs, luma_bs)
            o.odo_resample_luma_coeffs(P(b), n + 3, P(luma), 2 * n, bs, luma_bs)
            assert np.array_equal(a, b), (bs, luma_bs)
