"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the
same seeded inputs.  Bit-exact or fail; fp64 results compared bit for bit."""
import ctypes

import numpy as np
import pytest

from _libs import P, oracle, synth_frame

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    import torch
    import daala_amd
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    daala_amd.init(0)
    return daala_amd


def _cuda(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("ln", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("exact32", [0, 1])
def test_dct_batch_matches_oracle(hip, ln, exact32):
    n = 4 << ln
    rng = np.random.RandomState(100 + ln)
    nb = 517  # ragged: not a multiple of the blocks per workgroup
    # IEEE-1180 style ranges scaled by OD_COEFF_SCALE (reference dct.c:8277)
    x = np.concatenate([
        (rng.randint(0, 511, size=(nb, n, n)) - 255) * 16,
        (rng.randint(0, 11, size=(nb, n, n)) - 5) * 16,
        (rng.randint(0, 601, size=(nb, n, n)) - 300) * 16,
    ]).astype(np.int32)
    o = oracle()
    yo = np.zeros_like(x)
    o.odo_fdct_2d_batch(ln, P(yo), P(x), ctypes.c_long(len(x)))
    yg = hip.fdct2d_batch(ln, _cuda(x), exact32=exact32).cpu().numpy()
    assert np.array_equal(yg, yo)
    xg = hip.idct2d_batch(ln, _cuda(yo), exact32=exact32).cpu().numpy()
    assert np.array_equal(xg, x), "iDCT(fDCT(x)) must be the identity"


def test_dct_batch_empty_and_inplace(hip):
    import torch
    e = torch.empty((0, 8, 8), dtype=torch.int32, device="cuda")
    assert hip.fdct2d_batch(1, e).shape == (0, 8, 8)
    rng = np.random.RandomState(7)
    x = ((rng.randint(0, 511, size=(300, 16, 16)) - 255) * 16).astype(np.int32)
    yo = np.zeros_like(x)
    oracle().odo_fdct_2d_batch(2, P(yo), P(x), ctypes.c_long(len(x)))
    t = _cuda(x)
    hip.fdct2d_batch(2, t, out=t)
    assert np.array_equal(t.cpu().numpy(), yo)


def test_dct_exact32_arbitrary_input(hip):
    """The generic od_dct_func_2d surface must wrap like the C for any int32."""
    rng = np.random.RandomState(11)
    for ln in range(5):
        n = 4 << ln
        x = rng.randint(-2**20, 2**20, size=(64, n, n)).astype(np.int32)
        yo = np.zeros_like(x)
        oracle().odo_fdct_2d_batch(ln, P(yo), P(x), ctypes.c_long(len(x)))
        yg = hip.fdct2d_batch(ln, _cuda(x), exact32=1).cpu().numpy()
        assert np.array_equal(yg, yo)


@pytest.mark.parametrize("ln", [0, 2, 4])
def test_dct_plane_matches_oracle(hip, ln):
    n = 4 << ln
    rng = np.random.RandomState(5)
    h, w = 192, 320  # w, h multiples of 64; also try a ragged region below
    x = ((rng.randint(0, 511, size=(h, w)) - 255) * 16).astype(np.int32)
    yo = np.zeros_like(x)
    for by in range(h // n):
        for bx in range(w // n):
            oracle().odo_fdct_2d(ln, ctypes.c_void_p(yo.ctypes.data + 4 * (by * n * w + bx * n)), w,
                                 ctypes.c_void_p(x.ctypes.data + 4 * (by * n * w + bx * n)), w)
    yg = hip.fdct2d_plane(ln, _cuda(x)).cpu().numpy()
    assert np.array_equal(yg, yo)
    assert np.array_equal(hip.idct2d_plane(ln, _cuda(yo)).cpu().numpy(), x)


def _oracle_pyramid(px, dec, pic_w, pic_h):
    h, w = px.shape
    top = 4 - dec
    lv = [np.zeros((h, w), np.int32) for _ in range(top + 1)]
    arr = (ctypes.c_void_p * 5)(*[l.ctypes.data for l in lv])
    c = np.zeros((h, w), np.int32)
    oracle().odo_forward_pyramid_plane(arr, P(c), P(px), w, w, h, dec, pic_w, pic_h)
    return lv


@pytest.mark.parametrize("pic", [(320, 192), (312, 180)])
def test_forward_pyramid_matches_oracle(hip, pic):
    W, H = 320, 192
    planes = synth_frame(W, H, seed=99)
    rng = np.random.RandomState(1)
    planes[0] = rng.randint(0, 256, size=(H, W)).astype(np.uint8)  # full-range noise
    for dec, idx in ((0, [0]), (1, [1, 2])):
        px = np.stack([planes[i] for i in idx])
        got = hip.forward_pyramid(_cuda(px), dec, pic[0], pic[1])
        for p, i in enumerate(idx):
            want = _oracle_pyramid(planes[i], dec, pic[0], pic[1])
            for bs in range(5 - dec):
                assert np.array_equal(got[bs][p].cpu().numpy(), want[bs]), (dec, p, bs)


@pytest.mark.parametrize("dec", [0, 1])
def test_inverse_level_matches_oracle_and_reconstructs(hip, dec):
    W, H = 256, 192
    pic = (250, 180)
    planes = synth_frame(W, H, seed=3)
    px = planes[0] if dec == 0 else planes[1]
    h, w = px.shape
    levels = _oracle_pyramid(px, dec, *pic)
    for leaf in range(5 - dec):
        got = hip.inverse_level(_cuda(levels[leaf][None]), dec, leaf, *pic).cpu().numpy()[0]
        want = np.zeros((h, w), np.uint8)
        c = np.zeros((h, w), np.int32)
        oracle().odo_inverse_level_plane(P(want), w, P(c), P(levels[leaf]), w, h, dec, leaf, *pic)
        assert np.array_equal(got, want), leaf
        assert np.array_equal(got, px), "lossless round trip through the lapped transform"
    # quantised (perturbed) coefficients: exercises clamping and rounding
    rng = np.random.RandomState(8)
    d = (levels[1] // 64 * 64 + rng.randint(-40, 40, size=levels[1].shape)).astype(np.int32)
    got = hip.inverse_level(_cuda(d[None]), dec, 1, *pic).cpu().numpy()[0]
    want = np.zeros((h, w), np.uint8)
    c = np.zeros((h, w), np.int32)
    oracle().odo_inverse_level_plane(P(want), w, P(c), P(d), w, h, dec, 1, *pic)
    assert np.array_equal(got, want)


def _oracle_search(x, k, g2, lam, prev_k=None, y0=None):
    nb, n = x.shape
    y = np.zeros((nb, n), np.int32) if y0 is None else y0.copy()
    cos = np.zeros(nb, np.float64)
    oracle().odo_pvq_search_batch(P(x), n, P(k), P(y), P(g2), ctypes.c_double(lam),
                                  None if prev_k is None else P(prev_k), P(cos),
                                  ctypes.c_long(nb))
    return y, cos


@pytest.mark.parametrize("n", [16, 15, 8, 32, 128, 7, 14, 31, 127])
def test_pvq_search_matches_oracle(hip, n):
    rng = np.random.RandomState(200 + n)
    nb = 3001
    x = np.where(rng.rand(nb, 1) < .5, rng.randint(-1000, 1001, size=(nb, n)),
                 rng.laplace(0, 200, size=(nb, n))).astype(np.int16)
    x[0] = 0  # null vector
    x[1] = 17  # all ties
    k = rng.choice([1, 2, 3, 4, 8, 16, 33], size=nb).astype(np.int32)
    g2 = rng.choice([1.0, 0.01, 37.5, 1e4], size=nb).astype(np.float64)
    yo, co = _oracle_search(x, k, g2, 0.147)
    yg, cg = hip.pvq_search_batch(_cuda(x), _cuda(k), _cuda(g2), 0.147)
    assert np.array_equal(yg.cpu().numpy(), yo)
    assert np.array_equal(cg.cpu().numpy().view(np.int64), co.view(np.int64)), "cosine bit-exact"
    # chained search reusing the previous pulses (prev_k > 0)
    k2 = (k + rng.randint(0, 6, size=nb)).astype(np.int32)
    yo2, co2 = _oracle_search(x, k2, g2, 0.147, prev_k=k, y0=yo)
    yg2, cg2 = hip.pvq_search_batch(_cuda(x), _cuda(k2), _cuda(g2), 0.147,
                                    prev_k=_cuda(k), y=_cuda(yo))
    assert np.array_equal(yg2.cpu().numpy(), yo2)
    assert np.array_equal(cg2.cpu().numpy().view(np.int64), co2.view(np.int64))


@pytest.mark.parametrize("f", [0, 1, 2, 3])
def test_filter_batch_matches_oracle(hip, f):
    """od_pre_filterN / od_post_filterN, N = 4 << f, batched and per call."""
    import ctypes as ct
    n = 4 << f
    rng = np.random.RandomState(60 + f)
    x = np.concatenate([rng.randint(-a, a + 1, size=(2500, n)) for a in (3, 255, 4096, 1 << 19)])
    x = x.astype(np.int32)
    o = oracle()
    for inverse in (False, True):
        want = np.zeros_like(x)
        fn = o.odo_post_filter if inverse else o.odo_pre_filter
        for i in range(len(x)):
            fn(n, P(want[i]), P(x[i]))
        got = hip.filter_batch(f, _cuda(x), inverse=inverse)
        assert np.array_equal(got.cpu().numpy(), want), (f, inverse)
    # in place, and pre -> post is the identity at full size
    t = _cuda(x)
    hip.filter_batch(f, t, out=t)
    hip.filter_batch(f, t, inverse=True, out=t)
    assert np.array_equal(t.cpu().numpy(), x)
    assert hip.filter_batch(f, _cuda(x[:0])).shape == (0, n)
    # per-call host-pointer surface and the table installer (od_filter_func shape)
    L = hip.lib()
    FT = ct.CFUNCTYPE(None, ct.c_void_p, ct.c_void_p)
    pre = (ct.c_void_p * 4)()
    post = (ct.c_void_p * 4)()
    L.odhip_install_filter_tables(pre, post)
    a = np.zeros(n, np.int32)
    b = np.zeros(n, np.int32)
    FT(pre[f])(a.ctypes.data, x[7].ctypes.data)
    o.odo_pre_filter(n, P(b), P(x[7]))
    assert np.array_equal(a, b)
    FT(post[f])(a.ctypes.data, x[7].ctypes.data)
    o.odo_post_filter(n, P(b), P(x[7]))
    assert np.array_equal(a, b)


def test_per_call_surfaces(hip):
    """The reference-signature host-pointer entry points (drop-in surface)."""
    rng = np.random.RandomState(42)
    o = oracle()
    for ln in range(5):
        n = 4 << ln
        x = ((rng.randint(0, 511, size=(n, n)) - 255) * 16).astype(np.int32)
        yo = np.zeros_like(x)
        o.odo_fdct_2d(ln, P(yo), n, P(x), n)
        assert np.array_equal(hip.host.dct2d(ln, x), yo)
        assert np.array_equal(hip.host.dct2d(ln, yo, inverse=True), x)
    x = rng.randint(-500, 500, size=16).astype(np.int16)
    yo = np.zeros(16, np.int32)
    o.odo_pvq_search_rdo_double.restype = ctypes.c_double
    co = o.odo_pvq_search_rdo_double(P(x), 16, 5, P(yo), ctypes.c_double(2.5),
                                     ctypes.c_double(0.147), 0)
    yg, cg = hip.host.pvq_search(x, 5, 2.5, 0.147)
    assert np.array_equal(yg, yo) and cg == co
    # filters on a host plane
    L = hip.lib()
    W, H = 128, 128
    c = ((rng.randint(0, 256, size=(H, W)) - 128) * 16).astype(np.int32)
    a = c.copy()
    b = c.copy()
    o.odo_apply_prefilter_frame_sbs(P(a), W, 2, 2, 0, 0)
    L.od_apply_prefilter_frame_sbs_hip(P(b), W, 2, 2, 0, 0)
    assert np.array_equal(a, b)
    o.odo_prefilter_split(P(a), W, 3, 1, 1)
    L.od_prefilter_split_hip(P(b), W, 3, 0, 1, 1)
    assert np.array_equal(a, b)
    o.odo_postfilter_split(P(a), W, 3, 1, 0)
    L.od_postfilter_split_hip(P(b), W, 3, 0, 0, None, 0, 1, 0)
    assert np.array_equal(a, b)
    o.odo_apply_postfilter_frame_sbs(P(a), W, 2, 2, 0, 0)
    L.od_apply_postfilter_frame_sbs_hip(P(b), W, 2, 2, 0, 0, 0, None, 0)
    assert np.array_equal(a, b)


@pytest.mark.parametrize("pic,plane", [((70, 50), (128, 64)), ((1920, 1080), (1920, 1088)),
                                       ((960, 540), (960, 544)), ((33, 17), (64, 64)),
                                       ((64, 64), (64, 64)), ((65, 129), (128, 192)),
                                       ((1, 1), (32, 32)), ((0, 0), (64, 64))])
def test_image_planes_copy_pad_matches_oracle(hip, pic, plane):
    """odhip_image_planes_copy_pad (od_img_plane_copy_pad, src/encode.c:752-837) on a
    batch of planes vs the oracle (pinned to the reference encoder's padded input)."""
    import torch
    o = oracle()
    (pw, ph), (fw, fh) = pic, plane
    rng = np.random.RandomState(pw * 7 + ph)
    nplanes = 3
    src = rng.randint(0, 256, size=(nplanes, ph, pw)).astype(np.uint8)
    out = torch.full((nplanes, fh, fw), 77, dtype=torch.uint8, device="cuda")
    hip.image_planes_copy_pad(torch.from_numpy(src).cuda(), fw, fh, out=out)
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    for p in range(nplanes):
        want = np.zeros((fh, fw), np.uint8)
        s = np.ascontiguousarray(src[p]) if pw and ph else np.zeros((1, 1), np.uint8)
        o.odo_img_plane_copy_pad(P(want), fw, fw, fh, P(s), max(pw, 1), pw, ph)
        assert np.array_equal(got[p], want), (pic, p)


def test_inverse_levels_equals_inverse_level_per_level(hip):
    """odhip_inverse_levels (several partition levels of one plane set in one set of
    launches) gives exactly what odhip_inverse_level gives level by level."""
    import torch
    rng = np.random.RandomState(4)
    for dec, (h, w) in ((0, (128, 192)), (1, (64, 96))):
        top = 4 - dec
        coefs = [torch.from_numpy(rng.randint(-3000, 3000, size=(2, h, w)).astype(np.int32)).cuda()
                 for _ in range(top + 1)]
        got = hip.inverse_levels(coefs, dec, list(range(top + 1)), 2 * w if dec else w, 2 * h - 5 if dec else h - 5)
        torch.cuda.synchronize()
        for bs in range(top + 1):
            want = hip.inverse_level(coefs[bs], dec, bs, 2 * w if dec else w, 2 * h - 5 if dec else h - 5)
            assert torch.equal(got[bs], want), (dec, bs)
