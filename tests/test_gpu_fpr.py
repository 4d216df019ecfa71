"""Full-precision references on the GPU (odhip_ctx_set_fpr, odhip_image_planes_copy_pad16):
16-bit picture planes at 12 bits through the input padding, the forward pyramid and the
inverse, bit-exact against the oracle in FPR mode (itself pinned to the reference's xstride-2
branches and to a real FPR encoder's padded input, tests/test_oracle_fpr.py)."""
import ctypes

import numpy as np
import pytest

from _libs import P, oracle

pytestmark = pytest.mark.gpu


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


@pytest.mark.parametrize("bitdepth", [8, 10, 12])
def test_image_pad16_matches_oracle(hip, bitdepth):
    o = oracle()
    rng = np.random.RandomState(bitdepth)
    for (pw, ph, plane_w, plane_h) in ((70, 50, 128, 64), (64, 64, 64, 64), (1, 1, 64, 64), (1920, 1080, 1920, 1088),
                                       (0, 0, 64, 64), (100, 64, 128, 64), (128, 37, 128, 64)):
        if bitdepth == 8:
            src = rng.randint(0, 256, size=(3, ph, pw)).astype(np.uint8)
        else:
            src = rng.randint(0, 1 << bitdepth, size=(3, ph, pw)).astype(np.int16)
        got = hip.image_planes_copy_pad16(_cuda(src), plane_w, plane_h, bitdepth).cpu().numpy().view(np.uint16)
        for p in range(3):
            want = np.zeros((plane_h, plane_w), np.uint16)
            o.odo_img_plane_copy_pad16(P(want), plane_w, plane_w, plane_h,
                                       P(np.ascontiguousarray(src[p])) if src[p].size else None, bitdepth,
                                       pw, pw, ph)
            assert np.array_equal(got[p], want), (bitdepth, pw, ph, p)


@pytest.mark.parametrize("dec,shape", [(0, (2, 192, 320)), (1, (4, 96, 160)), (0, (1, 64, 64))])
def test_fpr_pyramid_and_inverse_match_oracle(hip, dec, shape):
    o = oracle()
    rng = np.random.RandomState(31 + dec)
    nplanes, h, w = shape
    px = rng.randint(0, 4096, size=shape).astype(np.int16)
    px[:, :8, :8] = 4095
    px[:, 8:16, :8] = 0
    pic_w, pic_h = (w << dec) - 20, (h << dec) - 12      # the split filters are gated by the picture size
    top = 4 - dec
    ctx = hip.Context(0).set_fpr(True)
    try:
        with ctx:
            levels = hip.forward_pyramid(_cuda(px), dec, pic_w, pic_h)
            assert hip.px_dtype().itemsize == 2
            d = [np.ascontiguousarray(levels[bs].cpu().numpy() + rng.randint(-700, 701, size=shape).astype(np.int32))
                 for bs in range(top + 1)]
            recon = hip.inverse_levels([_cuda(a) for a in d], dec, list(range(top + 1)), pic_w, pic_h)
            recon = [t.cpu().numpy().view(np.uint16) for t in recon]
        o.odo_set_fpr(1)
        for p in range(nplanes):
            want = [np.zeros((h, w), np.int32) for _ in range(5)]
            c = np.zeros((h, w), np.int32)
            o.odo_forward_pyramid_plane((ctypes.c_void_p * 5)(*[a.ctypes.data for a in want]), P(c),
                                        P(np.ascontiguousarray(px[p])), w, w, h, dec, pic_w, pic_h)
            for bs in range(top + 1):
                assert np.array_equal(levels[bs][p].cpu().numpy(), want[bs]), (dec, p, bs)
                rec = np.zeros((h, w), np.uint16)
                cc = np.zeros((h, w), np.int32)
                o.odo_inverse_level_plane(P(rec), w, P(cc), P(np.ascontiguousarray(d[bs][p])), w, h, dec, bs,
                                          pic_w, pic_h)
                assert np.array_equal(recon[bs][p], rec), (dec, p, bs)
                assert int(rec.max()) <= 4095
    finally:
        o.odo_set_fpr(0)
        ctx.destroy()


def test_fpr_of_an_8bit_source_is_the_8bit_pyramid(hip):
    """A 12-bit plane that is an 8-bit picture shifted up by OD_COEFF_SHIFT gives the very
    coefficients of the 8-bit path ((p << 4) - 2048 == (p - 128) << 4)."""
    rng = np.random.RandomState(2)
    p8 = rng.randint(0, 256, size=(2, 128, 192)).astype(np.uint8)
    a = hip.forward_pyramid(_cuda(p8), 0, 192, 128)
    ctx = hip.Context(0).set_fpr(True)
    try:
        with ctx:
            b = hip.forward_pyramid(_cuda((p8.astype(np.int16) << 4)), 0, 192, 128)
    finally:
        ctx.destroy()
    import torch
    for bs in range(5):
        assert torch.equal(a[bs], b[bs]), bs


@pytest.mark.parametrize("bits", [8, 10, 12])
def test_fpr_whole_frame_step_equals_compiled_reference(hip, bits):
    """The frame-batch step in full-precision-references mode (odhip_pipe_config.fpr_bits):
    pictures of 8 / 10 / 12 bits, the choice priced on the device, every reconstructed
    12-bit sample of every level of Y, Cb, Cr against the reference's own C functions in
    FPR mode."""
    from _libs import ref
    if ref() is None:
        pytest.skip("oracle/_ref not present")
    import sys
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    import _pipeline_check as C
    qt = hip.QuantTables.for_quality(40)
    full = bench.natural_like_frame_np(4, 77)
    pw, ph = 312, 180
    rng = np.random.RandomState(bits)
    pics = []
    for pl, (w_, h_) in zip(full, ((pw, ph), (pw // 2, ph // 2), (pw // 2, ph // 2))):
        p8 = pl[:h_, :w_].astype(np.int32)
        if bits == 8:
            pics.append(p8.astype(np.uint8))
        else:
            lo = rng.randint(0, 1 << (bits - 8), size=p8.shape)      # use the extra bits
            pics.append(((p8 << (bits - 8)) + lo).astype(np.int16))
    for cfl in (True, False):
        cpu, _, _ = C.cpu_frame(qt, pics, pw, ph, chroma_cfl=cfl, fpr_bits=bits)
        gpu, _ = C.gpu_device_priced(hip, qt, pics, pw, ph, chroma_cfl=cfl, fpr_bits=bits)
        assert C.compare_frame(gpu, cpu) == [], (bits, cfl)
        assert int(cpu[0][0].max()) > 255          # really 12-bit samples
    # and as an inter frame against a prediction of the same depth
    pred = []
    for p_ in pics:
        q = np.roll(p_.astype(np.int32), 1, axis=1) + rng.randint(-3, 4, size=p_.shape) * (1 << (bits - 8))
        pred.append(np.clip(q, 0, (1 << bits) - 1).astype(p_.dtype))
    cpu, _, _ = C.cpu_frame(qt, pics, pw, ph, fpr_bits=bits, inter_pred=pred)
    gpu, _ = C.gpu_device_priced(hip, qt, pics, pw, ph, fpr_bits=bits, inter_pred=pred)
    assert C.compare_frame(gpu, cpu) == [], (bits, "inter")


@pytest.mark.parametrize("fpr", [False, True])
def test_inverse_partition_uniform_map_equals_inverse_level(hip, fpr):
    """odhip_inverse_partition with a block-size map that says `level L everywhere` is
    odhip_inverse_levels at level L - luma, and chroma (one size down, 4x4 for 8x8 and 4x4
    luma) - for 8-bit planes and for the int16 planes of a full-precision-references
    context; plus a mixed map: every superblock at its own level."""
    import torch
    L_ = hip.lib()
    rng = np.random.RandomState(12)
    F, h, w = 2, 128, 256
    ctx = hip.Context(0).set_fpr(fpr)
    try:
        with ctx:
            dt = hip.px_dtype()
            for dec in (0, 1):
                hh, ww = h >> dec, w >> dec
                nplanes = F * (2 if dec else 1)
                top = 4 - dec
                coefs = [_cuda((rng.randint(-200, 201, size=(nplanes, hh, ww)) * 8).astype(np.int32))
                         for _ in range(top + 1)]
                want = hip.inverse_levels(coefs, dec, list(range(top + 1)), w, h)
                bstride = (w // 64) * 8
                rows = (h // 64) * 8
                for lum in range(5):
                    lvl = max(lum - dec, 0)                       # the chroma block of a luma block
                    bsize = _cuda(np.full((F, rows, bstride), lum, np.uint8))
                    out = torch.empty((nplanes, hh, ww), dtype=dt, device="cuda")
                    rc = L_.odhip_inverse_partition(ctypes.c_void_p(out.data_ptr()), ww, ctypes.c_long(hh * ww),
                                                    ctypes.c_void_p(coefs[lvl].data_ptr()), nplanes, ww, hh, dec,
                                                    ctypes.c_void_p(bsize.data_ptr()), bstride,
                                                    ctypes.c_long(rows * bstride), 2 if dec else 1, w, h, None)
                    assert rc == 0
                    torch.cuda.synchronize()
                    assert torch.equal(out, want[lvl]), (fpr, dec, lum)
                if dec == 0:
                    # superblock (sx, sy) coded at level (sx + sy) % 5: every superblock must equal
                    # the uniform reconstruction of its own level, given coefficients that agree
                    mix = np.zeros((F, rows, bstride), np.uint8)
                    coef_mix = torch.empty_like(coefs[0])
                    for sy in range(h // 64):
                        for sx in range(w // 64):
                            lv = (sx + sy) % 5
                            mix[:, sy * 8:(sy + 1) * 8, sx * 8:(sx + 1) * 8] = lv
                            coef_mix[:, sy * 64:(sy + 1) * 64, sx * 64:(sx + 1) * 64] = \
                                coefs[lv][:, sy * 64:(sy + 1) * 64, sx * 64:(sx + 1) * 64]
                    out = torch.empty((nplanes, hh, ww), dtype=dt, device="cuda")
                    rc = L_.odhip_inverse_partition(ctypes.c_void_p(out.data_ptr()), ww, ctypes.c_long(hh * ww),
                                                    ctypes.c_void_p(coef_mix.data_ptr()), nplanes, ww, hh, 0,
                                                    ctypes.c_void_p(_cuda(mix).data_ptr()), bstride,
                                                    ctypes.c_long(rows * bstride), 1, w, h, None)
                    assert rc == 0
                    torch.cuda.synchronize()
                    # inside a superblock - away from the 2 samples the superblock-edge post-filter
                    # mixes with the neighbours - the samples are those of the uniform level
                    for sy in range(h // 64):
                        for sx in range(w // 64):
                            lv = (sx + sy) % 5
                            ys, xs = slice(sy * 64 + 2, sy * 64 + 62), slice(sx * 64 + 2, sx * 64 + 62)
                            assert torch.equal(out[:, ys, xs], want[lv][:, ys, xs]), (fpr, sx, sy, lv)
    finally:
        ctx.destroy()
