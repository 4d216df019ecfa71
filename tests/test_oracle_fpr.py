"""Full-precision references (info.full_precision_references: 16-bit picture buffers at 12
bits, src/encode.c:212-213): the oracle's 16-bit restatements pinned against the compiled
reference - the padded input of a real FPR encoder for 8-, 10- and 12-bit sources, the
pixel <-> coefficient conversions, and the plane-level stage functions in FPR mode."""
import ctypes

import numpy as np
import pytest

from _libs import P, oracle, ref


def _pad16(o, fr, w, h, bitdepth):
    cw, ch = (w + 1) // 2, (h + 1) // 2
    fw, fh = (w + 63) // 64 * 64, (h + 63) // 64 * 64
    off = 0
    out = []
    for pli, (pw, ph) in enumerate(((w, h), (cw, ch), (cw, ch))):
        src = np.ascontiguousarray(fr[off:off + pw * ph].reshape(ph, pw))
        off += pw * ph
        s = 1 if pli else 0
        dst = np.zeros((fh >> s, fw >> s), np.uint16)
        o.odo_img_plane_copy_pad16(P(dst), fw >> s, fw >> s, fh >> s, P(src), bitdepth, pw, pw, ph)
        out.append(dst)
    return out


@pytest.mark.skipif(ref() is None, reason="oracle/_ref not present")
@pytest.mark.parametrize("bitdepth", [8, 10, 12])
def test_fpr_padded_input_vs_reference_encoder(bitdepth):
    o, r = oracle(), ref()
    rng = np.random.RandomState(40 + bitdepth)
    for (w, h) in [(2, 2), (63, 65), (100, 38), (130, 64), (320, 180)]:
        cw, ch = (w + 1) // 2, (h + 1) // 2
        n = w * h + 2 * cw * ch
        if bitdepth == 8:
            fr = rng.randint(0, 256, size=n).astype(np.uint8)
        else:
            fr = rng.randint(0, 1 << bitdepth, size=n).astype(np.uint16)
        fw, fh = (w + 63) // 64 * 64, (h + 63) // 64 * 64
        outs = [np.zeros((fh >> s, fw >> s), np.uint16) for s in (0, 1, 1)]
        arr = (ctypes.c_void_p * 3)(*[a.ctypes.data for a in outs])
        dims = (ctypes.c_int * 6)()
        assert r.ref_image_copy_pad_fpr(P(fr), w, h, bitdepth, arr, dims) == 0
        got = _pad16(o, fr, w, h, bitdepth)
        for pli in range(3):
            assert np.array_equal(got[pli], outs[pli]), (w, h, pli)
            assert int(outs[pli].max()) < 4096


@pytest.mark.skipif(ref() is None, reason="oracle/_ref not present")
def test_fpr_plane_stage_vs_reference():
    """Pyramid and inverse of a plane of 12-bit samples: the oracle in FPR mode against the
    reference's functions in FPR mode (their xstride-2 conversion branches)."""
    o, r = oracle(), ref()
    rng = np.random.RandomState(5)
    for dec, (w, h) in ((0, (128, 64)), (1, (64, 64))):
        px = rng.randint(0, 4096, size=(h, w)).astype(np.uint16)
        px[:8, :8] = 4095
        px[8:16, :8] = 0
        top = 4 - dec
        try:
            o.odo_set_fpr(1)
            r.ref_set_fpr(1)
            lo = [np.zeros((h, w), np.int32) for _ in range(5)]
            lr = [np.zeros((h, w), np.int32) for _ in range(5)]
            co, cr = np.zeros((h, w), np.int32), np.zeros((h, w), np.int32)
            o.odo_forward_pyramid_plane((ctypes.c_void_p * 5)(*[a.ctypes.data for a in lo]), P(co), P(px), w, w, h,
                                        dec, w << dec, h << dec)
            r.ref_forward_pyramid_plane((ctypes.c_void_p * 5)(*[a.ctypes.data for a in lr]), P(cr), P(px), w, w, h,
                                        dec, w << dec, h << dec)
            for bs in range(top + 1):
                assert np.array_equal(lo[bs], lr[bs]), (dec, bs)
            for bs in range(top + 1):
                # a scaled level overflows 12 bits in places: the clamp of the store is exercised
                d = np.ascontiguousarray(lo[bs] + rng.randint(-600, 601, size=(h, w)).astype(np.int32))
                po, pr = np.zeros((h, w), np.uint16), np.zeros((h, w), np.uint16)
                o.odo_inverse_level_plane(P(po), w, P(co), P(d), w, h, dec, bs, w << dec, h << dec)
                r.ref_inverse_level_plane(P(pr), w, P(cr), P(d), w, h, dec, bs, w << dec, h << dec)
                assert np.array_equal(po, pr), (dec, bs)
                assert int(po.max()) <= 4095
        finally:
            o.odo_set_fpr(0)
            r.ref_set_fpr(0)
