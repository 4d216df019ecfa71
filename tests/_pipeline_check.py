"""Whole-frame checker of the GPU frame-batch step (odhip_pipe) - TEST INFRASTRUCTURE,
shared by tests/test_gpu_pipeline.py, __graft_entry__.smoke() and the cpu_baseline /
verification leg of bench.py.

cpu_frame()         one 4:2:0 picture through the REFERENCE's own C functions
                    (oracle/_ref: ref_stage_plane_levels = padding, forward pyramid,
                    pvq_theta with closed-form pricing on every block of every level -
                    chroma WITH the chroma-from-luma reference -, inverse), keeping the
                    reconstruction of every level.
gpu_priced_frame()  the same picture through an F = 1 odhip_pipe stage by stage, with
                    the host's part in between: every candidate the band stages
                    searched is priced with the reference's od_pvq_rate (closed form)
                    and the rate tables go back for the choice - the split of
                    pvq_theta DESIGN.md describes, in batch form.
gpu_device_priced() the same picture(s) through odhip_pipe_step with price = 1: the
                    closed form evaluated on the device inside the choice kernels, one C
                    call per step, nothing in between.
All must agree bit for bit on every reconstructed pixel of every level."""
import ctypes
import time

import numpy as np

from _libs import P, oracle, ref

NBANDS = [1, 4, 7, 9, 9]


def _tables(qt, p):
    """p = plane index: the QM depends on the decimation only, the per-band steps on the plane
    (pvq_qm_q4[pli])."""
    qm_off = (ctypes.c_int * 5)(*[int(qt.qm_offset[bs][1 if p else 0]) for bs in range(5)])
    qb = (ctypes.c_int * 60)()
    bb = (ctypes.c_int * 60)()
    for bs in range(5):
        for i, v in enumerate(qt.q_band(p, bs)):
            qb[bs * 12 + i] = v
        for i, v in enumerate(qt.beta_band(p, bs)):
            bb[bs * 12 + i] = v
    return qm_off, qb, bb


def cpu_frame(qt, pics, pic_w, pic_h, chroma_cfl=True, lam=0.147, lib=None, fpr_bits=0, inter_pred=None,
              decisions=None, planes_out=None):
    """pics: [Y, Cb, Cr] uint8 pictures.  Returns (recon, blocks, seconds): recon[pli][bs]
    = uint8 plane of the coded size, reconstructed at uniform partition level bs.  lib:
    another build of the reference (the x86-intrinsics one) instead of oracle/_ref's default.
    fpr_bits = 8 / 10 / 12: full-precision references - pictures of that depth, planes of
    int16 samples at 12 bits, the reference's xstride-2 conversions (recon is uint16).
    inter_pred = [Y, Cb, Cr] prediction pictures: an INTER frame - every plane through
    pvq_theta with is_keyframe = 0 against the pyramid of its prediction.
    decisions: a list that receives, per plane, [(y, band)] per level - what pvq_theta decided
    for every band of every block (ref_stage_set_dump: y int32 [blocks][len] = the pulse
    vectors in coding order, band int32 [blocks][nb][4] = {coded gain index (the return
    value), itheta, max_theta, k}), blocks in raster order."""
    r = lib if lib is not None else ref()
    r.ref_set_fpr(1 if fpr_bits else 0)
    r.ref_stage_set_inter(1 if inter_pred is not None else 0)
    try:
        return _cpu_frame(r, qt, pics, pic_w, pic_h, chroma_cfl, lam, fpr_bits, inter_pred, decisions,
                          planes_out)
    finally:
        r.ref_set_fpr(0)
        r.ref_stage_set_inter(0)
        if decisions is not None:
            r.ref_stage_set_dump(None, None)


def _pad(px, pic, fpr_bits):
    h, w = px.shape
    if fpr_bits:
        oracle().odo_img_plane_copy_pad16(P(px), w, w, h, P(pic), fpr_bits, pic.shape[1], pic.shape[1],
                                          pic.shape[0])
    else:
        oracle().odo_img_plane_copy_pad(P(px), w, w, h, P(pic), pic.shape[1], pic.shape[1], pic.shape[0])


def _cpu_frame(r, qt, pics, pic_w, pic_h, chroma_cfl, lam, fpr_bits, inter_pred, decisions=None,
               planes_out=None):
    """planes_out: a list that receives, per plane, {"dq": the dequantised coefficient plane of
    every level (what pvq_theta synthesised "like the decoder would"), "ref": the reference
    plane of every level (chroma from luma; None for luma)} - keyframes only."""
    pdt = np.uint16 if fpr_bits else np.uint8
    assert r is not None, "oracle/_ref/libdaalaref.so not built"
    r.ref_stage_plane_levels.restype = ctypes.c_long
    W, H = (pic_w + 63) & ~63, (pic_h + 63) & ~63
    qm = np.ascontiguousarray(qt.qm)
    qmi = np.ascontiguousarray(qt.qm_inv)
    recon = []
    blocks = 0
    busy = 0.0
    ldq = [np.zeros((H, W), np.int32) for _ in range(5)]
    refs = None
    for pli, dec in ((0, 0), (1, 1), (2, 1)):
        p = 1 if pli else 0
        h, w = H >> dec, W >> dec
        # the per-band steps are per PLANE: pvq_qm_q4[pli], Cb != Cr (src/encode.c:3052-3072)
        qm_off, qb, bb = _tables(qt, pli)
        pic = np.ascontiguousarray(pics[pli])
        px = np.zeros((h, w), pdt)
        nlev = 5 - dec
        rec = [np.zeros((h, w), pdt) for _ in range(nlev)]
        rec_arr = (ctypes.c_void_p * 5)(*([a.ctypes.data for a in rec] + [None] * (5 - nlev)))
        if decisions is not None:
            dump = []
            for bs in range(nlev):
                n = 4 << bs
                nblk = (h // n) * (w // n)
                dump.append((np.zeros((nblk, min(n * n, 512)), np.int32), np.zeros((nblk, NBANDS[bs], 4), np.int32)))
            # the shim keeps these two pointer tables until ref_stage_set_dump(NULL, NULL)
            ytab = (ctypes.c_void_p * 5)(*([d[0].ctypes.data for d in dump] + [None] * (5 - nlev)))
            btab = (ctypes.c_void_p * 5)(*([d[1].ctypes.data for d in dump] + [None] * (5 - nlev)))
            r.ref_stage_set_dump(ytab, btab)
            decisions.append(dump)
        t0 = time.perf_counter()
        # od_img_plane_copy_pad is file-static in the reference's encode.c: the restatement
        # (pinned to the encoder's own padded input, tests/test_oracle_golden.py)
        _pad(px, pic, fpr_bits)
        if inter_pred is not None:
            ppx = np.zeros((h, w), pdt)
            _pad(ppx, np.ascontiguousarray(inter_pred[pli]), fpr_bits)
            plev = [np.zeros((h, w), np.int32) for _ in range(5)]
            r.ref_forward_pyramid_plane((ctypes.c_void_p * 5)(*[a.ctypes.data for a in plev]),
                                        P(np.zeros((h, w), np.int32)), P(ppx), w, w, h, dec, pic_w, pic_h)
            arr = (ctypes.c_void_p * 5)(*[a.ctypes.data for a in plev])
            blocks += r.ref_stage_plane_levels(P(px), w, w, h, dec, pic_w, pic_h, p, P(qm), P(qmi), qm_off,
                                               qb, bb, ctypes.c_double(lam), rec_arr, None, arr)
        elif pli == 0:
            dq = (ctypes.c_void_p * 5)(*[a.ctypes.data for a in ldq])
            blocks += r.ref_stage_plane_levels(P(px), w, w, h, 0, pic_w, pic_h, 0, P(qm), P(qmi), qm_off,
                                               qb, bb, ctypes.c_double(lam), rec_arr, dq, None)
        elif chroma_cfl:
            arr = (ctypes.c_void_p * 5)(*([a.ctypes.data for a in refs] + [None]))
            cdq = [np.zeros((h, w), np.int32) for _ in range(4)] if planes_out is not None else None
            dqa = (ctypes.c_void_p * 5)(*([a.ctypes.data for a in cdq] + [None])) if cdq else None
            blocks += r.ref_stage_plane_levels(P(px), w, w, h, 1, pic_w, pic_h, 1, P(qm), P(qmi), qm_off,
                                               qb, bb, ctypes.c_double(lam), rec_arr, dqa, arr)
            if planes_out is not None:
                planes_out.append({"dq": cdq, "ref": [a.copy() for a in refs]})
        else:
            blocks += r.ref_stage_plane_levels(P(px), w, w, h, 1, pic_w, pic_h, 1, P(qm), P(qmi), qm_off,
                                               qb, bb, ctypes.c_double(lam), rec_arr, None, None)
        busy += time.perf_counter() - t0
        if decisions is not None:
            r.ref_stage_set_dump(None, None)
        if pli == 0 and planes_out is not None:
            planes_out.append({"dq": [a.copy() for a in ldq], "ref": None})
        if pli == 0 and chroma_cfl:
            # od_resample_luma_coeffs for luma blocks one size up (src/intra.c:97-108: the
            # upper-left quarter of the decoded block), timed like the GPU's kernel
            t0 = time.perf_counter()
            refs = []
            for bs in range(4):
                n = 4 << bs
                c = ldq[bs + 1].reshape(H // (2 * n), 2 * n, W // (2 * n), 2 * n)[:, :n, :, :n]
                refs.append(np.ascontiguousarray(c.reshape(H // 2, W // 2)))
            busy += time.perf_counter() - t0
        recon.append(rec)
    return recon, blocks, busy


def _price_lib():
    """The reference's od_pvq_rate when oracle/_ref is here, the restatement otherwise."""
    r = ref()
    if r is not None:
        return r.ref_price_noref, r.ref_price_ref
    o = oracle()
    return o.odo_price_noref, o.odo_price_ref


def price_levels(D, pipe, set_, with_ref, pli):
    """Host pricing of every candidate of a plane set's band stage -> its rate tables."""
    price_noref, price_ref = _price_lib()
    for bs in range(5 - (1 if set_ else 0)):
        nb, offs, ln = D.pvq_band_layout(bs)
        off = (ctypes.c_int * 13)(*offs)
        B = pipe.nblocks(set_, bs)
        rec = pipe.read(D.BUF_BAND, set_, bs)
        y = pipe.read(D.BUF_Y, set_, bs, dtype=np.int16)
        if with_ref:
            items = pipe.read(D.BUF_ITEMS, set_, bs)
            rate = np.zeros((B, nb, 17), np.float64)
            price_ref(P(rate), P(rec), P(items), P(y), ctypes.c_long(B), nb, off, ln, 1, pli)
        else:
            rate = np.zeros((B, nb, 2), np.float64)
            price_noref(P(rate), P(rec), P(y), ctypes.c_long(B), nb, off, ln, 1, pli)
        pipe.write(D.BUF_RATE, set_, bs, rate)


def gpu_priced_frame(D, qt, pics, pic_w, pic_h, chroma_cfl=True, lam=0.147, frames=None):
    """pics: [Y, Cb, Cr] of one picture (or stacked [F,...] arrays with frames=F).
    Returns recon[set][bs] = uint8 [nplanes, H/dec, W/dec]."""
    F = frames or 1
    luma = np.ascontiguousarray(pics[0]).reshape(F, pic_h, pic_w)
    chroma = np.concatenate([np.ascontiguousarray(pics[1]).reshape(F, pic_h // 2, pic_w // 2),
                             np.ascontiguousarray(pics[2]).reshape(F, pic_h // 2, pic_w // 2)])
    pipe = D.Pipe(qt, F, pic_w, pic_h, chroma_cfl=chroma_cfl, serial=True, pvq_norm_lambda=lam)
    try:
        pipe.set_pictures(luma, chroma)
        pipe.stage("image_copy_pad_luma")
        pipe.stage("forward_pyramid_luma")
        if not chroma_cfl:
            pipe.stage("image_copy_pad_chroma")
            pipe.stage("forward_pyramid_chroma")
        pipe.stage("pvq_noref_bands")
        price_levels(D, pipe, 0, False, 0)
        if not chroma_cfl:
            price_levels(D, pipe, 1, False, 1)
        pipe.stage("pvq_choose")
        if chroma_cfl:
            pipe.stage("cfl_refs_from_luma")
        pipe.stage("dequant_inverse_luma")
        if chroma_cfl:
            pipe.stage("image_copy_pad_chroma")
            pipe.stage("forward_pyramid_chroma")
            pipe.stage("pvq_ref_bands")
            price_levels(D, pipe, 1, True, 1)
            pipe.stage("pvq_ref_choose")
        pipe.stage("dequant_inverse_chroma")
        W, H = pipe.W, pipe.H
        out = [[pipe.read(D.BUF_RECON, 0, bs).reshape(F, H, W) for bs in range(5)],
               [pipe.read(D.BUF_RECON, 1, bs).reshape(2 * F, H // 2, W // 2) for bs in range(4)]]
    finally:
        pipe.destroy()
    return out


def gpu_device_priced(D, qt, pics, pic_w, pic_h, chroma_cfl=True, lam=0.147, frames=None, serial=False,
                      steps=None, fpr_bits=0, inter_pred=None, decisions=False):
    """The F pictures through `steps` odhip_pipe_step calls (each codes all F) of a price=1
    pipe.  Returns (recon like gpu_priced_frame(), bands the host libm re-decided) and, with
    decisions=True, gpu_decisions() of the last step as a third element."""
    F = frames or 1
    luma = np.ascontiguousarray(pics[0]).reshape(F, pic_h, pic_w)
    chroma = np.concatenate([np.ascontiguousarray(pics[1]).reshape(F, pic_h // 2, pic_w // 2),
                             np.ascontiguousarray(pics[2]).reshape(F, pic_h // 2, pic_w // 2)])
    pipe = D.Pipe(qt, F, pic_w, pic_h, chroma_cfl=chroma_cfl, serial=serial, pvq_norm_lambda=lam,
                  price=True, fpr_bits=fpr_bits, inter=inter_pred is not None)
    rdt = np.uint16 if fpr_bits else np.uint8
    try:
        pipe.set_pictures(luma, chroma)
        if inter_pred is not None:
            pipe.set_reference_pictures(
                np.ascontiguousarray(inter_pred[0]).reshape(F, pic_h, pic_w),
                np.concatenate([np.ascontiguousarray(inter_pred[1]).reshape(F, pic_h // 2, pic_w // 2),
                                np.ascontiguousarray(inter_pred[2]).reshape(F, pic_h // 2, pic_w // 2)]))
        for _ in range(steps or 2):
            pipe.step()
        pipe.flush()
        pipe.sync()
        W, H = pipe.W, pipe.H
        out = [[pipe.read(D.BUF_RECON, 0, bs, dtype=rdt).reshape(F, H, W) for bs in range(5)],
               [pipe.read(D.BUF_RECON, 1, bs, dtype=rdt).reshape(2 * F, H // 2, W // 2) for bs in range(4)]]
        reruns = pipe.price_reruns()
        dec = gpu_decisions(D, pipe) if decisions else None
    finally:
        pipe.destroy()
    return (out, reruns, dec) if decisions else (out, reruns)


def compare_frame(gpu, cpu, frame=0, frames=1):
    """Differences between gpu_priced_frame()'s planes of `frame` and cpu_frame()'s: a
    list of (plane, level, differing pixels); empty = bit-exact."""
    bad = []
    for pli in range(3):
        set_ = 1 if pli else 0
        plane = frame if pli == 0 else (pli - 1) * frames + frame
        for bs in range(5 - set_):
            d = int(np.count_nonzero(gpu[set_][bs][plane] != cpu[pli][bs]))
            if d:
                bad.append((pli, bs, d))
    return bad


def gpu_decisions(D, pipe):
    """What the pipe's last step decided for every band of every block, in the form
    cpu_frame(decisions=...) hands back: {(set, level): (y int32 [B][len], band int32
    [B][nb][4] = {coded gain index, itheta, max_theta, k}, coded bool [B][nb])}; `coded` =
    the band's pulses are defined (not skipped).  Read from the choice records and the pulse
    slot they name - the same fields the inverse stage consumes."""
    out = {}
    cfl = pipe.chroma_cfl
    for set_ in (0, 1):
        for bs in range(5 - set_):
            nb, offs, ln = D.pvq_band_layout(bs)
            B = pipe.nblocks(set_, bs)
            with_ref = bool(set_ and cfl)
            ych = np.zeros((B, ln), np.int32)
            band = np.zeros((B, nb, 4), np.int32)
            coded = np.zeros((B, nb), bool)
            yall = pipe.read(D.BUF_Y, set_, bs, dtype=np.int16)
            if with_ref:
                ch = pipe.read(D.BUF_CHOICE, set_, bs, dtype=np.int32).reshape(B, nb, 16)
                y = yall.reshape(-1, B, ln)
                for i in range(nb):
                    a, b = offs[i], offs[i + 1]
                    noref = ch[:, i, 2]
                    skip = ch[:, i, 6]
                    slot = ch[:, i, 9]
                    band[:, i, 0] = ch[:, i, 7]
                    band[:, i, 1] = ch[:, i, 3]
                    band[:, i, 2] = ch[:, i, 4]
                    band[:, i, 3] = ch[:, i, 5]
                    on = (skip == 0) & (slot >= 0)
                    idx = np.nonzero(on)[0]
                    v = y[slot[idx], idx, a:b].astype(np.int32)
                    v[noref[idx] == 0, -1] = 0       # a theta winner holds n - 1 pulses
                    ych[idx, a:b] = v
                    coded[:, i] = skip == 0
            else:
                ch = pipe.read(D.BUF_CHOICE, set_, bs, dtype=np.int32).reshape(B, nb, 4)
                y = yall.reshape(2, B, ln)
                for i in range(nb):
                    a, b = offs[i], offs[i + 1]
                    sel = ch[:, i, 0]
                    qg = ch[:, i, 1]
                    band[:, i, 0] = qg            # keyframe, no reference: the gain index itself
                    band[:, i, 1] = -1
                    band[:, i, 2] = 0
                    idx = np.nonzero(qg != 0)[0]
                    ych[idx, a:b] = y[sel[idx], idx, a:b]
                    # K of a no-reference winner = its pulse count (the band records of the
                    # corner bands are not written by the stage that decides inside the search)
                    band[:, i, 3] = np.abs(ych[:, a:b]).sum(axis=1)
                    coded[:, i] = True
            out[(set_, bs)] = (ych, band, coded)
    return out


def compare_decisions(gpu, cpu, frame=0, frames=1):
    """gpu: gpu_decisions(); cpu: the `decisions` list of cpu_frame() for picture `frame`.
    Returns a list of (plane, level, what, count) mismatches; empty = every gain index,
    theta, K and pulse vector equals the reference's."""
    bad = []
    for pli in range(3):
        set_ = 1 if pli else 0
        plane = frame if pli == 0 else (pli - 1) * frames + frame
        for bs, (yc, bc) in enumerate(cpu[pli]):
            yg, bg, coded = gpu[(set_, bs)]
            per = yc.shape[0]
            sl = slice(plane * per, (plane + 1) * per)
            yg, bg, coded = yg[sl], bg[sl], coded[sl]
            d = int(np.count_nonzero((bg != bc).any(axis=2)))
            if d:
                bad.append((pli, bs, "band", d))
            nb, offs, _ = _layout(bs)
            for i in range(nb):
                a, b = offs[i], offs[i + 1]
                on = coded[:, i]
                d = int(np.count_nonzero((yg[on, a:b] != yc[on, a:b]).any(axis=1)))
                if d:
                    bad.append((pli, bs, "y[band %d]" % i, d))
    return bad


def _layout(bs):
    import daala_amd as D
    return D.pvq_band_layout(bs)
