"""GPU parity of the PVQ band stage (pvq_theta's no-reference path, candidates,
choice and dequantisation) against the CPU oracle, band by band."""
import ctypes

import numpy as np
import pytest

from _libs import P, oracle, synth_frame

pytestmark = pytest.mark.gpu
cd = ctypes.c_double
MAXN = 128


class Cand(ctypes.Structure):
    _fields_ = [("with_ref", ctypes.c_int32), ("gain", ctypes.c_int32),
                ("theta", ctypes.c_int32), ("ts", ctypes.c_int32), ("k", ctypes.c_int32),
                ("qcg", ctypes.c_int32), ("qtheta", ctypes.c_int32),
                ("searched", ctypes.c_int32), ("cos_dist", ctypes.c_double),
                ("dist", ctypes.c_double), ("y", ctypes.c_int32 * MAXN)]


class Trace(ctypes.Structure):
    _fields_ = [("xshift", ctypes.c_int32), ("rshift", ctypes.c_int32),
                ("g", ctypes.c_int32), ("gr", ctypes.c_int32), ("cg", ctypes.c_int32),
                ("cgr", ctypes.c_int32), ("icgr", ctypes.c_int32),
                ("gain_offset", ctypes.c_int32), ("m", ctypes.c_int32), ("s", ctypes.c_int32),
                ("theta", ctypes.c_int32), ("corr", ctypes.c_double),
                ("dist0", ctypes.c_double), ("skip_dist", ctypes.c_double),
                ("x16", ctypes.c_int16 * MAXN), ("r16", ctypes.c_int16 * MAXN),
                ("ncands", ctypes.c_int32), ("cands", Cand * 24)]


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


def _oracle_pyramid(px, dec, pic):
    h, w = px.shape
    lv = [np.zeros((h, w), np.int32) for _ in range(5 - dec)]
    arr = (ctypes.c_void_p * 5)(*[l.ctypes.data for l in lv])
    c = np.zeros((h, w), np.int32)
    oracle().odo_forward_pyramid_plane(arr, P(c), P(px), w, w, h, dec, pic[0], pic[1])
    return lv


@pytest.mark.parametrize("pli,dec", [(0, 0), (1, 1)])
def test_band_stage_matches_oracle(hip, pli, dec):
    W, H = 128, 128
    planes = synth_frame(W, H, seed=5)
    rng = np.random.RandomState(17)
    # stronger texture so that gains > 1 and multi-pulse searches occur
    planes[0] = np.clip(planes[0].astype(int) + rng.randint(-60, 61, size=(H, W)), 0, 255).astype(np.uint8)
    px = planes[0] if dec == 0 else planes[1]
    h, w = px.shape
    levels = _oracle_pyramid(px, dec, (W, H))
    qt = hip.QuantTables.load()
    o = oracle()
    lam = hip.OD_PVQ_LAMBDA
    for bs in range(5 - dec):
        n = 4 << bs
        coef = levels[bs]
        qm, qmi = qt.qm_slices(pli, bs)
        qb = qt.q_band(pli, bs)
        bb = qt.beta_band(pli, bs)
        nb, offs, ln = hip.pvq_band_layout(bs)
        tc = _cuda(coef[None])
        cands = hip.pvq_noref_bands(tc, bs, _cuda(qm), qb, bb, lam, cos_dist=True)
        import torch
        torch.cuda.synchronize()
        dq, qg = hip.pvq_select_synth_noref(tc, bs, _cuda(qmi), qb, bb, lam, cands)
        c = hip.unpack_cands(cands)
        dq = dq.cpu().numpy()[0]
        qg = qg.cpu().numpy()
        want_dq = np.zeros_like(coef)
        nsearched = 0
        for by in range(h // n):
            for bx in range(w // n):
                blk = by * (w // n) + bx
                vec = np.zeros(n * n, np.int32)
                o.odo_raster_to_coding_order(P(vec), n, ctypes.c_void_p(
                    coef.ctypes.data + 4 * (by * n * w + bx * n)), w)
                outvec = np.zeros(n * n, np.int32)
                outvec[0] = vec[0]
                for band in range(nb):
                    a, b = offs[band], offs[band + 1]
                    m = b - a
                    x0 = np.ascontiguousarray(vec[a:b])
                    r0 = np.zeros(m, np.int32)
                    out = np.zeros(m, np.int32)
                    y = np.zeros(m, np.int32)
                    i1, i2, i3 = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
                    sd = cd(0)
                    tr = Trace()
                    qq = np.ascontiguousarray(qm[a:b])
                    qi = np.ascontiguousarray(qmi[a:b])
                    o.odo_pvq_theta(P(out), P(x0), P(r0), m, qb[band], P(y), ctypes.byref(i1),
                                    ctypes.byref(i2), ctypes.byref(i3), bb[band],
                                    ctypes.byref(sd), 1, 1, 0, P(qq), P(qi), cd(lam), 1,
                                    ctypes.byref(tr))
                    assert c["cg"][blk, band] == tr.cg
                    assert c["dist0"][blk, band] == tr.dist0
                    nr = [tr.cands[i] for i in range(tr.ncands) if not tr.cands[i].with_ref]
                    assert len(nr) in (1, 2)
                    best_cost, best_qg, best_y = tr.dist0, 0, None
                    for slot in range(2):
                        if slot >= len(nr):
                            assert c["gain"][blk, band, slot] == 0 and c["flags"][blk, band, slot] == 0
                            continue
                        cnd = nr[slot]
                        assert c["gain"][blk, band, slot] == cnd.gain
                        assert c["k"][blk, band, slot] == cnd.k
                        assert c["flags"][blk, band, slot] == cnd.searched
                        if cnd.searched:
                            nsearched += 1
                            assert c["cos_dist"][blk, band, slot] == cnd.cos_dist
                            assert c["dist"][blk, band, slot] == cnd.dist
                            yy = np.array(cnd.y[:m], np.int32)
                            assert np.array_equal(c["y"][slot, blk, a:b], yy)
                            assert c["moment"][blk, band, slot] == int((np.arange(m) * np.abs(yy)).sum())
                            if cnd.dist <= best_cost:
                                best_cost, best_qg, best_y = cnd.dist, cnd.gain, yy
                    assert qg[blk, band] == best_qg
                    if best_qg:
                        gexp = o.odo_gain_expand(best_qg << 8, qb[band], bb[band])
                        syn = np.zeros(m, np.int32)
                        o.odo_pvq_synthesis_partial(P(syn), P(best_y), None, m, 1, gexp, 0, 0, 1,
                                                    P(qi))
                        outvec[a:b] = syn
                blkout = np.zeros((n, n), np.int32)
                o.odo_coding_order_to_raster(P(blkout), n, P(outvec), n)
                want_dq[by * n:(by + 1) * n, bx * n:(bx + 1) * n] = blkout
        assert nsearched > 0
        assert np.array_equal(dq, want_dq), bs


@pytest.mark.parametrize("dec", [0, 1])
def test_inverse_from_pvq_equals_synth_then_inverse(hip, dec):
    """odhip_inverse_level_pvq (dequantise on load, no dq plane) must give the
    same pixels as select_synth + inverse_level, at every level, with and
    without a host rate table."""
    import torch
    W, H = 256, 192
    planes = synth_frame(W, H, seed=21)
    rng = np.random.RandomState(4)
    src = planes[0] if dec == 0 else planes[1]
    src = np.clip(src.astype(int) + rng.randint(-70, 71, size=src.shape), 0, 255).astype(np.uint8)
    px = _cuda(np.stack([src, src[::-1].copy()]))
    pli = 0 if dec == 0 else 1
    levels = hip.forward_pyramid(px, dec, W, H)
    qt = hip.QuantTables.load()
    for bs in range(5 - dec):
        qm, qmi = qt.qm_slices(pli, bs)
        for use_rate in (False, True):
            job = hip.PvqJob(levels[bs], bs, _cuda(qm), _cuda(qmi), qt.q_band(pli, bs),
                             qt.beta_band(pli, bs), dq=torch.empty_like(levels[bs]))
            if use_rate:
                job.rate = torch.from_numpy(rng.uniform(0, 60, size=tuple(job.cands["band"].shape[:2]) + (2,))).cuda()
            hip.pvq_noref_bands_multi([job], hip.OD_PVQ_LAMBDA)
            hip.pvq_select_synth_noref_multi([job], hip.OD_PVQ_LAMBDA)
            want = hip.inverse_level(job.dq, dec, bs, W, H)
            qg_a = job.cands["choice"][..., 1].clone()
            hip.pvq_choose_multi([job], hip.OD_PVQ_LAMBDA)
            assert torch.equal(job.cands["choice"][..., 1], qg_a)
            got = hip.inverse_level_pvq(job, dec, W, H)
            assert torch.equal(got, want), (dec, bs, use_rate)
            if use_rate:
                # the rate table must be able to change choices (cost = dist + lambda*rate)
                job.rate = None
                hip.pvq_choose_multi([job], hip.OD_PVQ_LAMBDA)
                assert not torch.equal(job.cands["choice"][..., 1], qg_a) or bs == 4


@pytest.mark.parametrize("dec", [0, 1])
def test_inverse_levels_in_one_launch_equal_level_by_level(hip, dec):
    import torch
    W, H = 256, 192
    planes = synth_frame(W, H, seed=27)
    rng = np.random.RandomState(5)
    src = planes[0] if dec == 0 else planes[1]
    src = np.clip(src.astype(int) + rng.randint(-70, 71, size=src.shape), 0, 255).astype(np.uint8)
    px = _cuda(np.stack([src, src[::-1].copy(), src[:, ::-1].copy()]))
    pli = 0 if dec == 0 else 1
    levels = hip.forward_pyramid(px, dec, W, H)
    qt = hip.QuantTables.load()
    jobs = []
    for bs in range(5 - dec):
        qm, qmi = qt.qm_slices(pli, bs)
        jobs.append(hip.PvqJob(levels[bs], bs, _cuda(qm), _cuda(qmi), qt.q_band(pli, bs),
                               qt.beta_band(pli, bs)))
    hip.pvq_noref_bands_multi(jobs, hip.OD_PVQ_LAMBDA)
    hip.pvq_choose_multi(jobs, hip.OD_PVQ_LAMBDA)
    want = [hip.inverse_level_pvq(j, dec, W, H) for j in jobs]
    got = hip.inverse_levels_pvq(jobs, dec, W, H)
    for bs, (g, w_) in enumerate(zip(got, want)):
        assert torch.equal(g, w_), (dec, bs)
    # a subset of levels, in a different order
    got2 = hip.inverse_levels_pvq([jobs[2], jobs[0]], dec, W, H)
    assert torch.equal(got2[0], want[2]) and torch.equal(got2[1], want[0])


@pytest.mark.parametrize("dec,size", [(0, (256, 192)), (1, (256, 192)), (0, (64, 128)), (1, (128, 64))])
def test_priced_choice_inside_the_search_equals_the_choice_kernel(hip, dec, size):
    """odhip_pvq_noref_bands_priced_multi (the search kernels decide from their registers) and
    odhip_pvq_noref_bands_multi + odhip_pvq_choose_priced_multi (a choice kernel reads the
    records) leave identical choice records and identical pulses of every chosen candidate (the
    fused stage does not write a losing second candidate or the second half of a record unless
    the decision was a close call); and with the decision margin forced wide - every band a
    close call - the host-libm resolve re-decides bands to the same result, byte for byte."""
    import torch
    W, H = size              # the small ones: fewer blocks than a wavefront has lanes at the upper levels
    big = W*H > 20000
    planes = synth_frame(W, H, seed=29)
    rng = np.random.RandomState(6)
    src = planes[0] if dec == 0 else planes[1]
    src = np.clip(src.astype(int) + rng.randint(-70, 71, size=src.shape), 0, 255).astype(np.uint8)
    px = _cuda(np.stack([src, src[::-1].copy()]))
    pli = 0 if dec == 0 else 1
    levels = hip.forward_pyramid(px, dec, W, H)
    qt = hip.QuantTables.load()
    lam = hip.OD_PVQ_LAMBDA

    def jobs():
        out = []
        for bs in range(5 - dec):
            qm, qmi = qt.qm_slices(pli, bs)
            out.append(hip.PvqJob(levels[bs], bs, _cuda(qm), _cuda(qmi), qt.q_band(pli, bs),
                                  qt.beta_band(pli, bs)))
        return out

    a, b, c = jobs(), jobs(), jobs()
    hip.pvq_noref_bands_multi(a, lam)
    assert hip.pvq_choose_priced_multi(a, lam) == 0
    assert hip.pvq_choose_priced_multi(b, lam, fused_bands=True) == 0
    hip.set_price_tol_scale(1e12)
    try:
        redone = hip.pvq_choose_priced_multi(c, lam, fused_bands=True)
    finally:
        hip.set_price_tol_scale(1.)
    assert redone > (100 if big else 10)
    torch.cuda.synchronize()
    nonzero = 0
    second = 0
    for ja, jb, jc in zip(a, b, c):
        # every decision listed as a close call (c): everything is written, as by the unfused stage
        for key in ("choice", "band", "y"):
            assert torch.equal(ja.cands[key], jc.cands[key]), (ja.bs, key, "resolved")
        # decided in the search (b): the choice and the chosen candidate's pulses; what nobody
        # reads (a losing candidate, the band records: every band is decided by the lanes that
        # prepared it) is not written
        assert torch.equal(ja.cands["choice"], jb.cands["choice"]), (ja.bs, "choice")
        nb, offs, ln = hip.pvq_band_layout(ja.bs)
        band_of = torch.zeros(ln, dtype=torch.long, device="cuda")
        for bnd in range(nb):
            band_of[offs[bnd]:offs[bnd + 1]] = bnd
        ch = ja.cands["choice"]
        for slot in (0, 1):
            picked = (ch[:, :, 0] == slot) & (ch[:, :, 1] != 0)     # [B][nb]
            per_coef = picked[:, band_of]                             # [B][len]
            per_coef[:, 0] = False                                    # the DC slot belongs to no band
            assert torch.equal(ja.cands["y"][slot][per_coef], jb.cands["y"][slot][per_coef]), (ja.bs, slot)
            if slot:
                second += int(picked.sum())

        nonzero += int((ch.view(-1, 4)[:, 1] != 0).sum())
    assert nonzero > (1000 if big else 100) and second > (100 if big else 0)
    # pricing really changes decisions: the distortion-only choice differs somewhere
    d = jobs()
    hip.pvq_noref_bands_multi(d, lam)
    hip.pvq_choose_multi(d, lam)
    assert any(not torch.equal(ja.cands["choice"], jd.cands["choice"]) for ja, jd in zip(a, d))


def test_pair_search_sequential_combine_equals_exact_combine(hip, monkeypatch):
    """The 128-coefficient band is searched by two lanes per band; the halves of
    the greedy argmax are combined by an exact-arithmetic argument, with a
    literal sequential rescan when its bound does not hold.
    ODHIP_PVQ_FORCE_SEQ=1 forces the rescan: both must give the same records
    and pulses (the default path is checked against the oracle above)."""
    import torch
    W, H = 256, 192
    planes = synth_frame(W, H, seed=33)
    rng = np.random.RandomState(9)
    src = np.clip(planes[0].astype(int) + rng.randint(-90, 91, size=planes[0].shape), 0, 255)
    px = _cuda(src.astype(np.uint8)[None])
    levels = hip.forward_pyramid(px, 0, W, H)
    qt = hip.QuantTables.load()
    for bs in (2, 3, 4):
        qm, _ = qt.qm_slices(0, bs)
        outs = []
        for force in ("0", "1"):
            monkeypatch.setenv("ODHIP_PVQ_FORCE_SEQ", force)
            job = hip.PvqJob(levels[bs], bs, _cuda(qm), None, qt.q_band(0, bs), qt.beta_band(0, bs))
            hip.pvq_noref_bands_multi([job], hip.OD_PVQ_LAMBDA)
            torch.cuda.synchronize()
            outs.append((job.cands["band"].clone(), job.cands["y"].clone()))
        assert torch.equal(outs[0][0], outs[1][0]), bs
        assert torch.equal(outs[0][1], outs[1][1]), bs
        k = hip.unpack_cands({"band": outs[0][0], "y": outs[0][1], "choice": outs[0][1]})["k"]
        assert k.max() > 8   # multi-pulse searches did run


def test_pulse_count_above_layout_limit_is_reported_not_searched(hip):
    """A band whose K exceeds ODHIP_PVQ_MAX_K (32767; only reachable with absurd
    quantisers) must come back as flags == 2 with k saturated, must not be
    searched (no hang: K pulses x n candidates would run for minutes) and must
    never be chosen."""
    import torch
    rng = np.random.RandomState(3)
    coef = (rng.randint(-(1 << 19), 1 << 19, size=(1, 64, 64))).astype(np.int32)
    qt = hip.QuantTables.load()
    qm, qmi = qt.qm_slices(0, 0)
    tc = _cuda(coef)
    job = hip.PvqJob(tc, 0, _cuda(qm), _cuda(qmi), [1], [4096], dq=torch.empty_like(tc))
    hip.pvq_noref_bands_multi([job], hip.OD_PVQ_LAMBDA)
    hip.pvq_choose_multi([job], hip.OD_PVQ_LAMBDA)
    torch.cuda.synchronize()
    c = hip.unpack_cands(job.cands)
    over = c["flags"] == 2
    assert over.any()
    assert (c["k"][over] == 32767).all()
    qg = c["choice"][..., 1]
    for slot in range(2):
        bad = over[..., slot] & (qg == c["gain"][..., slot]) & (qg != 0)
        assert not bad.any()
    assert (c["yy"][over] == 0).all()
