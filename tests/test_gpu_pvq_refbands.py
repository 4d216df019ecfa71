"""GPU parity of the with-reference band stage on whole planes
(odhip_pvq_ref_bands_multi, odhip_pvq_ref_resolve,
odhip_pvq_ref_select_synth_multi; SURVEY.md 8 rows a18 with-reference branch,
a22, a23, a24 and the chroma-from-luma flip of a26) against the CPU oracle, band
by band: every record field, the QM-scaled and reflected vectors, the candidate
lists in search order, every candidate's pruning decision, pulse vector, cosine
and distortion, and - with the host pricing every candidate as the reference's
od_pvq_rate does - the choice and the dequantised plane."""
import math
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from _refbands import (THETA_SCALE, Mismatch, compare_bands, compare_choice, host_rates,
                       make_planes, oracle_traces)

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


def _job(hip, rng, bs, is_keyframe, pli, h=64, w=128, nplanes=2):
    qt = hip.QuantTables.load()
    dec = 1 if pli else 0
    qm, qmi = qt.qm_slices(dec, bs)
    qb, bb = qt.q_band(pli, bs), qt.beta_band(pli, bs)
    x, r = make_planes(rng, nplanes, h, w, bs)
    job = hip.PvqRefJob(_cuda(x), _cuda(r), bs, _cuda(qm), _cuda(qmi), qb, bb, is_keyframe, pli)
    return job, x, r, qm, qmi, qb, bb


MODES = [(1, 1), (0, 0), (0, 1), (1, 0)]   # (is_keyframe, pli): CfL, inter luma/chroma, key luma


@pytest.mark.parametrize("is_keyframe,pli,env", [m + (None,) for m in MODES]
                         + [(1, 1, "ODHIP_PVQ_FORCE_SEQ")])
def test_ref_band_stage_matches_oracle(hip, is_keyframe, pli, env, monkeypatch):
    """env: ODHIP_PVQ_FORCE_SEQ=1 makes the row-parallel search take the literal
    left-to-right scan for every greedy pulse."""
    import torch
    if env:
        monkeypatch.setenv(env, "1")
    lam = hip.OD_PVQ_LAMBDA
    rng = np.random.RandomState(41 + 2 * is_keyframe + pli)
    top = 3 if pli else 4
    jobs, meta = [], []
    for bs in range(top + 1):
        job, x, r, qm, qmi, qb, bb = _job(hip, rng, bs, is_keyframe, pli)
        jobs.append(job)
        meta.append((x, r, qm, qmi, qb, bb))
    rerun = hip.pvq_ref_bands_multi(jobs, lam)
    torch.cuda.synchronize()
    assert rerun == 0
    mm = Mismatch()
    all_traces = []
    for job, (x, r, qm, qmi, qb, bb) in zip(jobs, meta):
        traces, _ = oracle_traces(x, r, job.bs, qm, qmi, qb, bb, is_keyframe, pli, lam)
        all_traces.append(traces)
        compare_bands(hip, job, traces, mm)
    assert mm.total() == 0, mm.summary()
    assert mm.checked.get("item.y", 0) > 1000
    # the host prices every searched candidate, the GPU chooses and synthesises
    for job, traces in zip(jobs, all_traces):
        job.rate = _cuda(host_rates(job, traces, is_keyframe, pli))
    hip.pvq_ref_select_synth_multi(jobs, lam)
    torch.cuda.synchronize()
    for job, traces in zip(jobs, all_traces):
        compare_choice(job, traces, mm)
    assert mm.total() == 0, mm.summary()
    assert mm.checked["dq"] > 1000


def test_ref_band_stage_one_band_per_lane_experiments_build():
    """ODHIP_PVQ_REF_LANE=1 - the 32- and 128-coefficient bands searched one band per lane instead of
    one per quad / row - is a form of rounds 1-2 that only the EXPERIMENTS build of the library still
    holds (-DODHIP_EXPERIMENTS): the inter-luma case of the test above in a child process that loads
    that build with the switch set."""
    import subprocess
    import daala_amd
    env = dict(os.environ)
    env.update(ODHIP_LIB=daala_amd.EXPERIMENTS_LIB, ODHIP_PVQ_REF_LANE="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-m", "gpu", "-k",
                        "test_ref_band_stage_matches_oracle and 0-0-None"], env=env, capture_output=True,
                       text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0 and "1 passed" in r.stdout, r.stdout[-2000:] + r.stderr[-1000:]


def test_device_theta_argument_agrees_with_host_libm(hip):
    """The margin argument: .5 + OD_THETA_SCALE*acos(corr) on the device and with
    the host libm (math.acos) agree far inside the 1e-9 margin, so a band outside
    the margin has the reference's integer theta."""
    import torch
    rng = np.random.RandomState(9)
    corr = np.concatenate([rng.rand(300000), 1 - rng.rand(50000) * 1e-6, rng.rand(50000) * 1e-6,
                           np.array([1.0, 0.5, 2. ** -30])])
    t = hip.pvq_ref_theta_probe(_cuda(corr))
    torch.cuda.synchronize()
    t = t.cpu().numpy()
    want = np.array([.5 + THETA_SCALE * math.acos(c) for c in corr])
    assert np.abs(t - want).max() < 1e-10
    assert np.array_equal(np.floor(t)[np.abs(want - np.rint(want)) > 1e-9],
                          np.floor(want)[np.abs(want - np.rint(want)) > 1e-9])


def test_uncertain_theta_is_settled_by_the_host(hip):
    """With a huge margin and the debug perturbation, the device theta of every
    listed band is deliberately wrong; odhip_pvq_ref_resolve must list them,
    recompute theta with the host libm and re-run them, after which everything
    equals the oracle again."""
    import torch
    lam = hip.OD_PVQ_LAMBDA
    rng = np.random.RandomState(77)
    jobs, meta = [], []
    for bs in (0, 2):
        job, x, r, qm, qmi, qb, bb = _job(hip, rng, bs, 1, 1, h=32, w=64)
        jobs.append(job)
        meta.append((x, r, qm, qmi, qb, bb))
    hip.pvq_ref_set_theta_margin(0.25, True)
    try:
        hip.pvq_ref_bands_multi(jobs, lam, resolve=False)
        torch.cuda.synchronize()
        flagged = sum(int(((j.unpack()["rec"]["flags"] & hip.REFBAND_UNCERTAIN) != 0).sum()) for j in jobs)
        assert flagged > 20
        mm = Mismatch()
        for job, (x, r, qm, qmi, qb, bb) in zip(jobs, meta):
            traces, _ = oracle_traces(x, r, job.bs, qm, qmi, qb, bb, 1, 1, lam)
            compare_bands(hip, job, traces, mm)
        assert mm.counts.get("rec.theta", 0) == flagged      # wrong before the host looked
        rerun = hip.pvq_ref_bands_multi(jobs, lam, resolve=True)
        torch.cuda.synchronize()
        assert rerun == flagged
        # the same through the two-phase form (no host wait inside the stage)
        hip.pvq_ref_bands_multi(jobs, lam, resolve="async")
        assert hip.pvq_ref_resolve_finish(jobs, lam) == flagged
        torch.cuda.synchronize()
    finally:
        hip.pvq_ref_set_theta_margin(0, False)
    mm = Mismatch()
    for job, (x, r, qm, qmi, qb, bb) in zip(jobs, meta):
        traces, _ = oracle_traces(x, r, job.bs, qm, qmi, qb, bb, 1, 1, lam)
        compare_bands(hip, job, traces, mm)
    assert mm.total() == 0, mm.summary()


@pytest.mark.parametrize("is_keyframe,pli", MODES)
def test_priced_choice_on_the_device_equals_host_priced_choice(hip, is_keyframe, pli):
    """od_pvq_rate's closed form on the device - keyframe chroma, inter luma / chroma (theta
    cost, the `qg == icgr` term, the skip rules of inter frames) and keyframe luma with a
    reference - three ways: (a) the host prices every candidate with the oracle's od_pvq_rate
    and the choice kernel takes its rate table (checked against the oracle's pvq_theta),
    (b) the choice kernels price on the device, (c) the per-lane searches decide inside the
    band stage.  All three leave identical choice records."""
    import torch
    lam = hip.OD_PVQ_LAMBDA
    rng = np.random.RandomState(91 + 2 * is_keyframe + pli)
    top = 3 if pli else 4
    sets = []
    planes = [make_planes(rng, 2, 64, 128, bs) for bs in range(top + 1)]
    for _ in range(3):
        jobs = []
        for bs in range(top + 1):
            qt = hip.QuantTables.load()
            dec = 1 if pli else 0
            qm, qmi = qt.qm_slices(dec, bs)
            x, r = planes[bs]
            jobs.append((hip.PvqRefJob(_cuda(x), _cuda(r), bs, _cuda(qm), _cuda(qmi), qt.q_band(pli, bs),
                                       qt.beta_band(pli, bs), is_keyframe, pli), x, r, qm, qmi,
                         qt.q_band(pli, bs), qt.beta_band(pli, bs)))
        sets.append(jobs)
    # (a) host-priced, validated against the oracle
    ja = [j[0] for j in sets[0]]
    hip.pvq_ref_bands_multi(ja, lam)
    torch.cuda.synchronize()
    mm = Mismatch()
    for (job, x, r, qm, qmi, qb, bb) in sets[0]:
        traces, _ = oracle_traces(x, r, job.bs, qm, qmi, qb, bb, is_keyframe, pli, lam)
        job.rate = _cuda(host_rates(job, traces, is_keyframe, pli))
    hip.pvq_ref_select_synth_multi(ja, lam)
    torch.cuda.synchronize()
    for (job, x, r, qm, qmi, qb, bb) in sets[0]:
        traces, _ = oracle_traces(x, r, job.bs, qm, qmi, qb, bb, is_keyframe, pli, lam)
        compare_choice(job, traces, mm)
    assert mm.total() == 0, mm.summary()
    # (b) choice kernels price on the device, (c) decided inside the band stage
    jb = [j[0] for j in sets[1]]
    jc = [j[0] for j in sets[2]]
    hip.pvq_ref_bands_multi(jb, lam)
    assert hip.pvq_ref_choose_priced_multi(jb, lam) == 0
    assert hip.pvq_ref_choose_priced_multi(jc, lam, fused_bands=True) == 0
    torch.cuda.synchronize()
    for a, b, c in zip(ja, jb, jc):
        assert torch.equal(a.choice, b.choice), (is_keyframe, pli, a.bs, "device-priced")
        assert torch.equal(a.choice, c.choice), (is_keyframe, pli, a.bs, "decided in the search")
    # with the margin forced wide the host libm re-decides bands, to the same records
    hip.set_price_tol_scale(1e12)
    try:
        jd = [j[0] for j in sets[1]]
        assert hip.pvq_ref_choose_priced_multi(jd, lam, fused_bands=True) > 50
    finally:
        hip.set_price_tol_scale(1.)
    torch.cuda.synchronize()
    for a, d in zip(ja, jd):
        assert torch.equal(a.choice, d.choice), (is_keyframe, pli, a.bs, "host-libm resolve")
    # (e) the DECIDED stage: every band chosen inside its search, nothing per candidate
    # exported - same choice records (word 9 names the pulse slot: 0 there) and the same
    # winner's pulses; then with the margin forced wide (listed bands re-run by the exporting
    # kernels and re-decided from host rates)
    for scale in (1., 1e12):
        hip.set_price_tol_scale(scale)
        try:
            je = [hip.PvqRefJob(j.coef, j.ref, j.bs, j.qm, j.qm_inv, list(j.q_band)[:j.nb],
                                list(j.beta_band)[:j.nb], is_keyframe, pli) for j in ja]
            nt, npz = hip.pvq_ref_bands_decided_multi(je, lam)
            assert nt == 0 and (npz == 0 if scale == 1. else npz > 50)
        finally:
            hip.set_price_tol_scale(1.)
        torch.cuda.synchronize()
        for a, e in zip(ja, je):
            ca = a.choice.cpu().numpy()
            ce = e.choice.cpu().numpy()
            keep = [i for i in range(16) if i != 9]
            assert np.array_equal(ca[..., keep], ce[..., keep]), (is_keyframe, pli, a.bs, "decided", scale)
            ya = a.y.cpu().numpy()
            ye = e.y.cpu().numpy()
            for band in range(a.nb):
                lo, hi = a.offsets[band], a.offsets[band + 1]
                sa = ca[:, band, 9]
                se = ce[:, band, 9]
                assert np.array_equal(sa >= 0, se >= 0)
                idx = np.nonzero(sa >= 0)[0]
                # a theta winner holds n - 1 pulses (the last position is a pad)
                last = hi - 1 - (ca[idx, band, 2] == 0)
                for w in range(lo, hi):
                    m = w <= last
                    assert np.array_equal(ya[sa[idx], idx, w][m], ye[se[idx], idx, w][m]), \
                        (is_keyframe, pli, a.bs, band, w, scale)


def test_ref_jobs_argument_validation(hip):
    import ctypes
    L = hip.lib()
    assert L.odhip_pvq_ref_bands_multi(None, 1, ctypes.c_double(0.1), None) != 0
    assert L.odhip_pvq_ref_select_synth_multi(None, 0, ctypes.c_double(0.1), None) != 0


def test_chroma_from_luma_planes_match_the_compiled_reference(hip):
    """End to end against the REAL reference (oracle/_ref): luma planes through
    the reference's stage give the dequantised luma coefficients; their
    upper-left quarters are the chroma-from-luma predictions (the non-TF branch of
    od_resample_luma_coeffs, src/intra.c:97-108); the chroma plane then goes
    through pvq_theta with those references on the CPU (ref_stage_plane_cfl) and
    through forward pyramid -> with-reference band stage -> choice with the
    host's rates -> synthesis -> inverse on the GPU.  Dequantised coefficient
    planes of all four levels and the reconstructed pixels must be identical."""
    import ctypes
    import torch
    from _libs import P, ref, synth_frame
    r = ref()
    if r is None:
        pytest.skip("oracle/_ref not present")
    lam = hip.OD_PVQ_LAMBDA
    W, H = 128, 128
    planes = synth_frame(W, H, seed=21)
    rng = np.random.RandomState(5)
    planes = [np.clip(p.astype(int) + rng.randint(-40, 41, size=p.shape), 0, 255).astype(np.uint8)
              for p in planes]
    qt = hip.QuantTables.load()
    r.ref_stage_plane_cfl.restype = ctypes.c_long
    qm_all = np.ascontiguousarray(qt.qm)
    qmi_all = np.ascontiguousarray(qt.qm_inv)

    def tables(p):
        qm_off = (ctypes.c_int * 5)(*[int(qt.qm_offset[bs][p]) for bs in range(5)])
        qb = (ctypes.c_int * 60)()
        bb = (ctypes.c_int * 60)()
        for bs in range(5):
            for i, v in enumerate(qt.q_band(p, bs)):
                qb[bs * 12 + i] = v
            for i, v in enumerate(qt.beta_band(p, bs)):
                bb[bs * 12 + i] = v
        return qm_off, qb, bb

    # luma on the CPU: dequantised planes of the five levels
    luma = planes[0]
    ldq = [np.zeros((H, W), np.int32) for _ in range(5)]
    arr = (ctypes.c_void_p * 5)(*[a.ctypes.data for a in ldq])
    qm_off, qb, bb = tables(0)
    recon = np.zeros_like(luma)
    r.ref_stage_plane_cfl(P(luma), W, W, H, 0, W, H, 0, P(qm_all), P(qmi_all), qm_off, qb, bb,
                          ctypes.c_double(lam), P(recon), arr, None)
    # chroma-from-luma predictions: upper-left quarter of every luma block one size up
    cw, chh = W // 2, H // 2
    refs = []
    for bs in range(4):
        n = 4 << bs
        corner = ldq[bs + 1].reshape(H // (2 * n), 2 * n, W // (2 * n), 2 * n)[:, :n, :, :n]
        refs.append(np.ascontiguousarray(corner.reshape(chh, cw)))
    chroma = planes[1]
    cdq = [np.zeros((chh, cw), np.int32) for _ in range(4)]
    arr_dq = (ctypes.c_void_p * 5)(*([a.ctypes.data for a in cdq] + [None]))
    arr_ref = (ctypes.c_void_p * 5)(*([a.ctypes.data for a in refs] + [None]))
    qm_off, qb, bb = tables(1)
    crecon = np.zeros_like(chroma)
    nblk = r.ref_stage_plane_cfl(P(chroma), cw, cw, chh, 1, W, H, 1, P(qm_all), P(qmi_all), qm_off,
                                 qb, bb, ctypes.c_double(lam), P(crecon), arr_dq, arr_ref)
    assert nblk == sum((cw // (4 << bs)) * (chh // (4 << bs)) for bs in range(4))
    # the same on the GPU
    levels = hip.forward_pyramid(_cuda(chroma[None]), 1, W, H)
    jobs, meta = [], []
    for bs in range(4):
        qm, qmi = qt.qm_slices(1, bs)
        job = hip.PvqRefJob(levels[bs], _cuda(refs[bs][None]), bs, _cuda(qm), _cuda(qmi),
                            qt.q_band(1, bs), qt.beta_band(1, bs), 1, 1)
        jobs.append(job)
        meta.append((qm, qmi))
    hip.pvq_ref_bands_multi(jobs, lam)
    torch.cuda.synchronize()
    flips = 0
    for bs, (job, (qm, qmi)) in enumerate(zip(jobs, meta)):
        traces, _ = oracle_traces(levels[bs].cpu().numpy(), refs[bs][None], bs, qm, qmi,
                                  qt.q_band(1, bs), qt.beta_band(1, bs), 1, 1, lam)
        job.rate = _cuda(host_rates(job, traces, 1, 1))
        flips += sum(t["flip"] for (blk, b), t in traces.items() if b == 0)
    assert flips > 0
    hip.pvq_ref_select_synth_multi(jobs, lam)
    torch.cuda.synchronize()
    for bs, job in enumerate(jobs):
        assert np.array_equal(job.dq[0].cpu().numpy(), cdq[bs]), "dequantised chroma plane, level %d" % bs
        assert np.abs(cdq[bs]).sum() > 0
    got = hip.inverse_level(jobs[3].dq, 1, 3, W, H)[0].cpu().numpy()
    assert np.array_equal(got, crecon), "reconstructed chroma pixels (32x32 level)"


def test_pulse_counts_beyond_the_int16_vectors_are_reported_not_searched(hip):
    """Quantiser 1 on large coefficients asks for K far above ODHIP_PVQ_MAX_K
    (32767): such candidates are flagged ODHIP_REFITEM_K_RANGE, never searched,
    never chosen; everything that is searched still equals the oracle."""
    import torch
    lam = hip.OD_PVQ_LAMBDA
    rng = np.random.RandomState(123)
    qt = hip.QuantTables.load()
    bs = 2
    qm, qmi = qt.qm_slices(0, bs)
    nb = hip.pvq_band_layout(bs)[0]
    x, r = make_planes(rng, 1, 32, 32, bs, zero_ref_frac=0.0)
    x = (x.astype(np.int64) * 64).clip(-(1 << 21), 1 << 21).astype(np.int32)
    r = (r.astype(np.int64) * 64).clip(-(1 << 21), 1 << 21).astype(np.int32)
    job = hip.PvqRefJob(_cuda(x), _cuda(r), bs, _cuda(qm), _cuda(qmi), [1] * nb, [4096] * nb, 0, 0)
    hip.pvq_ref_bands_multi([job], lam)
    hip.pvq_ref_select_synth_multi([job], lam)
    torch.cuda.synchronize()
    u = job.unpack()
    items, rec, ch = u["items"], u["rec"], u["choice"]
    big = 0
    for blk in range(rec.shape[0]):
        for b in range(nb):
            for i in range(int(rec[blk, b]["nitems"])):
                it = items[blk, b, i]
                fl = int(it["flags"])
                if fl & hip.REFITEM_K_RANGE:
                    big += 1
                    assert int(it["k"]) > 32767 and not (fl & hip.REFITEM_SEARCHED)
                    assert int(ch[blk, b, 0]) != i
                elif fl & hip.REFITEM_SEARCHED:
                    assert int(it["k"]) <= 32767
    assert big > 0


def test_cfl_refs_from_luma_equal_the_quarter_of_the_dequantised_luma_planes(hip):
    """odhip_cfl_refs_from_luma (od_resample_luma_coeffs for luma blocks >= 8x8, fed
    by the chosen luma candidates) against the construction from the dequantised
    luma planes that odhip_pvq_select_synth_noref_multi writes (themselves checked
    against the oracle in test_gpu_pvq_bands.py): the upper-left quarter of every
    luma block, for Cb and Cr."""
    import torch
    from _libs import synth_frame
    lam = hip.OD_PVQ_LAMBDA
    W, H = 192, 128
    rng = np.random.RandomState(31)
    px = np.stack([np.clip(synth_frame(W, H, seed=3 + i)[0].astype(int)
                           + rng.randint(-40, 41, size=(H, W)), 0, 255).astype(np.uint8) for i in range(2)])
    qt = hip.QuantTables.load()
    levels = hip.forward_pyramid(_cuda(px), 0, W, H)
    jobs = []
    for bs in range(5):
        qm, qmi = qt.qm_slices(0, bs)
        jobs.append(hip.PvqJob(levels[bs], bs, _cuda(qm), _cuda(qmi), qt.q_band(0, bs),
                               qt.beta_band(0, bs), dq=torch.zeros_like(levels[bs])))
    hip.pvq_noref_bands_multi(jobs, lam)
    hip.pvq_choose_multi(jobs, lam)
    refs = hip.cfl_refs_from_luma(jobs[1:], copies=2)
    hip.pvq_select_synth_noref_multi(jobs, lam)
    torch.cuda.synchronize()
    nonzero = 0
    for bs in range(4):
        n = 4 << bs
        dq = jobs[bs + 1].dq
        corner = dq.view(2, H // (2 * n), 2 * n, W // (2 * n), 2 * n)[:, :, :n, :, :n]
        want = corner.reshape(2, H // 2, W // 2)
        want = torch.cat([want, want], dim=0)
        assert refs[bs].shape == want.shape
        assert torch.equal(refs[bs], want), "chroma level %d" % bs
        nonzero += int((want != 0).sum())
    assert nonzero > 5000
    # 4x4 luma blocks: the TF branch (od_tf_up_hv_lp + OD_CFL_SCALING4) against the oracle's
    # od_resample_luma_coeffs (pinned to the reference) on the dequantised level-0 plane
    from _libs import P, oracle
    o = oracle()
    tf = hip.cfl_refs_from_luma(jobs[:1], copies=2)[0].cpu().numpy()
    dq0 = jobs[0].dq.cpu().numpy()
    tfnz = 0
    for p in range(2):
        for by in range(H // 8):
            for bx in range(W // 8):
                area = np.ascontiguousarray(dq0[p, by * 8:by * 8 + 8, bx * 8:bx * 8 + 8])
                want4 = np.zeros((4, 4), np.int32)
                o.odo_resample_luma_coeffs(P(want4), 4, P(area), 8, 0, 0)
                for cp in (p, p + 2):
                    assert np.array_equal(tf[cp, by * 4:by * 4 + 4, bx * 4:bx * 4 + 4], want4), (p, by, bx)
                tfnz += int((want4 != 0).sum())
    assert tfnz > 2000


def test_two_contexts_in_flight_give_the_sequential_results(hip):
    """Two call sequences (bands -> resolve -> select_synth) on two streams, one per
    library context, interleaved from one host thread: records, candidates, pulses,
    choices and dequantised planes equal those of the same jobs run one after the
    other in context 0."""
    import torch
    lam = hip.OD_PVQ_LAMBDA
    rng = np.random.RandomState(55)

    def make(seed):
        r = np.random.RandomState(seed)
        return [_job(hip, r, bs, 1, 1, h=64, w=128)[0] for bs in (0, 1, 2, 3)]

    seq = [make(1), make(2)]
    for jobs in seq:
        hip.pvq_ref_bands_multi(jobs, lam)
        hip.pvq_ref_select_synth_multi(jobs, lam)
    torch.cuda.synchronize()
    par = [make(1), make(2)]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    try:
        for c in (0, 1):
            hip.pvq_ref_set_context(c)
            with torch.cuda.stream(streams[c]):
                hip.pvq_ref_bands_multi(par[c], lam, resolve="async")
        for c in (1, 0):
            hip.pvq_ref_set_context(c)
            with torch.cuda.stream(streams[c]):
                hip.pvq_ref_select_synth_multi(par[c], lam)
                assert hip.pvq_ref_resolve_finish(par[c], lam) == 0
    finally:
        hip.pvq_ref_set_context(0)
    torch.cuda.synchronize()
    for a, b in zip(seq[0] + seq[1], par[0] + par[1]):
        assert torch.equal(a.band, b.band)
        assert torch.equal(a.choice, b.choice)
        assert torch.equal(a.dq, b.dq)
        ua, ub = a.unpack(), b.unpack()
        nit = ua["rec"]["nitems"]
        for s in range(14):
            valid = s < nit
            for f in ("gain", "theta", "ts", "k"):
                assert (ua["items"][f][:, :, s][valid] == ub["items"][f][:, :, s][valid]).all()
            done = valid & ((ua["items"]["flags"][:, :, s] & 1) != 0)
            for f in ("flags", "yslot", "cos_dist", "dist"):
                assert (ua["items"][f][:, :, s][done] == ub["items"][f][:, :, s][done]).all()


@pytest.mark.parametrize("is_keyframe,pli,dec", [(1, 1, 1), (0, 0, 0), (0, 1, 1)])
def test_inverse_fed_by_the_with_reference_stage_equals_synthesis_plus_inverse(hip, is_keyframe, pli, dec):
    """odhip_pvq_ref_choose_multi + odhip_inverse_levels_pvq_ref (dequantise-on-load:
    with-reference and no-reference synthesis, skip-copy and skip-zero bands, inside
    the inverse kernel's tile load) give exactly the pixels of
    odhip_pvq_ref_select_synth_multi + odhip_inverse_levels on the dequantised planes,
    whose planes are checked against the oracle / the compiled reference above."""
    import torch
    lam = hip.OD_PVQ_LAMBDA
    rng = np.random.RandomState(70 + 2 * is_keyframe + pli)
    top = 4 - dec
    h, w = 64, 128
    jobs = [_job(hip, rng, bs, is_keyframe, pli, h=h, w=w)[0] for bs in range(top + 1)]
    hip.pvq_ref_bands_multi(jobs, lam)
    hip.pvq_ref_select_synth_multi(jobs, lam)
    pic = (2 * w - 6, 2 * h - 10) if dec else (w - 6, h - 10)
    want = hip.inverse_levels([j.dq for j in jobs], dec, list(range(top + 1)), pic[0], pic[1])
    torch.cuda.synchronize()
    modes = set()
    for j in jobs:
        j.choice.zero_()
        j.dq.fill_(12345)          # the fused path must not need it
    hip.pvq_ref_choose_multi(jobs, lam)
    got = hip.inverse_levels_pvq_ref(jobs, dec, pic[0], pic[1])
    torch.cuda.synchronize()
    for bs in range(top + 1):
        assert torch.equal(got[bs], want[bs]), (is_keyframe, pli, bs)
        modes |= set(np.unique(jobs[bs].choice.cpu().numpy()[:, :, 8]).tolist())
        assert int(jobs[bs].dq[0, 0, 1]) == 12345
    # the data exercises with-reference and no-reference synthesis and zero bands
    # (skip-copy bands, mode 1 / 4, only exist on inter frames)
    assert {0, 2, 3} <= modes
    if not is_keyframe:
        assert modes & {1, 4}
