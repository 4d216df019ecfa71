"""GPU parity of the with-reference band stage on whole planes
(odhip_pvq_ref_bands_multi, odhip_pvq_ref_resolve,
odhip_pvq_ref_select_synth_multi; SURVEY.md 8 rows a18 with-reference branch,
a22, a23, a24 and the chroma-from-luma flip of a26) against the CPU oracle, band
by band: every record field, the QM-scaled and reflected vectors, the candidate
lists in search order, every candidate's pruning decision, pulse vector, cosine
and distortion, and - with the host pricing every candidate as the reference's
od_pvq_rate does - the choice and the dequantised plane."""
import math

import numpy as np
import pytest

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
                         + [(1, 1, "ODHIP_PVQ_FORCE_SEQ"), (0, 0, "ODHIP_PVQ_REF_LANE")])
def test_ref_band_stage_matches_oracle(hip, is_keyframe, pli, env, monkeypatch):
    """env: ODHIP_PVQ_FORCE_SEQ=1 makes the row-parallel search take the literal
    left-to-right scan for every greedy pulse; ODHIP_PVQ_REF_LANE=1 searches the
    32- and 128-coefficient bands one band per lane instead of one per row."""
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
    finally:
        hip.pvq_ref_set_theta_margin(0, False)
    mm = Mismatch()
    for job, (x, r, qm, qmi, qb, bb) in zip(jobs, meta):
        traces, _ = oracle_traces(x, r, job.bs, qm, qmi, qb, bb, 1, 1, lam)
        compare_bands(hip, job, traces, mm)
    assert mm.total() == 0, mm.summary()


def test_ref_jobs_argument_validation(hip):
    import ctypes
    L = hip.lib()
    assert L.odhip_pvq_ref_bands_multi(None, 1, ctypes.c_double(0.1), None) != 0
    assert L.odhip_pvq_ref_select_synth_multi(None, 0, ctypes.c_double(0.1), None) != 0
