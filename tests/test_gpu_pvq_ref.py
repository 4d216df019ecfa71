"""GPU parity of the with-reference (theta / Householder) building blocks
against the CPU oracle: odhip_pvq_ref_prepare, odhip_pvq_ref_candidates,
odhip_pvq_synthesis (SURVEY.md 8 rows a18 with-reference part, a22, a23, a24)."""
import ctypes

import numpy as np
import pytest

from _libs import P, oracle
from test_gpu_pvq_bands import Trace, _cuda, cd, hip  # noqa: F401

pytestmark = pytest.mark.gpu


def _bands(rng, nbands, n, corr_mix):
    """Band vectors with the magnitudes of real coefficient bands (scale 2^4) and
    references of varying correlation, including zero references."""
    amp = rng.choice([40, 400, 4000, 40000], size=(nbands, 1))
    x = (rng.laplace(size=(nbands, n)) * amp).astype(np.int64)
    noise = (rng.laplace(size=(nbands, n)) * amp * corr_mix).astype(np.int64)
    sign = rng.choice([1, 1, 1, -1], size=(nbands, 1))
    r = sign * x + noise
    r[rng.rand(nbands) < 0.05] = 0
    return np.clip(x, -(1 << 22), 1 << 22).astype(np.int32), np.clip(r, -(1 << 22), 1 << 22).astype(np.int32)


@pytest.mark.parametrize("n", [8, 15, 32, 128])
@pytest.mark.parametrize("cfl,beta", [(0, 4096), (1, 4096), (0, 6144)])
def test_ref_prepare_and_candidates_match_oracle(hip, n, cfl, beta):
    import torch
    o = oracle()
    rng = np.random.RandomState(100 + n + cfl)
    nbands = 300
    x0, r0 = _bands(rng, nbands, n, rng.choice([0.05, 0.3, 1.0, 3.0], size=(nbands, 1)))
    qt = hip.QuantTables.load()
    qm_full, qmi_full = qt.qm_slices(1 if cfl else 0, 2)
    off = {8: 16, 15: 1, 32: 32, 128: 128}[n]
    qm = np.ascontiguousarray(qm_full[off:off + n])
    qmi = np.ascontiguousarray(qmi_full[off:off + n])
    q0 = 37
    x16, r16, xr, prep = hip.pvq_ref_prepare(_cuda(x0), _cuda(r0), _cuda(qm), q0, beta, cfl)
    torch.cuda.synchronize()
    rec = prep.cpu().numpy().view(hip.REFPREP_RECORD)[:, 0]
    x16 = x16.cpu().numpy()
    r16 = r16.cpu().numpy()
    xr = xr.cpu().numpy()
    traces = []
    ran = 0
    for b in range(nbands):
        out = np.zeros(n, np.int32)
        y = np.zeros(n, np.int32)
        i1, i2, i3 = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        sd = cd(0)
        tr = Trace()
        # is_keyframe = 1, pli = 1 -> cfl_enabled; pli = 0 with is_keyframe = 0 otherwise
        is_key = 1 if cfl else 0
        pli = 1 if cfl else 0
        o.odo_pvq_theta(P(out), P(x0[b]), P(r0[b]), n, q0, P(y), ctypes.byref(i1), ctypes.byref(i2),
                        ctypes.byref(i3), beta, ctypes.byref(sd), 1, is_key, pli, P(qm), P(qmi),
                        cd(hip.OD_PVQ_LAMBDA), 1, ctypes.byref(tr))
        traces.append(tr)
        for f in ("xshift", "rshift", "g", "gr", "cg", "cgr", "icgr", "gain_offset", "m", "s"):
            assert rec[f][b] == getattr(tr, f), (b, f)
        assert rec["corr"][b] == tr.corr, b
        assert rec["r_null"][b] == int(not r0[b].any())
        assert np.array_equal(x16[b], np.array(tr.x16[:n], np.int16))
        assert np.array_equal(r16[b], np.array(tr.r16[:n], np.int16))
        if not rec["r_null"][b] and tr.corr > 0:
            ran += 1
            want = np.zeros(n, np.int16)
            o.odo_apply_householder(P(want), P(np.array(tr.x16[:n], np.int16)),
                                    P(np.array(tr.r16[:n], np.int16)), n)
            want = np.delete(want, tr.m)
            assert np.array_equal(xr[b], want), b
    assert ran > nbands // 3
    # candidates: theta from the host's acos, as the integration prescribes
    theta = np.floor(.5 + (32768 * 2. / np.pi) * np.arccos(rec["corr"])).astype(np.int32)
    items, nitems = hip.pvq_ref_candidates(prep, _cuda(theta), n, beta)
    torch.cuda.synchronize()
    items = items.cpu().numpy().view(hip.REFCAND_RECORD)[..., 0]
    nitems = nitems.cpu().numpy()
    checked = 0
    for b in range(nbands):
        tr = traces[b]
        want = [tr.cands[i] for i in range(tr.ncands) if tr.cands[i].with_ref]
        if tr.ncands == 24:
            continue  # trace capacity reached: the no-reference candidates displaced nothing, skip
        assert nitems[b] == len(want), b
        for i, c in enumerate(want):
            for f in ("gain", "theta", "ts", "k", "qcg", "qtheta"):
                assert items[f][b, i] == getattr(c, f), (b, i, f)
            checked += 1
    assert checked > nbands


@pytest.mark.parametrize("n", [8, 15, 32, 128])
def test_synthesis_matches_oracle(hip, n):
    import torch
    o = oracle()
    rng = np.random.RandomState(7 + n)
    nbands = 400
    qt = hip.QuantTables.load()
    _, qmi_full = qt.qm_slices(0, 3)
    off = {8: 16, 15: 1, 32: 32, 128: 128}[n]
    qmi = np.ascontiguousarray(qmi_full[off:off + n])
    y = np.zeros((nbands, n), np.int32)
    for b in range(nbands):
        k = int(rng.choice([0, 1, 2, 5, 17, 60]))
        for _ in range(k):
            y[b, rng.randint(n - 1)] += rng.choice([-1, 1])
    # reference vectors as pvq_theta produces them: the VECTOR norm fits 14..15 bits
    # (rshift = od_vector_log_mag - 14), so sum r^2 stays below 2^31
    v = rng.laplace(size=(nbands, n))
    v /= np.sqrt((v * v).sum(axis=1, keepdims=True))
    r16 = np.round(v * rng.choice([300, 5000, 16000, 30000], size=(nbands, 1))).astype(np.int16)
    params = np.zeros((nbands, 5), np.int32)
    params[:, 0] = rng.randint(0, 2, size=nbands)                      # noref
    params[:, 1] = rng.choice([0, 300, 5000, 70000, 900000], size=nbands)  # g
    params[:, 2] = rng.randint(0, 32769, size=nbands)                  # theta, Q15 angle
    params[:, 3] = rng.randint(0, n, size=nbands)                      # m
    params[:, 4] = rng.choice([-1, 1], size=nbands)                    # s
    got = hip.pvq_synthesis(_cuda(y), _cuda(r16), _cuda(params), _cuda(qmi))
    torch.cuda.synchronize()
    got = got.cpu().numpy()
    for b in range(nbands):
        want = np.zeros(n, np.int32)
        noref, g, theta, m, s = (int(v) for v in params[b])
        o.odo_pvq_synthesis_partial(P(want), P(y[b]), P(r16[b]), n, noref, g, theta, m, s, P(qmi))
        assert np.array_equal(got[b], want), (b, params[b])


@pytest.mark.parametrize("n,cfl", [(15, 0), (32, 1), (8, 0)])
def test_with_reference_candidates_end_to_end(hip, n, cfl):
    """The integration path of the with-reference search, as INTEGRATION.md lays it
    out: prepare (GPU) -> acos (host) -> candidate list (GPU) -> per candidate the
    reference's pruning test and K-pulse search on the reflected vector with
    prev_k continuation (GPU, odhip_pvq_search_batch) -> distortion.  Every
    candidate's searched flag, pulse vector, cosine and distortion must equal the
    oracle's trace of pvq_theta."""
    import torch
    o = oracle()
    rng = np.random.RandomState(900 + n)
    nbands = 60
    beta = 4096
    q0 = 23
    lam = hip.OD_PVQ_LAMBDA
    x0, r0 = _bands(rng, nbands, n, rng.choice([0.1, 0.4, 1.0], size=(nbands, 1)))
    qt = hip.QuantTables.load()
    qm_full, qmi_full = qt.qm_slices(1 if cfl else 0, 2)
    off = {8: 16, 15: 1, 32: 32}[n]
    qm = np.ascontiguousarray(qm_full[off:off + n])
    qmi = np.ascontiguousarray(qmi_full[off:off + n])
    x16, r16, xr, prep = hip.pvq_ref_prepare(_cuda(x0), _cuda(r0), _cuda(qm), q0, beta, cfl)
    rec = prep.cpu().numpy().view(hip.REFPREP_RECORD)[:, 0]
    theta = np.floor(.5 + (32768 * 2. / np.pi) * np.arccos(rec["corr"])).astype(np.int32)
    items, nitems = hip.pvq_ref_candidates(prep, _cuda(theta), n, beta)
    items = items.cpu().numpy().view(hip.REFCAND_RECORD)[..., 0]
    nitems = nitems.cpu().numpy()
    s2 = (1. / 256) * (1. / 256)
    nsearched = 0
    for b in range(nbands):
        out = np.zeros(n, np.int32)
        y = np.zeros(n, np.int32)
        i1, i2, i3 = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        sd = cd(0)
        tr = Trace()
        o.odo_pvq_theta(P(out), P(x0[b]), P(r0[b]), n, q0, P(y), ctypes.byref(i1), ctypes.byref(i2),
                        ctypes.byref(i3), beta, ctypes.byref(sd), 1, 1 if cfl else 0, 1 if cfl else 0,
                        P(qm), P(qmi), cd(lam), 1, ctypes.byref(tr))
        want = [tr.cands[i] for i in range(tr.ncands) if tr.cands[i].with_ref]
        if tr.ncands == 24:
            continue
        assert nitems[b] == len(want)
        cg = int(rec["cg"][b])
        prev_k = 0
        y_tmp = torch.zeros((1, n - 1), dtype=torch.int32, device="cuda")
        cos_dist = 0.0
        xb = xr[b:b + 1].contiguous()
        for i in range(nitems[b]):
            qcg, qtheta, k = int(items["qcg"][b, i]), int(items["qtheta"][b, i]), int(items["k"][b, i])
            th = int(theta[b])
            dist_theta = 2 - 2. * o.odo_pvq_cos(th - qtheta) * (1. / 32768)
            dist = (1.4 * (qcg - cg)) * (qcg - cg) + (qcg * float(cg)) * dist_theta
            dist *= s2
            c = want[i]
            if dist > tr.dist0 + 1.0 * lam and k != 0:
                assert not c.searched
                continue
            sin_prod = ((o.odo_pvq_sin(th) * (1. / 32768)) * o.odo_pvq_sin(qtheta)) * (1. / 32768)
            if k == 0:
                cos_dist = 0.0
                y_tmp.zero_()
            elif k != prev_k:
                g2 = np.array([((qcg * float(cg)) * sin_prod) * s2], np.float64)
                y_tmp, cosv = hip.pvq_search_batch(xb, _cuda(np.array([k], np.int32)), _cuda(g2), lam,
                                                   prev_k=_cuda(np.array([prev_k], np.int32)), y=y_tmp)
                cos_dist = float(cosv.cpu().numpy()[0])
                nsearched += 1
            prev_k = k
            dist_theta = 2 - 2. * o.odo_pvq_cos(th - qtheta) * (1. / 32768) + sin_prod * (2 - 2 * cos_dist)
            dist = (1.4 * (qcg - cg)) * (qcg - cg) + (qcg * float(cg)) * dist_theta
            dist *= s2
            assert c.searched
            assert c.cos_dist == cos_dist, (b, i)
            assert c.dist == dist, (b, i)
            assert np.array_equal(y_tmp.cpu().numpy()[0], np.array(c.y[:n - 1], np.int32)), (b, i)
    assert nsearched > nbands // 2
