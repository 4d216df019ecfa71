"""Development aid: the comparisons of tests/test_gpu_pvq_refbands.py without
stopping at the first mismatch (prints per-field counts).  Run on the GPU box:
python tools/refbands_check.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import daala_amd as hip  # noqa: E402
from _refbands import Mismatch, compare_bands, compare_choice, host_rates, make_planes, oracle_traces  # noqa: E402

hip.init(0)
lam = hip.OD_PVQ_LAMBDA
qt = hip.QuantTables.load()
for is_keyframe, pli in [(1, 1), (0, 0), (0, 1), (1, 0)]:
    rng = np.random.RandomState(41 + 2 * is_keyframe + pli)
    mm = Mismatch()
    jobs, meta = [], []
    for bs in range((3 if pli else 4) + 1):
        dec = 1 if pli else 0
        qm, qmi = qt.qm_slices(dec, bs)
        qb, bb = qt.q_band(pli, bs), qt.beta_band(pli, bs)
        x, r = make_planes(rng, 2, 64, 128, bs)
        c = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
        jobs.append(hip.PvqRefJob(c(x), c(r), bs, c(qm), c(qmi), qb, bb, is_keyframe, pli))
        meta.append((x, r, qm, qmi, qb, bb))
    rerun = hip.pvq_ref_bands_multi(jobs, lam)
    torch.cuda.synchronize()
    trs = []
    for job, (x, r, qm, qmi, qb, bb) in zip(jobs, meta):
        traces, _ = oracle_traces(x, r, job.bs, qm, qmi, qb, bb, is_keyframe, pli, lam)
        trs.append(traces)
        compare_bands(hip, job, traces, mm)
    for job, traces in zip(jobs, trs):
        job.rate = torch.from_numpy(host_rates(job, traces, is_keyframe, pli)).cuda()
    hip.pvq_ref_select_synth_multi(jobs, lam)
    torch.cuda.synchronize()
    for job, traces in zip(jobs, trs):
        compare_choice(job, traces, mm)
    print("== is_keyframe %d pli %d: rerun %d, mismatches %d" % (is_keyframe, pli, rerun, mm.total()))
    print(mm.summary())
