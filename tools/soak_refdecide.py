#!/usr/bin/env python3
"""Soak: the decided with-reference stage (every band chosen inside its search, nothing per candidate
exported) against the exporting stage + device-priced choice kernels over random planes, all four modes
(keyframe chroma, inter luma, inter chroma, keyframe luma with a reference) and quantisers: choice records
and the winners' pulses of every band."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import daala_amd as hip  # noqa: E402
from _refbands import make_planes  # noqa: E402

hip.init(0)
lam = hip.OD_PVQ_LAMBDA
MODES = [(1, 1), (0, 0), (0, 1), (1, 0), (1, 2), (0, 2)]
bands = 0
cu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 24):
    rng = np.random.RandomState(5000 + seed)
    is_keyframe, pli = MODES[seed % len(MODES)]
    quality = [1, 5, 10, 20, 40, 100, 511][seed % 7]
    qt = hip.QuantTables.for_quality(quality, use_masking=(seed // 7) % 2)
    top = 3 if pli else 4
    dec = 1 if pli else 0
    h, w = [(64, 128), (128, 64), (64, 192)][seed % 3]
    ja, je = [], []
    for bs in range(top + 1):
        x, r = make_planes(rng, 2, h, w, bs, zero_ref_frac=[0.05, 0.3][seed % 2])
        qm, qmi = qt.qm_slices(dec, bs)
        for lst in (ja, je):
            lst.append(hip.PvqRefJob(cu(x), cu(r), bs, cu(qm), cu(qmi), qt.q_band(pli, bs), qt.beta_band(pli, bs),
                                     is_keyframe, pli))
    hip.pvq_ref_bands_multi(ja, lam)
    ra = hip.pvq_ref_choose_priced_multi(ja, lam)
    nt, npz = hip.pvq_ref_bands_decided_multi(je, lam)
    torch.cuda.synchronize()
    for a, e in zip(ja, je):
        ca = a.choice.cpu().numpy()
        ce = e.choice.cpu().numpy()
        keep = [i for i in range(16) if i != 9]
        assert np.array_equal(ca[..., keep], ce[..., keep]), (seed, a.bs, "choice")
        ya = a.y.cpu().numpy()
        ye = e.y.cpu().numpy()
        for band in range(a.nb):
            lo, hi = a.offsets[band], a.offsets[band + 1]
            sa = ca[:, band, 9]
            se = ce[:, band, 9]
            assert np.array_equal(sa >= 0, se >= 0), (seed, a.bs, band)
            idx = np.nonzero(sa >= 0)[0]
            last = hi - 1 - (ca[idx, band, 2] == 0)
            for wq in range(lo, hi):
                m = wq <= last
                assert np.array_equal(ya[sa[idx], idx, wq][m], ye[se[idx], idx, wq][m]), (seed, a.bs, band, wq)
        bands += ca.shape[0] * ca.shape[1]
    print("seed %2d keyframe %d pli %d -v %3d %dx%d: equal (theta / price re-runs %d, %d; exporting stage %d)"
          % (seed, is_keyframe, pli, quality, w, h, nt, npz, ra), flush=True)
print("%d bands compared" % bands)
