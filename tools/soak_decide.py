#!/usr/bin/env python3
"""Soak: the no-reference stage that decides in place against the two-pass stage + choice kernel over random
pictures, noise levels and quantisers (choice records and the chosen candidates' pulses of every band)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import daala_amd as hip  # noqa: E402
from _libs import synth_frame  # noqa: E402

hip.init(0)
lam = hip.OD_PVQ_LAMBDA
bands = 0
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 24):
    rng = np.random.RandomState(1000 + seed)
    W, H = [(256, 192), (128, 320), (512, 64)][seed % 3]
    dec = seed % 2
    quality = [1, 5, 10, 20, 40, 100, 511][seed % 7]
    qt = hip.QuantTables.for_quality(quality, use_masking=(seed // 7) % 2, hvs_qm=1 - (seed // 14) % 2)
    planes = synth_frame(W, H, seed=seed)
    src = planes[0] if dec == 0 else planes[1]
    amp = [0, 20, 70, 127][seed % 4]
    src = np.clip(src.astype(int) + rng.randint(-amp, amp + 1, size=src.shape), 0, 255).astype(np.uint8)
    px = torch.from_numpy(np.stack([src, src[::-1].copy(), rng.randint(0, 256, size=src.shape).astype(np.uint8)])).cuda()
    pli = 0 if dec == 0 else 1
    levels = hip.forward_pyramid(px, dec, W, H)

    def jobs():
        out = []
        for bs in range(5 - dec):
            qm, qmi = qt.qm_slices(pli, bs)
            out.append(hip.PvqJob(levels[bs], bs, torch.from_numpy(qm).cuda(), torch.from_numpy(qmi).cuda(),
                                  qt.q_band(pli, bs), qt.beta_band(pli, bs)))
        return out
    a, b = jobs(), jobs()
    hip.pvq_noref_bands_multi(a, lam)
    ra = hip.pvq_choose_priced_multi(a, lam)
    rb = hip.pvq_choose_priced_multi(b, lam, fused_bands=True)
    torch.cuda.synchronize()
    for ja, jb in zip(a, b):
        assert torch.equal(ja.cands["choice"], jb.cands["choice"]), (seed, ja.bs, "choice")
        nb, offs, ln = hip.pvq_band_layout(ja.bs)
        band_of = torch.zeros(ln, dtype=torch.long, device="cuda")
        for bnd in range(nb):
            band_of[offs[bnd]:offs[bnd + 1]] = bnd
        ch = ja.cands["choice"]
        for slot in (0, 1):
            picked = (ch[:, :, 0] == slot) & (ch[:, :, 1] != 0)
            per_coef = picked[:, band_of]
            per_coef[:, 0] = False
            assert torch.equal(ja.cands["y"][slot][per_coef], jb.cands["y"][slot][per_coef]), (seed, ja.bs, slot)
        bands += ch.shape[0] * ch.shape[1]
    print("seed %2d %dx%d dec %d -v %d noise %3d: equal (host-libm re-decisions %d / %d)" % (seed, W, H, dec, quality, amp, ra, rb),
          flush=True)
print("%d bands compared" % bands)
