"""Development: the luma forward pyramid over F = 12..20 frames per launch - microseconds per
launch, per frame, and achieved algorithmic TB/s - to see how much of the gap to the VALU
floor is the last partial round of workgroups (3 workgroups of 2 superblocks per CU x 256 CUs
= 1536 superblocks in flight; 510 superblocks per frame)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import daala_amd as D  # noqa: E402

D.init(0)
g = torch.Generator(device="cuda").manual_seed(7)
for F in (12, 13, 14, 15, 16, 17, 18, 20, 24, 30):
    luma = torch.randint(0, 256, (F, 1088, 1920), dtype=torch.uint8, device="cuda", generator=g)
    lv = D.forward_pyramid(luma, 0, 1920, 1080)
    for _ in range(3):
        D.forward_pyramid(luma, 0, 1920, 1080, levels=lv)
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True)
    b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        D.forward_pyramid(luma, 0, 1920, 1080, levels=lv)
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) / 20 * 1e3
    sbs = F * 510
    print("F %2d: %7.1f us per launch, %6.2f us per frame, %5.2f rounds of 1536 superblocks, %.2f TB/s algorithmic (%.3f of 8)"
          % (F, us, us / F, sbs / 1536.0, F * 1920 * 1088 * 21 / us / 1e6, F * 1920 * 1088 * 21 / us / 1e6 / 8))
    del lv, luma
