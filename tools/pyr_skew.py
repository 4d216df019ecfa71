#!/usr/bin/env python3
"""Development (DESIGN.md par. 4): the luma pyramid's time against the address distance between its five
output planes.  One allocation, plane p at p*(128 MiB + skew): 16 frames of 1080p luma, us per launch."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import daala_amd as D  # noqa: E402

D.init(0)
g = torch.Generator(device="cuda").manual_seed(7)
F = 16
luma = torch.randint(0, 256, (F, 1088, 1920), dtype=torch.uint8, device="cuda", generator=g)
n = F * 1088 * 1920
for skew in (0, 256, 1024, 4096, 16384, 65536, 262144, 524288, 1 << 20, (1 << 20) + 4096, 3 << 19):
    stride = (128 << 20) + skew
    big = torch.empty(5 * stride + 4096, dtype=torch.uint8, device="cuda")
    base = (-big.data_ptr()) % 4096
    lv = [big[base + p * stride: base + p * stride + 4 * n].view(torch.int32).view(F, 1088, 1920) for p in range(5)]
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
    print("plane stride 128 MiB + %8d B: %7.1f us" % (skew, a.elapsed_time(b) / 20 * 1e3), flush=True)
    del lv, big
