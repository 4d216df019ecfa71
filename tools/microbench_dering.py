"""Deringing filter on 1080p frames: GPU throughput of
odhip_dering_planes for the five non-zero levels of the encoder's search, with
the reference's own od_dering timed beside it on one host core."""
import ctypes
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import daala_amd as D  # noqa: E402
from _libs import P, ref  # noqa: E402
from make_golden_dering import dering_input  # noqa: E402

D.init(0)
F = 8
nhsb, nvsb = 30, 17
H, W = nvsb * 64, nhsb * 64
base = 37  # pow(quantizer 73, 0.84182), src/encode.c:2694
gains = (0.5, 0.707, 1, 1.41, 2)
x_np = np.stack([dering_input(H, W, 100 + f) for f in range(F)])
rng = np.random.RandomState(0)
bskip_np = (rng.rand(F, nvsb * 16, nhsb * 16) < 0.25).astype(np.uint8)
x = torch.from_numpy(x_np).cuda()
bskip = torch.from_numpy(bskip_np).cuda()
thr = torch.tensor([int(g * base) for g in gains], dtype=torch.int32, device="cuda")
thr = thr[None, :, None].expand(F, 5, nhsb * nvsb).contiguous()
dirs = torch.zeros((F, nvsb * 8, nhsb * 8), dtype=torch.int32, device="cuda")


def timeit(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True)
    b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


t = timeit(lambda: D.dering_planes(x, 0, dirs, 0, bskip, thr))
px = F * H * W
byts = px * 2 * (1 + 5)
print("luma, %d frames x 5 levels: %.1f us  %.2f G superblock-levels/s  %.0f GB/s algorithmic "
      "(2 B read + 5 x 2 B written per sample) = %.1f%% of 8 TB/s" % (
          F, t * 1e6, F * nhsb * nvsb * 5 / t / 1e9, byts / t / 1e9, 100 * byts / t / 8e12))
xc = torch.from_numpy(np.stack([dering_input(H // 2, W // 2, 300 + f) for f in range(2 * F)])).cuda()
bc = torch.from_numpy((rng.rand(2 * F, nvsb * 8, nhsb * 8) < 0.25).astype(np.uint8)).cuda()
dc = dirs.repeat_interleave(2, dim=0).contiguous()
thc = torch.full((2 * F, 1, nhsb * nvsb), int(base * 0.6), dtype=torch.int32, device="cuda")
tc = timeit(lambda: D.dering_planes(xc, 1, dc, 1, bc, thc))
print("chroma, %d planes x 1 level: %.1f us" % (2 * F, tc * 1e6))
r = ref()
if r is not None:
    y = np.zeros((64, 64), np.int16)
    d = (ctypes.c_int * 64)()
    t0 = time.perf_counter()
    n = 0
    for f in range(1):
        for sby in range(nvsb):
            for sbx in range(nhsb):
                xp = ctypes.c_void_p(x_np[f].ctypes.data + 2 * (sby * 64 * W + sbx * 64))
                bp = ctypes.c_void_p(bskip_np[f].ctypes.data + (sby * 16) * (nhsb * 16) + sbx * 16)
                for g in gains:
                    r.ref_dering(P(y), 64, xp, W, 8, 8, sbx, sby, nhsb, nvsb, 0, d, 0, bp, nhsb * 16,
                                 int(g * base), 1, 4)
                    n += 1
    dt = time.perf_counter() - t0
    print("reference od_dering, one core: %.3f s for %d superblock-levels of one frame = %.3g per s; "
          "GPU/CPU = %.0fx" % (dt, n, n / dt, (F * nhsb * nvsb * 5 / t) / (n / dt)))
