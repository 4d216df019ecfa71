"""od_compute_dist on 1080p frames: the batched GPU form
(odhip_dist_parts on coefficient plane pairs + the host's pow finish) for every block of a
level, the per-call surface od_compute_dist_hip (one block, host pointers, synchronous), and
the reference's own od_compute_dist on one host core (oracle/_ref/libdaalaref_dist.so)."""
import ctypes
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import daala_amd as D  # noqa: E402
import daala_amd.api as A  # noqa: E402
from _libs import P  # noqa: E402

D.init(0)
F = 8
H, W = 1088, 1920
rng = np.random.RandomState(3)
x_np = (rng.laplace(size=(F, H, W)) * 300).astype(np.int32)
y_np = (x_np + rng.laplace(size=(F, H, W)) * 40).astype(np.int32)
x = torch.from_numpy(x_np).cuda()
y = torch.from_numpy(y_np).cuda()
L = D.lib()
L.od_compute_dist_hip.restype = ctypes.c_double
ref_so = os.path.join(ROOT, "oracle", "_ref", "libdaalaref_dist.so")
r = ctypes.CDLL(ref_so) if os.path.exists(ref_so) else None
if r is not None:
    r.ref_compute_dist.restype = ctypes.c_double
print("od_compute_dist, %d luma planes of %dx%d (HVS matrices, activity masking on, coded quantiser 40)" % (F, W, H))
for bs in (1, 2, 3, 4):
    n = 4 << bs
    nblk = F * (H // n) * (W // n)
    parts = torch.empty((F, H // 8, W // 8, 3), dtype=torch.float64, device="cuda")

    def dev():
        A._check(L.odhip_dist_parts(A._p(parts), A._p(x), A._p(y), F, W, H, bs, 1, 0, None), "odhip_dist_parts")
    for _ in range(2):
        dev()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True)
    b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        dev()
    b.record()
    torch.cuda.synchronize()
    t_dev = a.elapsed_time(b) / 10 * 1e-3
    t0 = time.perf_counter()
    dist, _ = D.compute_dist(x, y, bs, 1, 0, 40)
    t_all = time.perf_counter() - t0
    # per-call surface and the reference, on a sample of blocks
    m = 300
    xb = [np.ascontiguousarray(x_np[0, i * n:(i + 1) * n, :n]) for i in range(min(m, H // n))]
    yb = [np.ascontiguousarray(y_np[0, i * n:(i + 1) * n, :n]) for i in range(min(m, H // n))]
    for xx, yy in zip(xb[:3], yb[:3]):
        L.od_compute_dist_hip(P(xx), P(yy), n, 1, 0, 40)
    t0 = time.perf_counter()
    got = [L.od_compute_dist_hip(P(xx), P(yy), n, 1, 0, 40) for xx, yy in zip(xb, yb)]
    t_call = (time.perf_counter() - t0) / len(xb)
    line = ("%2dx%-2d: %8d blocks  device parts %.3f ms (%.1f G blocks/s, %.0f GB/s of coefficients read)  "
            "device + D2H + host pow finish %.1f ms  per-call od_compute_dist_hip %.1f us"
            % (n, n, nblk, t_dev * 1e3, nblk / t_dev / 1e9, 2 * x_np.nbytes / t_dev / 1e9, t_all * 1e3, t_call * 1e6))
    if r is not None:
        t0 = time.perf_counter()
        want = [r.ref_compute_dist(P(xx), P(yy), n, 0, 1, 40) for xx, yy in zip(xb, yb)]
        t_ref = (time.perf_counter() - t0) / len(xb)
        assert np.array_equal(np.array(got).view(np.int64), np.array(want).view(np.int64))
        assert np.array_equal(dist[0, :len(xb), 0].view(np.int64), np.array(want).view(np.int64))
        line += "  reference C (1 core) %.1f us per block = %.1f ms per %d blocks: %.0fx batched" % (
            t_ref * 1e6, t_ref * nblk * 1e3, nblk, t_ref * nblk / t_all)
    print(line)
