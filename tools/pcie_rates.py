"""PCIe-inclusive rates of the host-pointer surfaces: the
frame cache (one batched pyramid per plane, host in / host out) and the per-call
transforms.  Development aid, run on the GPU box."""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import daala_amd as D  # noqa: E402

D.init(0)
L = D.lib()
L.odhip_cache_create.restype = ctypes.c_void_p
c = ctypes.c_void_p(L.odhip_cache_create())
L.odhip_cache_set_picture(c, 1920, 1080)
y, cb, cr = bench.synth_frame_np(0, 1)
planes = [((p.astype(np.int32) - 128) << 4) for p in (y, cb, cr)]
planes = [np.ascontiguousarray(p) for p in planes]


def load():
    for pli, p in enumerate(planes):
        h, w = p.shape
        rc = L.odhip_cache_load_plane(c, pli, p.ctypes.data_as(ctypes.c_void_p), w, w, h, 1 if pli else 0)
        assert rc == 0, rc


load()
t0 = time.perf_counter()
N = 10
for _ in range(N):
    load()
dt = (time.perf_counter() - t0) / N
bpf = bench.blocks_per_frame()
print("frame cache: %.2f ms per 1080p 4:2:0 frame host->pyramid->host (%.0f frames/s, %.3g blocks/s, "
      "%.2f GB/s of coefficients returned)" % (dt * 1e3, 1 / dt, bpf / dt,
                                                (5 * y.size + 4 * 2 * cb.size) * 4 / dt / 1e9))
blk = np.ascontiguousarray(planes[0][:8, :8])
out = np.zeros_like(blk)
for _ in range(20):
    L.od_bin_fdct8x8_hip(out.ctypes.data_as(ctypes.c_void_p), 8, blk.ctypes.data_as(ctypes.c_void_p), 8)
t0 = time.perf_counter()
for _ in range(2000):
    L.od_bin_fdct8x8_hip(out.ctypes.data_as(ctypes.c_void_p), 8, blk.ctypes.data_as(ctypes.c_void_p), 8)
print("per-call od_bin_fdct8x8_hip: %.1f us per block" % ((time.perf_counter() - t0) / 2000 * 1e6))
