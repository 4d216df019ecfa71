"""Child process of test_gpu_dropin_encoder.py: encode with the reference
encoder while its filter drivers and PVQ search are interposed by libdaalahip.
Prints a JSON line {"packets": hex, "sizes": [...], "calls": [...]}."""
import ctypes
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from _libs import P, synth_frame  # noqa: E402

interpose = int(sys.argv[1])
w = h = 64
nframes = int(os.environ.get("NFRAMES", "2"))
hip = ctypes.CDLL(os.path.join(ROOT, "daala_amd", "lib", "libdaalahip.so"), mode=ctypes.RTLD_GLOBAL)
ipo = None
if interpose:
    assert hip.odhip_init(0) == 0
    ipo = ctypes.CDLL(os.path.join(ROOT, "tests", "interpose", "libinterpose.so"),
                      mode=ctypes.RTLD_GLOBAL)
# REF_LIB=libdaalaref_4x4.so: the reference built with block sizes limited to 4x4 (BASELINE
# configs[0]; oracle/Makefile)
r = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", os.environ.get("REF_LIB", "libdaalaref.so")))
if ipo is not None:
    # pass-through mode (ODHIP_INTERPOSE_PASSTHROUGH=1) calls the reference's own definitions
    ipo.odhip_interpose_set_reference(ctypes.c_void_p(r._handle))
if interpose in (2, 3):
    # frame cache: batched pyramid per plane serves every fdct_2d call; mode 3 adds the
    # batched PVQ band stage of keyframe luma behind pvq_theta (host pricing in the loop)
    w, h = int(sys.argv[2]), int(sys.argv[3])
    ipo.odhip_interpose_enable_cache(w, h)
    if interpose == 3:
        ipo.odhip_interpose_enable_bands()
    fd = (ctypes.c_void_p * 5)()
    idt = (ctypes.c_void_p * 5)()
    hip.odhip_install_cached_dct_vtbl(fd, idt)
    # ODHIP_CACHE_FDCT_ONLY=1 leaves idct_2d on the reference's C tables (timing runs)
    r.ref_set_external_dct_vtbl(fd, None if os.environ.get("ODHIP_CACHE_FDCT_ONLY") == "1" else idt)
elif len(sys.argv) > 3:
    w, h = int(sys.argv[2]), int(sys.argv[3])
frames = np.concatenate([np.concatenate([p.ravel() for p in synth_frame(w, h, seed=7, phase=5 * f)])
                         for f in range(nframes)]).astype(np.uint8)
out = np.zeros(8 << 20, np.uint8)
sizes = (ctypes.c_long * 64)()
import time
t0 = time.perf_counter()
quality = int(os.environ.get("QUALITY", "20"))
complexity = int(os.environ.get("COMPLEXITY", "7"))
if os.environ.get("CONTENT") == "bench":
    # the bench generator's pictures (any size up to 1920x1080)
    import bench
    fr = []
    for f in range(nframes):
        pl = bench.picture_planes(bench.synth_frame_np(f, 1234))
        fr.append(np.concatenate([pl[0][:h, :w].ravel(), pl[1][:h // 2, :w // 2].ravel(),
                                  pl[2][:h // 2, :w // 2].ravel()]))
    frames = np.concatenate(fr).astype(np.uint8)
    out = np.zeros(64 << 20, np.uint8)
n = r.ref_encode_yuv420(P(frames), w, h, nframes, quality, complexity, 0, P(out), ctypes.c_long(out.size),
                        sizes)
seconds = time.perf_counter() - t0
assert n == nframes, n
total = sum(sizes[i] for i in range(n))
calls = [0] * 6
if ipo is not None:
    arr = (ctypes.c_long * 6).in_dll(ipo, "odhip_interposed_calls")
    calls = [arr[i] for i in range(6)]
stats = [0, 0]
theta = [0, 0, 0, 0]
gpu_ms = 0.0
if interpose in (2, 3):
    gpu_ms = ctypes.c_double.in_dll(ipo, "odhip_interposed_load_ms").value
    hits, misses = ctypes.c_long(), ctypes.c_long()
    ipo.odhip_interpose_cache_stats(ctypes.byref(hits), ctypes.byref(misses))
    stats = [hits.value, misses.value]
if interpose == 3:
    arr = (ctypes.c_long * 4).in_dll(ipo, "odhip_interposed_theta")
    theta = [arr[i] for i in range(4)]
dering = [0, 0]
if ipo is not None and os.environ.get("ODHIP_INTERPOSE_DERING_CACHE") == "1":
    ipo.odhip_interpose_dering_stats()
    arr = (ctypes.c_long * 2).in_dll(ipo, "odhip_interposed_dering")
    dering = [arr[0], arr[1]]      # batched launches, od_dering calls served from them
dist_cache = None
if ipo is not None and os.environ.get("ODHIP_INTERPOSE_DIST_CACHE") == "1":
    ipo.odhip_glue_flush_stats()
    arr = (ctypes.c_long * 2).in_dll(ipo, "odhip_interposed_dist")
    glue_calls = None
    try:
        g = (ctypes.c_long * 2).in_dll(r, "ref_dist_glue_calls")
        glue_calls = [g[0], g[1]]
    except ValueError:
        pass
    dist_cache = {"served": arr[0], "left_to_c": arr[1], "reference_side": glue_calls}
dist_calls = None
try:
    # the build with od_compute_dist bound to od_compute_dist_hip counts its calls
    dist_calls = ctypes.c_long.in_dll(r, "ref_dist_hip_calls").value
except ValueError:
    pass
import hashlib
pkt_digest = None
if os.environ.get("PACKET_DIGEST") == "1":
    # per-packet framing, as tests/_shard_encode.digest
    hh = hashlib.sha256()
    pos = 0
    for i in range(n):
        hh.update(int(sizes[i]).to_bytes(8, "little"))
        hh.update(bytes(out[pos:pos + sizes[i]]))
        pos += sizes[i]
    pkt_digest = hh.hexdigest()
print(json.dumps({"digest": pkt_digest, "packets": hashlib.sha256(bytes(out[:total])).hexdigest(),
                  "sizes": [sizes[i] for i in range(n)], "calls": calls, "cache": stats, "theta": theta, "dering": dering,
                  "gpu_batch_ms": gpu_ms, "dist_hip_calls": dist_calls, "dist_cache": dist_cache,
                  "encode_seconds": seconds}))
