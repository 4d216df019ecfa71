"""Child process of tests/test_gpu_decode_check.py: encode a clip with the real reference
encoder, decode it with the real reference decoder, and at the decoder's frame-level
post-filter let odhip_inverse_partition reconstruct every plane from the decoded
coefficients and the block-size map (tests/interpose mode 4).  Prints a JSON line."""
import ctypes
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from _libs import P, synth_frame  # noqa: E402

w, h, nframes = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
quality = int(os.environ.get("QUALITY", "20"))
check = os.environ.get("DECODE_CHECK", "1") == "1"
# DERING_CACHE=1: encoder AND decoder take every od_dering call from batched GPU passes
# (odhip_dering_cache), each served superblock compared with the reference's own od_dering
dering = os.environ.get("DERING_CACHE") == "1"
if dering:
    os.environ["ODHIP_INTERPOSE_DERING_CACHE"] = "1"
    os.environ["ODHIP_DERING_CHECK"] = "1"
# SYNTHESIS=1: every od_pvq_synthesis_partial call of the encoder and the decoder (od_pvq_decode's
# pvq_decode_partition, src/pvq_decoder.c:87) on the GPU, one band per call
synthesis = os.environ.get("SYNTHESIS") == "1"
if synthesis:
    os.environ["ODHIP_INTERPOSE_SYNTHESIS"] = "1"
os.environ["ODHIP_INTERPOSE_PASSTHROUGH"] = "1"
hip = ctypes.CDLL(os.path.join(ROOT, "daala_amd", "lib", "libdaalahip.so"), mode=ctypes.RTLD_GLOBAL)
ipo = None
if check or dering or synthesis:
    assert hip.odhip_init(0) == 0
    ipo = ctypes.CDLL(os.path.join(ROOT, "tests", "interpose", "libinterpose.so"), mode=ctypes.RTLD_GLOBAL)
r = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "libdaalaref.so"))
if ipo is not None:
    ipo.odhip_interpose_set_reference(ctypes.c_void_p(r._handle))
    if check:
        ipo.odhip_interpose_enable_decode_check()
if os.environ.get("CONTENT") == "bench":
    import bench
    fr = []
    for f in range(nframes):
        pl = bench.picture_planes(bench.synth_frame_np(f, 1234))
        fr.append(np.concatenate([pl[0][:h, :w].ravel(), pl[1][:h // 2, :w // 2].ravel(),
                                  pl[2][:h // 2, :w // 2].ravel()]))
    frames = np.concatenate(fr).astype(np.uint8)
else:
    frames = np.concatenate([np.concatenate([p.ravel() for p in synth_frame(w, h, seed=9, phase=3 * f)])
                             for f in range(nframes)]).astype(np.uint8)
decoded = np.zeros(frames.size, np.uint8)
n = r.ref_roundtrip_yuv420(P(frames), w, h, nframes, quality, 7, P(decoded))
assert n == nframes, n
stats = [0, 0, 0]
if ipo is not None:
    arr = (ctypes.c_long * 3).in_dll(ipo, "odhip_interposed_decode")
    stats = [arr[i] for i in range(3)]
err = float(np.abs(decoded.astype(np.int32) - frames.astype(np.int32)).mean())
dstats = [0, 0]
if dering:
    ipo.odhip_interpose_dering_stats()
    arr = (ctypes.c_long * 2).in_dll(ipo, "odhip_interposed_dering")
    dstats = [arr[0], arr[1]]
nsynth = 0
if synthesis:
    nsynth = ctypes.c_long.in_dll(ipo, "odhip_glue_synth_calls").value
print(json.dumps({"synthesis_calls": nsynth, "decoded": hashlib.sha256(decoded.tobytes()).hexdigest(), "check": stats, "dering": dstats,
                  "mean_abs_error_vs_source": err}))
