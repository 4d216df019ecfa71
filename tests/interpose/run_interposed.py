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
nframes = 2
hip = ctypes.CDLL(os.path.join(ROOT, "daala_amd", "lib", "libdaalahip.so"), mode=ctypes.RTLD_GLOBAL)
ipo = None
if interpose:
    assert hip.odhip_init(0) == 0
    ipo = ctypes.CDLL(os.path.join(ROOT, "tests", "interpose", "libinterpose.so"),
                      mode=ctypes.RTLD_GLOBAL)
r = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "libdaalaref.so"))
frames = np.concatenate([np.concatenate([p.ravel() for p in synth_frame(w, h, seed=7, phase=5 * f)])
                         for f in range(nframes)]).astype(np.uint8)
out = np.zeros(1 << 20, np.uint8)
sizes = (ctypes.c_long * 16)()
n = r.ref_encode_yuv420(P(frames), w, h, nframes, 20, 7, 0, P(out), ctypes.c_long(out.size), sizes)
assert n == nframes, n
total = sum(sizes[i] for i in range(n))
calls = [0] * 5
if ipo is not None:
    arr = (ctypes.c_long * 5).in_dll(ipo, "odhip_interposed_calls")
    calls = [arr[i] for i in range(5)]
print(json.dumps({"packets": bytes(out[:total]).hex(), "sizes": [sizes[i] for i in range(n)],
                  "calls": calls}))
