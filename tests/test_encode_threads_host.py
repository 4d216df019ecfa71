"""Several encoder contexts in several host threads of ONE process, with the shim loaded in front of
the reference encoder (every surface forwarded to the reference's own C: no GPU here): the packets
of every thread's frames must equal the sequential encoder's.  This is the host-side premise of
`bench.py --encode-frames N --threads-per-proc T` (encode_job.py): the reference encoder is
thread-compatible (one daala_enc_ctx per thread, src has no mutable globals besides logging), the
shim keeps its state per thread, and ctypes releases the GIL for the duration of the C call."""
import ctypes
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HAVE = os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libdaalaref.so")) and \
    os.path.exists(os.path.join(ROOT, "shim", "libdaalahipglue.so"))

CHILD = r'''
import ctypes, json, sys, threading
import numpy as np
sys.path.insert(0, %(root)r)
sys.path.insert(0, %(root)r + "/tests")
import encode_job as S
from _libs import synth_frame
hip = ctypes.CDLL(%(root)r + "/daala_amd/lib/libdaalahip.so", mode=ctypes.RTLD_GLOBAL)
glue = ctypes.CDLL(S.GLUE_LIB, mode=ctypes.RTLD_GLOBAL)
cfg = S.GlueConfig()
glue.odhip_glue_default_config(ctypes.byref(cfg))
cfg.bind_filters = cfg.bind_search = cfg.bind_dering = cfg.bind_dct_vtbl = 0    # everything -> the reference's C
glue.odhip_glue_configure(ctypes.byref(cfg))        # (odhip_init fails without a GPU: nothing is bound to it)
r = ctypes.CDLL(S.REFERENCE_LIB)
glue.odhip_glue_set_reference(ctypes.c_void_p(r._handle))
w, h, n, T = 176, 120, 8, 4
yuv = [np.concatenate([p.ravel() for p in synth_frame(w, h, seed=3, phase=7 * i)]).astype(np.uint8) for i in range(n)]
seq = S.encode_frames(r, list(range(n)), yuv, w, h)
outs = [{} for _ in range(T)]
def run(t):
    idx = list(range(t, n, T))
    outs[t].update(S.encode_frames(r, idx, [yuv[i] for i in idx], w, h))
ths = [threading.Thread(target=run, args=(t,)) for t in range(T)]
[th.start() for th in ths]
[th.join() for th in ths]
par = {}
for o in outs: par.update(o)
st = S.glue_stats(glue)
print(json.dumps({"equal": [par[i] == seq[i] for i in range(n)], "calls": [st.calls[i] for i in range(6)]}))
'''


@pytest.mark.skipif(not HAVE, reason="oracle/_ref or shim/libdaalahipglue.so not built")
def test_four_encoder_threads_equal_the_sequential_encoder():
    import json
    p = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads(p.stdout.strip().splitlines()[-1])
    assert d["equal"] == [True] * 8, d
    assert all(c > 0 for c in d["calls"][:5]), d["calls"]      # the calls went through the shim's definitions
