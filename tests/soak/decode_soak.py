#!/usr/bin/env python3
"""Randomised soak of the decoder-side surfaces inside the REAL reference codec (tests/interpose/run_decode_check.py:
encode a clip with the reference encoder, decode it with the reference decoder): for random picture sizes, clip
lengths and qualities
  * odhip_inverse_partition reconstructs every plane of every decoded frame from the decoder's dequantised coefficients
    and block-size map, compared pixel by pixel with the decoder's own reconstruction;
  * with SYNTHESIS=1 every od_pvq_synthesis_partial call of encoder and decoder runs on the GPU;
  * with DERING_CACHE=1 every od_dering call of encoder and decoder is served from batched passes (and compared);
and the decoded clip must hash like the plain reference's.  TEST INFRASTRUCTURE.
usage: decode_soak.py [max_seconds=600] [seed0=0]"""
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
max_seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 600
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0


def run(w, h, n, **env):
    e = dict(os.environ)
    e.update({k: str(v) for k, v in env.items()})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "interpose", "run_decode_check.py"), str(w), str(h),
                        str(n)], capture_output=True, text=True, timeout=1500, env=e)
    if p.returncode != 0:
        print(p.stderr[-3000:])
        sys.exit(1)
    return json.loads(p.stdout.strip().splitlines()[-1])


t0 = time.time()
case = seed0
planes = pixels = synth = served = 0
while time.time() - t0 < max_seconds:
    rng = np.random.RandomState(4100 + case)
    w, h = 2*int(rng.randint(16, 260)), 2*int(rng.randint(16, 160))
    n = int(rng.randint(1, 4))
    quality = int([3, 8, 15, 20, 30, 45, 60, 90][rng.randint(8)])
    mode = case % 3
    plain = run(w, h, n, QUALITY=quality, DECODE_CHECK=0)
    if mode == 0:
        res = run(w, h, n, QUALITY=quality)
    elif mode == 1:
        res = run(w, h, n, QUALITY=quality, SYNTHESIS=1)
    else:
        res = run(w, h, n, QUALITY=quality, DECODE_CHECK=0, DERING_CACHE=1)
    ok = res["decoded"] == plain["decoded"]
    if mode != 2:
        p_, px_, bad = res["check"]
        ok = ok and p_ == 3*n and bad == 0
        planes += p_
        pixels += px_
    if mode == 1:
        ok = ok and res["synthesis_calls"] > 0
        synth += res["synthesis_calls"]
    if mode == 2:
        served += res["dering"][1]
    tag = "%dx%d %d frame(s) -v %d %s" % (w, h, n, quality, ["inverse check", "inverse check + synthesis on the GPU",
                                                          "dering from batched passes"][mode])
    if not ok:
        print("case %d %s: MISMATCH %s" % (case, tag, res), flush=True)
        sys.exit(1)
    print("case %3d %-70s equal" % (case, tag), flush=True)
    case += 1
print("decode soak: %d clips equal; %d planes / %d pixels reconstructed by odhip_inverse_partition inside the decoder, "
      "%d synthesis calls on the GPU, %d dering superblocks served; %.0f s" % (case - seed0, planes, pixels, synth, served,
                                                                               time.time() - t0))
