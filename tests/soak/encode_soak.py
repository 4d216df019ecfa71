#!/usr/bin/env python3
"""Randomised soak of the bindings behind the REAL reference encoder (tests/interpose/run_interposed.py per clip): random
picture sizes, clip lengths (keyframe + inter frames), qualities and complexities; packets must be byte-identical to the
plain C encoder's.  Modes in turn:
  3  frame cache + batched PVQ band stage (one GPU pass per keyframe serves fdct_2d and the keyframe luma pvq_theta calls,
     host pricing), every served band and every batched rate self-checked (ODHIP_CACHE_CHECK, ODHIP_RATE_CHECK)
  3d the same plus the deringing level search from batched passes (ODHIP_DERING_CHECK)
  1  every per-call surface on the GPU (filters, pvq_search_rdo_double, od_dering; the 2-D transforms through the vtbl) -
     small pictures only: one round trip per call
TEST INFRASTRUCTURE.   usage: encode_soak.py [max_seconds=600] [seed0=0]"""
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
max_seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 600
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0


def run(mode, w, h, **env):
    e = dict(os.environ)
    e.update({k: str(v) for k, v in env.items()})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "interpose", "run_interposed.py"), str(mode), str(w),
                        str(h)], capture_output=True, text=True, timeout=1500, env=e)
    if p.returncode != 0:
        print("mode %s %dx%d %s\n%s" % (mode, w, h, env, p.stderr[-3000:]))
        sys.exit(1)
    return json.loads(p.stdout.strip().splitlines()[-1])


t0 = time.time()
case = seed0
served_bands = served_dering = percall = packets = 0
while time.time() - t0 < max_seconds:
    rng = np.random.RandomState(8800 + case)
    kind = case % 3
    if kind == 2:
        w, h = 2*int(rng.randint(16, 65)), 2*int(rng.randint(16, 49))       # per call: up to 128x96
    else:
        w, h = 2*int(rng.randint(16, 330)), 2*int(rng.randint(16, 200))
    n = int(rng.randint(1, 4))
    quality = int([3, 8, 15, 20, 30, 45, 60, 90][rng.randint(8)])
    complexity = int([7, 7, 7, 2, 10][rng.randint(5)])
    env = dict(NFRAMES=n, QUALITY=quality, COMPLEXITY=complexity)
    plain = run(0, w, h, ODHIP_INTERPOSE_PASSTHROUGH=1, **env)
    if kind == 0:
        res = run(3, w, h, ODHIP_INTERPOSE_PASSTHROUGH=1, ODHIP_CACHE_CHECK=1, ODHIP_RATE_CHECK=1, **env)
        tag = "frame cache + band stage"
        served_bands += res["theta"][0]
    elif kind == 1:
        res = run(3, w, h, ODHIP_INTERPOSE_PASSTHROUGH=1, ODHIP_CACHE_CHECK=1, ODHIP_INTERPOSE_DERING_CACHE=1,
                  ODHIP_DERING_CHECK=1, **env)
        tag = "frame cache + band stage + dering cache"
        served_bands += res["theta"][0]
        served_dering += res["dering"][1]
    else:
        res = run(1, w, h, ODHIP_INTERPOSE_VTBL=1, **env)
        tag = "every per-call surface"
        percall += sum(res["calls"])
    packets += n
    ok = res["sizes"] == plain["sizes"] and res["packets"] == plain["packets"]
    label = "%dx%d %d frame(s) -v %d -z %d %s" % (w, h, n, quality, complexity, tag)
    if not ok:
        print("case %d %s: MISMATCH %s vs %s" % (case, label, res, plain), flush=True)
        sys.exit(1)
    print("case %3d %-78s packets equal" % (case, label), flush=True)
    case += 1
print("encode soak: %d clips / %d packets byte-identical to the plain C encoder's; %d bands served from the batched stage, "
      "%d dering superblocks from batched passes, %d per-call GPU round trips; %.0f s"
      % (case - seed0, packets, served_bands, served_dering, percall, time.time() - t0))
