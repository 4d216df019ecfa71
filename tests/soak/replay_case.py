#!/usr/bin/env python3
"""Replay ONE case of parity_soak.py (same seed, same draws) and print what differs band by band.
usage: [SOAK_BIG=1] replay_case.py case"""
import os
import runpy
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _pipeline_check as C     # noqa: E402

case = int(sys.argv[1])
orig = C.compare_decisions


def verbose(gpu, cpu, frame=0, frames=1):
    bad = orig(gpu, cpu, frame, frames)
    for pli in range(3):
        set_ = 1 if pli else 0
        plane = frame if pli == 0 else (pli - 1)*frames + frame
        for bs, (yc, bc) in enumerate(cpu[pli]):
            yg, bg, coded = gpu[(set_, bs)]
            per = yc.shape[0]
            sl = slice(plane*per, (plane + 1)*per)
            yg, bg, coded = yg[sl], bg[sl], coded[sl]
            nb, offs, _ = C._layout(bs)
            rows, cols = np.nonzero((bg != bc).any(axis=2))
            for r, c in list(zip(rows, cols))[:12]:
                a, b = offs[c], offs[c + 1]
                print("plane %d level %d block %d band %d: gpu {qg, itheta, max_theta, k} %s  reference %s  sum|y| gpu %d ref %d"
                      % (pli, bs, r, c, bg[r, c].tolist(), bc[r, c].tolist(), int(np.abs(yg[r, a:b]).sum()),
                         int(np.abs(yc[r, a:b]).sum())))
            for i in range(nb):
                a, b = offs[i], offs[i + 1]
                on = coded[:, i]
                rr = np.nonzero(on & (yg[:, a:b] != yc[:, a:b]).any(axis=1))[0]
                for r in rr[:6]:
                    print("plane %d level %d block %d band %d pulses differ: gpu %s\n    reference %s" % (
                        pli, bs, r, i, yg[r, a:b][:24].tolist(), yc[r, a:b][:24].tolist()))
    return bad


C.compare_decisions = verbose
sys.argv = [os.path.join(HERE, "parity_soak.py"), "1", str(case)]
try:
    runpy.run_path(sys.argv[0], run_name="__main__")
except SystemExit as e:
    print("exit", e.code)
