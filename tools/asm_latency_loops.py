#!/usr/bin/env python3
"""Every kernel of a .hip source: loops whose body waits for its OWN global loads (a load followed by
s_waitcnt vmcnt inside one backward branch) - each trip pays a full memory latency unless other waves
cover it.  usage: tools/asm_latency_loops.py pvq_refbands.hip [kernel-regex]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "daala_amd", "csrc")
src = sys.argv[1]
kpat = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
out = "/tmp/asm_ll_%s.s" % os.path.basename(src)
r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                    "-fno-fast-math", "-S", "--cuda-device-only", os.path.join(CSRC, src), "-o", out],
                   capture_output=True, text=True)
if r.returncode:
    print(r.stderr[-2000:])
    sys.exit(1)
lines = open(out).read().splitlines()
i = 0
while i < len(lines):
    m = re.match(r"^(_Z\S+):", lines[i])
    if not m:
        i += 1
        continue
    name = subprocess.run(["/usr/bin/c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    j = i + 1
    labels = {}
    insts = []
    while j < len(lines) and not lines[j].startswith(".Lfunc_end"):
        t = lines[j].strip()
        mm = re.match(r"^(\.LBB\d+_\d+):", t)
        if mm:
            labels[mm.group(1)] = len(insts)
        elif t and not t.startswith(";") and not t.startswith("."):
            insts.append(t.split(";")[0].strip())
        j += 1
    i = j
    if kpat and not kpat.search(name):
        continue
    rows = []
    for k, ins in enumerate(insts):
        mm = re.match(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", ins) or re.match(r"s_branch\s+(\.LBB\d+_\d+)", ins)
        if not mm or mm.group(1) not in labels or labels[mm.group(1)] > k:
            continue
        body = insts[labels[mm.group(1)]:k + 1]
        nload = sum(1 for b in body if b.startswith("global_load") or b.startswith("buffer_load"))
        nstore = sum(1 for b in body if b.startswith("global_store"))
        nwait = sum(1 for b in body if b.startswith("s_waitcnt") and "vmcnt" in b)
        nlds = sum(1 for b in body if b.startswith("ds_"))
        nlgkm = sum(1 for b in body if b.startswith("s_waitcnt") and "lgkmcnt" in b)
        nvalu = sum(1 for b in body if b.startswith("v_"))
        if nload and nwait:
            rows.append("   loop of %4d instr: %3d global loads, %2d vmcnt waits, %3d LDS ops (%d lgkm waits), %4d VALU, %d stores"
                        % (len(body), nload, nwait, nlds, nlgkm, nvalu, nstore))
    if rows:
        print(name)
        print("\n".join(rows))
