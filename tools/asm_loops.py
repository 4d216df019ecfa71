#!/usr/bin/env python3
"""Loops of one kernel in a device assembly listing (hipcc -S --cuda-device-only): for every
backward branch, the instruction mix of the loop body.  usage: asm_loops.py file.s kernel-substring"""
import collections
import re
import sys

lines = open(sys.argv[1]).read().splitlines()
pat = sys.argv[2]
start = next(i for i, l in enumerate(lines) if pat in l and re.match(r"^_Z\S+:", l))
end = next(i for i in range(start, len(lines)) if ".amdhsa_kernel" in lines[i] or lines[i].startswith(".Lfunc_end"))
body = lines[start:end]
labels = {}
insts = []
for l in body:
    t = l.strip()
    m = re.match(r"^(\.LBB\d+_\d+):", t)
    if m:
        labels[m.group(1)] = len(insts)
        continue
    if not t or t.startswith(";") or t.startswith("."):
        continue
    insts.append(t.split(";")[0].strip())
print("instructions:", len(insts))


def cls(op):
    if op.startswith("v_") and "f64" in op:
        return "valu_f64"
    if op.startswith("v_cndmask"):
        return "valu_sel"
    if op.startswith("v_cmp"):
        return "valu_cmp"
    if op.startswith("v_"):
        return "valu_other"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_") or op.startswith("scratch_"):
        return "vmem"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_"):
        return "salu"
    return "other"


for i, t in enumerate(insts):
    m = re.match(r"^s_cbranch_\w+\s+(\.LBB\d+_\d+)", t) or re.match(r"^s_branch\s+(\.LBB\d+_\d+)", t)
    if m and m.group(1) in labels and labels[m.group(1)] <= i:
        a = labels[m.group(1)]
        c = collections.Counter(cls(x.split()[0]) for x in insts[a:i + 1])
        n = i + 1 - a
        if n >= int(sys.argv[3]) if len(sys.argv) > 3 else 20:
            print("loop %s [%d..%d] %d instr: %s" % (m.group(1), a, i, n, dict(c)))
