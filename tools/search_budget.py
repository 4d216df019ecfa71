#!/usr/bin/env python3
"""An instruction floor for the K-pulse searches.

From the ISA of the search kernels (hipcc -S of pvq_refbands.hip / pvq_bands.hip, the loops found by
the same walk as tools/asm_loops.py) this prints, for the pulse loops of every search kernel of the
priced step, the VALU instructions per candidate position per pulse, next to the MINIMUM of the
algorithm as the reference states it (src/pvq_encoder.c:165-219) on a machine whose fp64 select is
two 32-bit v_cndmask:

  greedy pulse (:165-187), per candidate j:
      tmp_xy = xy + x[j]                       1  v_add_f64         (x[j] kept as a double)
      tmp_yy = yy + 2*y[j] + 1                 1  v_add_f64         (kept incrementally as a double)
      tmp_xy *= tmp_xy                         1  v_mul_f64
      tmp_xy*best_yy > best_xy*tmp_yy          3  2 v_mul_f64 + v_cmp_gt_f64
      best_xy, best_yy, pos = ...              5  2 + 2 + 1 v_cndmask
      y[pos]++ / x[pos], y[pos] read back      2  v_cmp_eq + v_addc (the band lives in registers:
                                                  no indexed access)
                                              13
  rate-penalised pulse (:192-219), per candidate j:
      tmp_xy = xy + x[j]                       1
      rsqrt(yy + 2*y[j] + 1)                   2  integer index + LDS read (table of the same values)
      2*tmp_xy*norm_1*rsqrt - lambda*j*(...)   3  2 v_mul_f64 + v_sub / v_fma (the penalty hoisted)
      > best_cost; best_cost, pos = ...        4  v_cmp + 2 + 1 v_cndmask
      y[pos]++ ...                             2
                                              12

What the kernels keep OFF this floor on purpose: |x_j| and the denominators are NOT kept as doubles
(2N VGPRs each: an occupancy step, measured slower), so every candidate pays v_cvt_f64_u32 and an
integer add + convert - 13 -> 15-16 as executed for the scan alone.

Reading the table: the counts are STATIC (instructions in the loop body of one pulse).  The per-lane
searches (k_refb_lean_lane, band in registers) are straight-line scans: static = dynamic, 15.9 per
candidate against 13 = 0.82 of the minimum (18.1 = 0.72 before round 4 stopped reading y[pos] back
after a greedy pulse); the extra three are v_cvt_f64_u32 of |x_j| and the integer denominator with
its conversion.  The row searches (k_refb_lean_row, one band per 16 / 4 lanes, 8 positions per lane) carry,
per pulse, a FIXED part that the minimum does not have - float-key proposal, DPP max, ballot, two
64-bit broadcasts, the verification pass over the lane's 8 candidates (one product each since the
second half of round 4: 39 -> 32 static instructions per candidate in the greedy loop), the DPP max +
ballot of the tail - amortised over only 8 candidates per lane, and their tail loop holds both
table-lookup variants (one executes): 28-32 static, ~20-25 executed per candidate.  That fixed part is
the price of running a 128-coefficient band on 16 lanes instead of one (one band per lane: 2.66 ms
instead of 0.82); 16 positions per lane on 8 lanes would halve it per
candidate at +32 VGPRs (not built).  The LDS-column searches of the luma stage (k_decide_*) sit at
14-16 per candidate.

    python tools/search_budget.py            (compiles the two files, ~1 minute)
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-S",
         "--cuda-device-only"]
# kernel substring, source, candidate positions per lane per pulse-loop iteration, exclusive us (r3 serial trace)
KERNELS = [
    ("k_refb_lean_rowILi8ELi16", "pvq_refbands.hip", 8, "128-coefficient chroma bands, one band per 16-lane row (8 positions per lane)"),
    ("k_refb_lean_rowILi8ELi4", "pvq_refbands.hip", 8, "32-coefficient chroma bands, one band per quad"),
    ("k_refb_lean_laneILi15", "pvq_refbands.hip", 15, "15-coefficient chroma bands, one band per lane, band in registers"),
    ("k_refb_lean_laneILi8", "pvq_refbands.hip", 8, "8-coefficient chroma bands, one band per lane"),
    ("k_decide_pair128", "pvq_bands.hip", 64, "128-coefficient luma bands, a lane pair per band (64 positions per lane, LDS column)"),
    ("k_decide_lane32", "pvq_bands.hip", 16, "32-coefficient luma bands, a lane pair per band"),
    ("k_decide_cornerILi0", "pvq_bands.hip", 15, "band 0 of every luma block"),
    ("k_decide_cornerILi1", "pvq_bands.hip", 8, "bands 1-2 of luma blocks of 8x8 and up"),
]
MIN_GREEDY, MIN_TAIL = 13, 12


def cls(op):
    if op.startswith("v_") and "f64" in op:
        return "f64"
    if op.startswith("v_cndmask"):
        return "sel"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    return "other"


def loops(lines, pat):
    start = next(i for i, l in enumerate(lines) if pat in l and re.match(r"^_Z\S+:", l))
    end = next(i for i in range(start, len(lines)) if ".amdhsa_kernel" in lines[i] or lines[i].startswith(".Lfunc_end"))
    labels = {}
    insts = []
    for l in lines[start:end]:
        t = l.strip()
        m = re.match(r"^(\.LBB\d+_\d+):", t)
        if m:
            labels[m.group(1)] = len(insts)
            continue
        if not t or t.startswith(";") or t.startswith("."):
            continue
        insts.append(t.split(";")[0].strip())
    out = []
    for i, t in enumerate(insts):
        m = re.match(r"^s_cbranch_\w+\s+(\.LBB\d+_\d+)", t) or re.match(r"^s_branch\s+(\.LBB\d+_\d+)", t)
        if m and m.group(1) in labels and labels[m.group(1)] <= i:
            a = labels[m.group(1)]
            c = collections.Counter(cls(x.split()[0]) for x in insts[a:i + 1])
            out.append((a, i, c))
    return out


def main():
    tmp = tempfile.mkdtemp()
    asm = {}
    for src in sorted({k[1] for k in KERNELS}):
        dst = os.path.join(tmp, src.replace(".hip", ".s"))
        subprocess.run([HIPCC] + FLAGS + [os.path.join(ROOT, "daala_amd", "csrc", src), "-o", dst], check=True,
                       capture_output=True)
        asm[src] = open(dst).read().splitlines()
    print("%-28s %-6s %-28s %s" % ("kernel", "N/lane", "pulse loop (VALU per candidate)", "fraction of the minimum"))
    for pat, src, n, what in KERNELS:
        ls = loops(asm[src], pat)
        if "lean_row" in pat:
            # round 5: the greedy loop of the row searches is single precision (one v_rcp_f32 per pulse, no
            # double-precision arithmetic outside the replay), the tail loop holds two table-lookup variants of which
            # one executes: identified by what they contain, counted per executed variant
            start = next(i for i, l in enumerate(asm[src]) if pat in l and re.match(r"^_Z\S+:", l))
            end = next(i for i in range(start, len(asm[src])) if ".amdhsa_kernel" in asm[src][i] or asm[src][i].startswith(".Lfunc_end"))
            insts = [t.split(";")[0].strip() for t in (l.strip() for l in asm[src][start:end])
                     if t and not t.startswith(";") and not t.startswith(".") and not re.match(r"^\.LBB\d+_\d+:", t)]
            cands = []
            for a, b, c in ls:
                body = insts[a:b + 1]
                nrcp = sum(1 for x in body if x.startswith("v_rcp_f32"))
                nf32 = sum(1 for x in body if re.match(r"v_(mul|add|fma|cvt)_f32", x) or x.startswith("v_cmp_") and "f32" in x)
                valu = c["f64"] + c["sel"] + c["valu"]
                if nrcp >= 1 and nf32 >= 3*n and valu < 40*n:
                    cands.append((a, b, c, valu))
            # several back edges close one source loop: keep the OUTERMOST range of every nest
            outer = [x for x in cands if not any(y[0] <= x[0] and y[1] >= x[1] and (y[0], y[1]) != (x[0], x[1]) for y in cands)]
            for a, b, c, valu in outer:
                # the literal double-precision replay inside it (#pragma unroll 1) runs only on a failed screen:
                # the inner loops without a v_rcp_f32, one per start address (its largest extent)
                inner = {}
                for x in ls:
                    if x[0] > a and x[1] < b and not any(t.startswith("v_rcp_f32") for t in insts[x[0]:x[1] + 1]):
                        if x[0] not in inner or x[1] > inner[x[0]][1]:
                            inner[x[0]] = x
                rep = sum(x[2]["f64"] + x[2]["sel"] + x[2]["valu"] for x in inner.values())
                per = (valu - rep)/float(n)
                print("%-28s %-6d %-28s %.2f" % (pat.replace("ILi", "<").replace("ELi", ",").rstrip("E"), n,
                      "greedy, f32 screen: %.1f (+ %.1f in the replay path; f64 %.1f, select %.1f)" % (
                          per, rep/float(n), c["f64"]/n, c["sel"]/n), MIN_GREEDY/per))
            print("    %s" % what)
            continue
        # innermost pulse loops: bodies with at least 3 fp64 operations per candidate and no larger loop inside
        cand = []
        for a, b, c in ls:
            valu = c["f64"] + c["sel"] + c["valu"]
            if c["f64"] >= 3 * n and valu <= 40 * n and c["sel"] >= 2.5 * n:     # an argmax scan selects
                inner = [x for x in ls if x[0] >= a and x[1] <= b and (x[0], x[1]) != (a, b)
                         and x[2]["f64"] >= 3 * n]
                if not inner:
                    cand.append((a, b, c, valu))
        seen = set()
        for a, b, c, valu in cand:
            key = (c["f64"], c["sel"])
            if key in seen:
                continue
            seen.add(key)
            per = valu / float(n)
            # the tail loop reads the 1/sqrt table from LDS, the greedy loop does not
            tail = c["lds"] >= n // 2
            mn = MIN_TAIL if tail else MIN_GREEDY
            if per < mn:
                continue          # a partially unrolled piece of a pulse, not a whole pulse loop
            print("%-28s %-6d %-28s %.2f" % (pat.replace("ILi", "<").replace("ELi", ",").rstrip("E"), n,
                                             "%s: %.1f (f64 %.1f, select %.1f, other %.1f)" % (
                                                 "reads LDS" if tail else "registers", per, c["f64"] / n, c["sel"] / n,
                                                 c["valu"] / n), mn / per))
        print("    %s" % what)


if __name__ == "__main__":
    sys.exit(main())
