"""The inverse stage's VALU budget, level by level (the counterpart of tools/pyr_budget.py).

(1) The NETWORK MINIMUM per reconstructed pixel of a partition level with leaf size N (one launch
    group reconstructs every level): dequantisation of the coded coefficients
    (od_pvq_synthesis_partial noref, src/pvq.c:1081-1092: unpack, OD_MULT16_32_Q16 as one mulhi,
    inverse-QM multiply, round + shift = 5; with a reference, :1094-1114: + Householder term 8 = 13),
    the 2-D inverse transform (2 x the 1-D network of oracle/od_lifting_tables.h per N pixels,
    priced as in pyr_budget.py), od_postfilter_split of every level above the leaf (2 four-tap
    post-filters per N' pixels per level, src/filter.c:195-222: 24 instructions each - two of them
    the truncating divisions by 75 and 85, a v_mul_hi + 3), the superblock-edge post-filter, and
    od_coeff_to_ref_buf (src/state.c:1296-1304: add, shift, add, clamp + a quarter of the byte
    packing = 4.75 per pixel).
(2) What the kernels execute: SQ_INSTS_VALU of the round-4 PMC pass (profiles/r4_inverse_segments.txt),
    per pixel and level.

    python tools/inv_budget.py
"""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OPS = {0: "load", 1: "store", 2: "add", 3: "sub", 4: "rshift1", 5: "mulshift", 6: "neg", 7: "shr", 8: "mov"}
MIN_INSTR = {"load": 0, "store": 0, "add": 1, "sub": 1, "rshift1": 2, "mulshift": 2, "neg": 1, "shr": 1, "mov": 0}
POST4 = 24.0
TO_PX = 4.75


def networks(kind):
    txt = open(os.path.join(ROOT, "oracle", "od_lifting_tables.h")).read()
    out = {}
    for m in re.finditer(r"static const od_lift_op OD_LIFT_%s(\d+)\[(\d+)\] = \{(.*?)\};" % kind, txt, re.S):
        n = int(m.group(1))
        cnt = {}
        for row in re.finditer(r"\{(\d+),", m.group(3)):
            k = OPS[int(row.group(1))]
            cnt[k] = cnt.get(k, 0) + 1
        out[n] = sum(MIN_INSTR[k] * v for k, v in cnt.items())
    return out


def level_minimum(n, tile, net, deq):
    """per pixel of a tile of side `tile` at leaf size n"""
    coded = min(n * n, 512) / float(n * n)
    idct = 2.0 * net[n] / n
    split = sum(2.0 / m * POST4 for m in (8, 16, 32, 64) if m > n and m <= tile)
    edge = 2.0 / tile * POST4      # both directions of od_apply_postfilter_frame_sbs
    return coded * deq, idct, split, edge, TO_PX


def main():
    net = networks("IDCT")
    print("1-D inverse networks, minimum instructions: " + ", ".join("N=%d: %d" % (n, net[n]) for n in sorted(net)))
    px_luma = 16 * 1920 * 1088
    px_chroma = 32 * 960 * 544
    rows = []
    for name, tile, deq, levels in (("luma (no reference)", 64, 5.0, (4, 8, 16, 32, 64)),
                                    ("chroma (with reference)", 32, 13.0, (4, 8, 16, 32))):
        print("\n%s: minimum VALU instructions per reconstructed pixel" % name)
        tot = 0.0
        for n in levels:
            d, i, s, e, p = level_minimum(n, tile, net, deq)
            t = d + i + s + e + p
            tot += t
            print("  leaf %2dx%-2d: dequantise %.1f + iDCT %.1f + split post-filters %.1f + superblock edges %.1f + "
                  "pixels %.2f = %.1f" % (n, n, d, i, s, e, p, t))
            rows.append((name, n, t))
        print("  all %d levels: %.1f per pixel of the plane" % (len(levels), tot))
    # executed: round-4 counters (profiles/r4_inverse_segments.txt, per 16-frame launch)
    ex = {"luma": (81.6e6 + 48.9e6 + 1.26e6 + 2.35e6, px_luma, 5),
          "chroma": (66.5e6 + 16.3e6 + 0.51e6 + 2.94e6, px_chroma, 4)}
    mins = {"luma": sum(t for nm, n, t in rows if nm.startswith("luma")),
            "chroma": sum(t for nm, n, t in rows if nm.startswith("chroma"))}
    print()
    for k, (valu, px, nlev) in ex.items():
        per_px = valu * 64 / px
        print("%s executed (round 4, walking kernels + top2 + edges): %.1f M wave-instructions per step = %.1f per "
              "pixel of the plane (%.1f per pixel and level); minimum %.1f -> %.2fx the minimum" % (
                  k, valu / 1e6, per_px, per_px / nlev, mins[k], per_px / mins[k]))
    # round 5 (profiles/r5_pmc_traffic.json): 4x4 leaves in registers, 24-bit multiplies, loads one group ahead.
    # The luma walker executes MORE instructions than round 4's (the prefetch recomputes its block indices in
    # three places, OD_MULT16_32_Q16 became three full-rate instructions instead of one quarter-rate one) and is
    # faster: profiles/r5_inverse_phases.txt - the stage does not follow its instruction count.
    r5 = {"luma": (85.3e6 + 48.2e6 + 1.23e6 + 2.35e6, px_luma, 5), "chroma": (65.6e6 + 17.6e6 + 0.50e6 + 2.94e6, px_chroma, 4)}
    for k, (valu, px, nlev) in r5.items():
        per_px = valu * 64 / px
        print("%s executed (round 5): %.1f M wave-instructions per step = %.1f per pixel of the plane (%.1f per pixel "
              "and level) -> %.2fx the minimum" % (k, valu / 1e6, per_px, per_px / nlev, per_px / mins[k]))
    r3 = {"luma": 84.1e6 + 54.2e6 + 3.69e6 + 2.35e6, "chroma": 100.4e6 + 3.48e6 + 2.94e6}
    for k, valu in r3.items():
        per_px = valu * 64 / ex[k][1]
        print("%s executed (round 3): %.1f per pixel of the plane -> %.2fx the minimum" % (k, per_px, per_px / mins[k]))
    print("\nWhere the rest goes (ISA of k_inverse_walk, per 64x64 luma tile at leaf 4x4, wave-instructions x 64 / "
          "4096 px): dequantise-on-load ~9 per pixel for 5 of arithmetic (chunk / band / block index chains, the 16 "
          "scattered ds_write addresses); 4-point passes ~20 for 9.5 (each 4-point network of 12 instructions carries "
          "8 of LDS addressing and loop control, twice); split post-filters ~17 for 11; the pixel window ~6 for 4.75.")


if __name__ == "__main__":
    main()
