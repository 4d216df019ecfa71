"""The forward pyramid's VALU budget, from first principles.

(1) The NETWORK MINIMUM: the reference's lifting networks (src/dct.c od_bin_fdct4..64) as op
    tables (oracle/od_lifting_tables.h, extracted from the reference) give, per N-point 1-D
    transform, the additions / subtractions, OD_DCT_RSHIFT steps, multiply-shift steps,
    negations and shifts; a 2-D N x N transform is 2N of them per N*N pixels; the pyramid runs
    all five sizes; the split pre-filters (src/filter.c:147-193) add 2/N four-tap filters per
    pixel per level that is split.  Each op is priced at the FEWEST gfx950 instructions that
    implement it exactly:
        add / sub / neg / shift            1   (v_add / v_sub / v_ashrrev)
        OD_DCT_RSHIFT(a, 1)                2   (add of the sign bit + shift: (a + (a >>> 31)) >> 1;
                                               v_lshrrev + v_add + v_ashrrev would be 3 - the SDWA
                                               form used by OdMul24S folds two of them)
        (a*C + R) >> S                     2   (v_mad_i32_i24 + v_ashrrev_i32; a 32-bit
                                               v_mul_hi has no rounding term and a 4x issue cost)
(2) What the kernel executes: SQ_INSTS_VALU of k_forward_pyramid64x2 from the committed PMC
    pass (profiles/r3_pmc_traffic.json), per luma pixel.
(3) With the measured issue cost per wave-instruction (tools/ubench/valu_rate: pass its output
    with --rates FILE, else 4 cycles), the time below which NO implementation of these
    networks on 1024 SIMDs can go, and what that is as a fraction of the 8 TB/s HBM peak at 21
    algorithmic bytes per luma pixel.

    python tools/pyr_budget.py [--rates gpurun_out/valu_rate.txt] [--us-per-16-frames 216]
"""
import argparse
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OPS = {0: "load", 1: "store", 2: "add", 3: "sub", 4: "rshift1", 5: "mulshift", 6: "neg", 7: "shr", 8: "mov"}
MIN_INSTR = {"load": 0, "store": 0, "add": 1, "sub": 1, "rshift1": 2, "mulshift": 2, "neg": 1, "shr": 1, "mov": 0}


def networks():
    txt = open(os.path.join(ROOT, "oracle", "od_lifting_tables.h")).read()
    out = {}
    for m in re.finditer(r"static const od_lift_op OD_LIFT_FDCT(\d+)\[(\d+)\] = \{(.*?)\};", txt, re.S):
        n = int(m.group(1))
        cnt = {}
        for row in re.finditer(r"\{(\d+),", m.group(3)):
            k = OPS[int(row.group(1))]
            cnt[k] = cnt.get(k, 0) + 1
        out[n] = cnt
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rates", default=None)
    ap.add_argument("--us-per-16-frames", type=float, default=None)
    a = ap.parse_args()
    net = networks()
    print("1-D forward networks (ops per N-point transform):")
    tot_px = 0.0
    per_level = {}
    for n in sorted(net):
        c = net[n]
        instr = sum(MIN_INSTR[k] * v for k, v in c.items())
        per_px = 2.0 * instr / n          # column pass + row pass, N transforms of N points per N*N pixels
        per_level[n] = per_px
        tot_px += per_px
        print("  N=%2d: add/sub %3d  rshift1 %3d  mulshift %3d  neg %2d  shr %2d  -> %4d instructions = %.2f per "
              "pixel for the 2-D transform" % (n, c.get("add", 0) + c.get("sub", 0), c.get("rshift1", 0),
                                               c.get("mulshift", 0), c.get("neg", 0), c.get("shr", 0), instr, per_px))
    # od_pre_filter4 (src/filter.c:147-193): 4 butterflies in, 2 scalings (mul + shift + sign fix-up: 4 ops each),
    # 2 lifting steps (mul + add-round + shift + add: 3 each), 4 butterflies / shifts out  ~ 22 instructions
    filt = 22
    split = sum(2.0 / n for n in (64, 32, 16, 8)) * filt + (2.0 / 64) * filt
    conv = 2.0 / 1                                      # (p - 128) << 4: sub + shift per pixel
    net_total = tot_px + split + conv
    print("network minimum: transforms %.1f + split / superblock-edge pre-filters %.1f + pixel conversion %.1f = %.1f "
          "VALU instructions per luma pixel" % (tot_px, split, conv, net_total))
    pmc = json.load(open(os.path.join(ROOT, "profiles", "r3_pmc_traffic.json")))["kernels"]
    key = [k for k in pmc if k.startswith("k_forward_pyramid64x2")][0]
    valu = pmc[key]["valu_wave_instructions"]
    px = 16 * 1920 * 1088
    executed = valu * 64 / px
    print("executed (SQ_INSTS_VALU %.1f M wave-instructions per 16-frame launch): %.1f per luma pixel -> %.1f (%.0f %%) "
          "are not network arithmetic: LDS addressing, tile stores, copy-out of the parity-split layout, loop control"
          % (valu / 1e6, executed, executed - net_total, 100 * (executed - net_total) / executed))
    c_add = c_shr = c_mad = 4.0
    if a.rates:
        vals = {}
        for line in open(a.rates):
            m = re.match(r"(\S+)\s+[\d.]+ ms\s+([\d.]+) cycles", line)
            if m:
                vals[m.group(1)] = float(m.group(2))
        c_add = vals.get("v_add_u32", 4.0)
        c_shr = vals.get("v_ashrrev_i32", 4.0)
        c_mad = vals.get("v_mad_i32_i24", 4.0)
        print("measured issue cost (tools/ubench/valu_rate on this GPU, cycles per wave-instruction per SIMD at the "
              "2.4 GHz peak clock): v_add_u32 %.2f, v_ashrrev_i32 %.2f, v_mad_i32_i24 %.2f" % (c_add, c_shr, c_mad))
    # cycles per luma pixel of the network minimum, class by class
    cyc_px = 0.0
    for n, c in net.items():
        cy = ((c.get("add", 0) + c.get("sub", 0) + c.get("neg", 0)) * c_add + c.get("shr", 0) * c_shr
              + c.get("rshift1", 0) * (c_add + c_shr) + c.get("mulshift", 0) * (c_mad + c_shr))
        cyc_px += 2.0 * cy / n
    # pre-filters: 22 instructions of which 4 multiplies; pixel conversion: sub + shift
    cyc_px += (split / filt) * (14 * c_add + 4 * c_shr + 4 * c_mad) + c_add + c_shr
    other = executed - net_total
    mix = cyc_px / net_total
    rows = (("network minimum", net_total, cyc_px),
            ("as executed", executed, cyc_px + other * mix))
    for name, ipp, cpp in rows:
        us = cpp * px / 64 / 1024 / 2.4e9 * 1e6
        print("%-16s: %.1f instr/px, %.1f SIMD cycles/px (x64 lanes) = %.1f us per 16 frames on 1024 SIMDs at 2.4 GHz "
              "= %.2f TB/s algorithmic = %.3f of the 8 TB/s peak" % (name, ipp, cpp, us, px * 21 / us / 1e6,
                                                                    px * 21 / us / 1e6 / 8))
    print("(the non-network instructions are priced at the network's average cost, %.2f cycles)" % mix)
    if a.us_per_16_frames:
        print("measured: %.1f us per 16 frames = %.3f of peak" % (a.us_per_16_frames, px * 21 / a.us_per_16_frames / 1e6 / 8))


if __name__ == "__main__":
    main()
