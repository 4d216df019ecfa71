#!/usr/bin/env python3
"""Merge the rocprofv3 CSVs of tools/profile_round.sh into one text summary.

usage: pmc_summary.py gpurun_out/prof_<tag>  > profiles/<tag>_summary.txt

Per kernel (our k_* kernels only): launches, average duration from the kernel
trace, and per-launch averages of the PMC counters.  FETCH_SIZE / WRITE_SIZE
are in KiB as reported; the HBM column applies the gfx950 correction from
MI355X_MICROARCH.md (FETCH_SIZE counts 128-byte requests as 64 B for wide
coalesced reads: doubled) and is a calibrated ESTIMATE, see DESIGN.md."""
import collections
import csv
import re
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name)


def trace_table(root, sub, title):
    dur = collections.defaultdict(list)
    grid = {}
    try:
        f = open(root + "/" + sub + "/t_kernel_trace.csv")
    except FileNotFoundError:
        return
    with f:
        for r in csv.DictReader(f):
            n = short(r["Kernel_Name"])
            if not n.startswith("k_"):
                continue
            key = (n, r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"]) if "Grid_Size_X" in r else (n, r.get("Grid_Size", ""), "", "")
            dur[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
            grid[key] = (r.get("Workgroup_Size_X", r.get("Workgroup_Size", "")), r.get("LDS_Block_Size", ""), r.get("VGPR_Count", ""))
    print(title)
    print("%-28s %-18s %5s %6s %5s %6s %10s %10s" % ("kernel", "grid", "wg", "lds", "vgpr", "calls", "avg_us", "min_us"))
    tot = sum(sum(v) for v in dur.values())
    for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
        g = grid[k]
        print("%-28s %-18s %5s %6s %5s %6d %10.1f %10.1f  %5.1f%%" % (
            k[0], ",".join(x for x in k[1:] if x), g[0], g[1], g[2], len(v), sum(v) / len(v) / 1e3,
            min(v) / 1e3, 100.0 * sum(v) / tot))
    print()


def exclusive_durations(root):
    """kernel name -> average duration (us) in the serialised trace."""
    acc = collections.defaultdict(list)
    try:
        with open(root + "/trace_serial/t_kernel_trace.csv") as f:
            for r in csv.DictReader(f):
                acc[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    except FileNotFoundError:
        pass
    return {k: sum(v) / len(v) / 1e3 for k, v in acc.items()}


def main():
    root = sys.argv[1]
    excl = exclusive_durations(root)
    trace_table(root, "trace_serial", "# kernel trace, ODHIP_PVQ_SERIAL=1 (one stream: exclusive durations), our kernels only")
    dur = collections.defaultdict(list)
    grid = {}
    with open(root + "/trace/t_kernel_trace.csv") as f:
        for r in csv.DictReader(f):
            n = short(r["Kernel_Name"])
            if not n.startswith("k_"):
                continue
            key = (n, r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"]) if "Grid_Size_X" in r else (n, r.get("Grid_Size", ""), "", "")
            dur[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
            grid[key] = (r.get("Workgroup_Size_X", r.get("Workgroup_Size", "")), r.get("LDS_Block_Size", ""), r.get("VGPR_Count", ""))
    ctr = collections.defaultdict(lambda: collections.defaultdict(list))
    for sub in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_wait"):
        try:
            with open(root + "/" + sub + "/t_counter_collection.csv") as f:
                for r in csv.DictReader(f):
                    n = short(r["Kernel_Name"])
                    if not n.startswith("k_"):
                        continue
                    ctr[(n, r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        except FileNotFoundError:
            pass
    print("# kernel trace (rocprofv3 --kernel-trace --stats) of the default bench command: the luma and the")
    print("# chroma chain overlap on their two streams, durations include the time kernels share the GPU")
    print("%-28s %-18s %5s %6s %5s %6s %10s %10s" % ("kernel", "grid", "wg", "lds", "vgpr", "calls", "avg_us", "min_us"))
    tot = sum(sum(v) for v in dur.values())
    for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
        g = grid[k]
        print("%-28s %-18s %5s %6s %5s %6d %10.1f %10.1f  %5.1f%%" % (
            k[0], ",".join(x for x in k[1:] if x), g[0], g[1], g[2], len(v), sum(v) / len(v) / 1e3,
            min(v) / 1e3, 100.0 * sum(v) / tot))
    print()
    print("# PMC counters, average per launch, ODHIP_PVQ_SERIAL=1 (separate passes: FETCH_SIZE | WRITE_SIZE | SQ_* | SQ_WAIT_*)")
    names = ["FETCH_SIZE", "WRITE_SIZE", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INSTS_VALU",
             "SQ_INSTS_LDS", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VALU",
             "SQ_WAIT_ANY", "SQ_WAVES"]
    print("%-28s %-10s " % ("kernel", "grid") + " ".join("%14s" % n[-14:] for n in names)
          + " %12s %9s" % ("HBM_est_MB", "bankconf%"))
    traffic = {}
    for k, c in sorted(ctr.items()):
        vals = [sum(c[n]) / len(c[n]) if c.get(n) else float("nan") for n in names]
        hbm = (2 * vals[0] + vals[1]) * 1024 / 1e6
        if hbm == hbm:
            traffic["%s grid %s" % k] = {"fetch_size_KiB": vals[0], "write_size_KiB": vals[1],
                                         "hbm_bytes_per_launch": int(hbm * 1e6),
                                         "valu_wave_instructions": vals[4],
                                         "exclusive_avg_us": excl.get(k[0])}
        bc = 100.0 * vals[2] / vals[3] if vals[3] == vals[3] and vals[3] else float("nan")
        print("%-28s %-10s " % (k[0], k[1]) + " ".join("%14.0f" % v for v in vals) + " %12.1f %9.2f" % (hbm, bc))
    if len(sys.argv) > 2:
        import json
        import os
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench
        with open(sys.argv[2], "w") as f:
            json.dump({"source_hash": bench.source_hash(),
                       "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), "
                                 "HBM bytes = (2*FETCH_SIZE + WRITE_SIZE) KiB, MI355X_MICROARCH.md gfx950 correction",
                       "kernels": traffic}, f, indent=1)


if __name__ == "__main__":
    main()
