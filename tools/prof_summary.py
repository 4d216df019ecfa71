#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace result (rocpd .db) per kernel and grid.

usage: prof_summary.py <results.db> [min_share_percent]
Prints name, grid, workgroup, LDS, VGPRs, launches, average/min/max duration
and share of total GPU kernel time - the table committed under profiles/."""
import collections
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z0-9_:]+(<[^(]*>)?)", name)
    s = m.group(1) if m else name
    return s[:70]


def main():
    db = sqlite3.connect(sys.argv[1])
    minshare = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
    rows = list(db.execute(
        "select name, grid_x, grid_y, grid_z, workgroup_x, lds_size, vgpr_count, "
        "accum_vgpr_count, (end - start) from kernels"))
    agg = collections.defaultdict(list)
    for name, gx, gy, gz, wx, lds, vg, ag, dur in rows:
        agg[(short(name), gx, gy, gz, wx, lds, vg + ag)].append(dur)
    tot = sum(sum(v) for v in agg.values())
    print("%-70s %-22s %5s %6s %5s %6s %10s %10s %10s %6s" % (
        "kernel", "grid(x,y,z)", "wg", "lds", "vgpr", "calls", "avg_us", "min_us", "max_us", "share"))
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        share = 100.0 * sum(v) / tot
        if share < minshare:
            continue
        print("%-70s %-22s %5d %6d %5d %6d %10.1f %10.1f %10.1f %5.1f%%" % (
            k[0], "%d,%d,%d" % (k[1], k[2], k[3]), k[4], k[5], k[6], len(v),
            sum(v) / len(v) / 1e3, min(v) / 1e3, max(v) / 1e3, share))
    print("total kernel time: %.3f ms over %d dispatches" % (tot / 1e6, len(rows)))


if __name__ == "__main__":
    main()
