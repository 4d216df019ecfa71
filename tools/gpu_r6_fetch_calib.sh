#!/bin/bash
# Round 6: calibrate rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access patterns of this library
# (tools/ubench/fetch_calib.hip); separate --pmc passes as the guide prescribes.
set -u
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6_fetch_calib
mkdir -p $OUT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 $GRAFT_REPO_ROOT/tools/ubench/fetch_calib.hip -o /tmp/fetch_calib || exit 1
/tmp/fetch_calib > $OUT/plain.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/fc_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/fc_$c -o t -- /tmp/fetch_calib > $OUT/$c.log 2>&1
  cp $(find /tmp/fc_$c -name "*counter_collection.csv" | head -1) $OUT/$c.csv
done
python - $OUT <<'PY'
import csv, sys, re, collections
o = sys.argv[1]
pat = {}
for l in open(o + "/plain.txt"):
    m = re.match(r"(\S+)\s+elements\s+(\d+) bytes\s+(\d+) stride\s+(\d+)\s+([\d.]+) ms", l)
    if m:
        pat[m.group(1)] = (int(m.group(2)), int(m.group(3)), int(m.group(4)), float(m.group(5)))
cnt = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(o + "/" + c + ".csv")):
        n = re.sub(r"\(.*$", "", r["Kernel_Name"])
        if r["Counter_Name"] == c:
            acc[n].append(float(r["Counter_Value"]))
    for n, v in acc.items():
        cnt[n][c] = sum(v)/len(v)
def touched(n, b, s, x):
    # bytes of the distinct x-byte blocks the pattern touches (first 8192 elements, scaled)
    m = min(n, 8192)
    blocks = set()
    for i in range(m):
        for a in range(i*s, i*s + b, 4):
            blocks.add(a//x)
    return len(blocks)*x*(n/m)
print("%-16s %9s %9s %9s %9s | %10s %10s | %s" % ("kernel", "useful", "sect32", "line64", "line128", "FETCH_KiB", "WRITE_KiB",
                                                  "FETCH*1024 / useful, /sect32, /line64, /line128   (WRITE likewise)"))
for n, (ne, b, s, ms) in pat.items():
    u = ne*b
    t32, t64, t128 = (touched(ne, b, s, x) for x in (32, 64, 128))
    f = cnt.get(n, {}).get("FETCH_SIZE", 0.)*1024
    w = cnt.get(n, {}).get("WRITE_SIZE", 0.)*1024
    v = f if n.startswith("r_") else w
    print("%-16s %8.0fM %8.0fM %8.0fM %8.0fM | %10.0f %10.0f | %.3f %.3f %.3f %.3f   (%.3f ms, other counter / useful %.3f)" % (
        n, u/1e6, t32/1e6, t64/1e6, t128/1e6, f/1024, w/1024, v/u, v/t32, v/t64, v/t128, ms, (w if n.startswith("r_") else f)/u))
PY
