#!/bin/bash
cd /tmp && export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=8
export ODHIP_LIB=$GRAFT_REPO_ROOT/daala_amd/lib_p1024/libdaalahip.so
for mode in resident fed; do
  rm -rf /tmp/tr_$mode
  rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/tr_$mode -o t -- python $GRAFT_REPO_ROOT/tools/export_trace.py $mode > /dev/null 2>&1
  python - /tmp/tr_$mode $mode <<'PY'
import csv, sys, glob, re
d, mode = sys.argv[1], sys.argv[2]
kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
mc = glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True)
rows = []
for r in csv.DictReader(open(kt)):
    n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]); n = re.sub(r"^void ", "", n)
    m = re.match(r"([A-Za-z0-9_:]+(<[^(]*>)?)", n)
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K q%s %s" % (r.get("Queue_Id"), (m.group(1) if m else n)[:40])))
if mc:
    for r in csv.DictReader(open(mc[0])):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY %s %s B" % (r.get("Direction"), r.get("Size", r.get("Bytes", "?")))))
rows.sort()
packs = [r for r in rows if "k_export_pack" in r[2]]
print("== %s: %d pack launches; durations us: %s" % (mode, len(packs), " ".join("%.0f" % ((e - s)/1e3) for s, e, _ in packs)))
# the window of the 9th pack launch (a luma one well after warm-up)
big = [p for p in packs if (p[1] - p[0]) > 0][8:12]
for s, e, nme in big[:2]:
    print("  window of a pack launch %.0f us:" % ((e - s)/1e3))
    for s2, e2, n2 in rows:
        if e2 > s and s2 < e and (e2 - s2) > 2000:
            print("     %8.0f .. %8.0f  (%7.0f us)  %s" % ((s2 - s)/1e3, (e2 - s)/1e3, (e2 - s2)/1e3, n2))
PY
done
