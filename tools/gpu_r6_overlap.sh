#!/bin/bash
# Round 6, item 1: how the two chains of the step share the GPU.  One box, alternating runs.
#   (a) a kernel trace of the default step with start / end / queue of every dispatch (the timeline the
#       A/Bs below are read against): gpurun_out/r6_overlap/timeline.csv
#   (b) ODHIP_PIPE_CUSPLIT 1..7 (luma chain n of every 8 CUs, chroma chain the rest), ODHIP_PIPE_PRIO 1 / 2
#       (queue priorities), ODHIP_PIPE_FORK masks - experiments build
set -u
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6_overlap
mkdir -p $OUT
EXP=$GRAFT_REPO_ROOT/daala_amd/lib/libdaalahip_exp.so
B="python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-streaming"
run() {
  $B 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
k=d['kernels']
print('%-14s ms_per_step %.3f  pipe==serial %s  pyr_in_step %.3f  wait %.3f launch %.3f' % ('$1', d['ms_per_step'], d.get('pipelined_equals_serial'), d['roofline_filter_dct']['in_step']['avg_ms_per_launch'], d['host_wait_ms_per_step'], d['host_launch_ms_per_step']))
"
}
{
for rep in 1 2; do
  run default
  export ODHIP_LIB=$EXP
  run exp_default
  for n in 1 2 3 4 5 6 7; do ODHIP_PIPE_CUSPLIT=$n run cusplit_$n; done
  ODHIP_PIPE_PRIO=1 run prio_luma
  ODHIP_PIPE_PRIO=2 run prio_chroma
  ODHIP_PIPE_FORK=1 run fork_luma
  unset ODHIP_LIB
done
} 2>&1 | tee $OUT/ab.txt
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-replay --no-streaming > $OUT/trace_bench.json 2> $OUT/trace.err )
find $OUT/trace -name "*kernel_trace.csv" | head -3
F=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python - "$F" > $OUT/timeline.csv <<'EOF'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n)
    m = re.match(r"([A-Za-z0-9_:]+(<[^(]*>)?)", n); return (m.group(1) if m else n)[:48]
t0 = min(int(r["Start_Timestamp"]) for r in rows)
print("queue,stream,kernel,start_us,end_us,dur_us,lds,vgpr,wg,grid")
for r in sorted(rows, key=lambda r: int(r["Start_Timestamp"])):
    s = (int(r["Start_Timestamp"]) - t0) / 1e3; e = (int(r["End_Timestamp"]) - t0) / 1e3
    print("%s,%s,%s,%.1f,%.1f,%.1f,%s,%s,%s,%s" % (r.get("Queue_Id"), r.get("Stream_Id", ""), short(r["Kernel_Name"]).replace(",", ";"), s, e, e - s,
          r.get("LDS_Block_Size"), r.get("VGPR_Count"), r.get("Workgroup_Size_X", r.get("Workgroup_Size")), r.get("Grid_Size_X", r.get("Grid_Size"))))
EOF
wc -l $OUT/timeline.csv
gzip -c "$F" > $OUT/kernel_trace.csv.gz
rm -rf $OUT/trace
