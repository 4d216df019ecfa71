#!/bin/bash
# What the counting sort of the with-reference bands buys today: serial kernel traces with the work-class weights at
# their defaults and at zero (ODHIP_SORT_W=0,0,0: every key equal, the stable sort keeps the natural block order).
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r5_sortw}
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export ODHIP_LIB=$GRAFT_REPO_ROOT/daala_amd/lib/libdaalahip_exp.so
for v in default zero; do
  if [ $v = zero ]; then export ODHIP_SORT_W=0,0,0; else unset ODHIP_SORT_W; fi
  PO=$OUT/prof_$v; rm -rf $PO; mkdir -p $PO
  ( cd /tmp && export TMPDIR=/tmp && ODHIP_PVQ_SERIAL=1 rocprofv3 --kernel-trace --stats -d $PO/trace -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-streaming --no-replay > $PO/out.json 2> $PO/trace.err )
  DB=$(find $PO/trace -name "*.db" | head -1)
  python tools/prof_summary.py $DB 0.0 > $OUT/summary_$v.txt 2>&1
  echo "== $v"; grep -E "k_refb_lean|k_refb_prep|k_refb_(hist|scatter|prefix)" $OUT/summary_$v.txt
done
