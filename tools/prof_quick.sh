#!/bin/bash
# Quick kernel-trace of a short bench run with the band-stage streams serialised
# (ODHIP_PVQ_SERIAL=1), so that per-kernel durations do not overlap.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_quick
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export ODHIP_PVQ_SERIAL=1
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench.json 2> $OUT/trace.err
DB=$(find $OUT/trace -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/prof_summary.py $DB 0.2
