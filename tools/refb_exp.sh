#!/bin/bash
# timing experiment: with-reference searches with the K-pulse search skipped
cd /tmp && export TMPDIR=/tmp
export ODHIP_PVQ_SERIAL=1
for mode in normal nosearch; do
OUT=$GRAFT_REPO_ROOT/gpurun_out/exp_$mode; rm -rf $OUT; mkdir -p $OUT
if [ $mode = nosearch ]; then export ODHIP_REF_DEBUG_NOSEARCH=1; fi
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench.json 2> $OUT/trace.err
DB=$(find $OUT/trace -name "*.db" | head -1)
echo "== $mode"
python $GRAFT_REPO_ROOT/tools/prof_summary.py $DB 0.5 | grep "k_refb_search\|total"
done
