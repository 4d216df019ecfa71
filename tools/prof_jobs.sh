#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_jobs
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export ODHIP_PVQ_SERIAL=1
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python $GRAFT_REPO_ROOT/tools/prof_jobs.py > $OUT/out.txt 2> $OUT/trace.err
cat $OUT/out.txt | grep -v amdgpu
DB=$(find $OUT/trace -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/prof_summary.py $DB 0.05 | grep -E "k_prep|k_search|k_sort|^kernel"
