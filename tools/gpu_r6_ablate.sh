#!/bin/bash
# Round 6: what the with-reference preparation and k_refb_lean_lane<15> spend their time on (experiments build,
# ODHIP_REFB_ABL bits; serial kernel trace of a short bench run per variant; results are WRONG by design, timing only).
cd /tmp && export TMPDIR=/tmp
export ODHIP_LIB=$GRAFT_REPO_ROOT/daala_amd/lib/libdaalahip_exp.so
export ODHIP_PVQ_SERIAL=1
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6_ablate
mkdir -p $OUT
for abl in 0 1 2 4 7 8; do
  rm -rf $OUT/t$abl
  ODHIP_REFB_ABL=$abl rocprofv3 --kernel-trace --stats -d $OUT/t$abl -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-replay --no-streaming > /dev/null 2> $OUT/err$abl.txt
  DB=$(find $OUT/t$abl -name "*.db" | head -1)
  echo "== ODHIP_REFB_ABL=$abl"
  python $GRAFT_REPO_ROOT/tools/prof_summary.py $DB 0.2 | grep -E "k_refb_prep|k_refb_lean_lane|k_inverse_walk<32"
  rm -rf $OUT/t$abl
done
