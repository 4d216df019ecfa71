#!/bin/bash
# Collect the rocprofv3 evidence committed under profiles/ (run on the GPU box
# through gpurun).  Kernel trace and each PMC group are SEPARATE runs (TCC has
# 4 slots: FETCH_SIZE needs 3, WRITE_SIZE 2; see MI355X_MICROARCH.md).
#   trace         default bench command (band-stage kernels overlap on forked streams)
#   trace_serial  ODHIP_PVQ_SERIAL=1: one stream, exclusive per-kernel durations
#   pmc_*         ODHIP_PVQ_SERIAL=1 as well (counters are per dispatch)
set -u
TAG=${1:-r4}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-replay --no-streaming"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $B > $OUT/bench_trace.json 2> $OUT/trace.err
export ODHIP_PVQ_SERIAL=1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_serial -o t -- $B > $OUT/bench_trace_serial.json 2> $OUT/trace_serial.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o t -- $B > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o t -- $B > /dev/null 2> $OUT/pmc_write.err
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $OUT/pmc_sq -o t -- $B > /dev/null 2> $OUT/pmc_sq.err
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_wait -o t -- $B > /dev/null 2> $OUT/pmc_wait.err
find $OUT -name "*.csv" | head -40
