#!/bin/bash
# Development loop of the with-reference band stage on the GPU box: parity
# tests, the bench line, and a serialised kernel trace of it.
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_pvq_refbands.py -x -q 2>&1 | tail -8
timeout 300 python bench.py --no-cpu-baseline --steps 5 --warmup 1 2>&1 | tail -1 > gpurun_out/bench_cfl.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/bench_cfl.json"))
print(d["value"], d["ms_per_step"])
for k,v in d["kernels"].items(): print("  ", k, v["avg_ms_per_launch"], v["launches"], v["share_of_step"])
PY
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_cfl; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export ODHIP_PVQ_SERIAL=1
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench.json 2> $OUT/trace.err
DB=$(find $OUT/trace -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/prof_summary.py $DB 0.5 | grep "k_refb\|total\|^kernel"
