#!/bin/bash
# device-priced choice: its tests, then bench lines with and without pricing (no CPU legs)
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/price
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_pipeline.py -m gpu -x -q -s 2>&1 | tail -25
for extra in "" "--no-price"; do
  timeout 300 python bench.py --no-cpu-baseline $extra > $OUT/b.json 2> $OUT/b.err; echo "bench $extra rc=$?"
  tail -3 $OUT/b.err
  python - <<'PY'
import json,os
p=os.path.join(os.environ["GRAFT_REPO_ROOT"],"gpurun_out/price/b.json")
d=json.loads(open(p).read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step","host_launch_ms_per_step","host_wait_ms_per_step","pipelined_equals_serial","price_margin_reruns","theta_margin_reruns")})
print("roof", d["roofline"]["kernel"][:24], d["roofline"]["frac"], d["roofline"]["avg_ms_per_launch"])
print(" ".join("%s=%.3f/%.3f" % (k[:18], v["avg_ms_per_launch"], v.get("exclusive_avg_ms",0)) for k,v in d["kernels"].items()))
PY
done
