#!/bin/bash
# quick perf check: optional test filter ($1), then bench lines without the CPU legs
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/quick
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
if [ -n "${1:-}" ]; then
  timeout 600 python -m pytest tests -m gpu -x -q -k "$1" 2>&1 | tail -4
fi
for extra in "" "--frames 8"; do
  timeout 300 python bench.py --no-cpu-baseline $extra > $OUT/b.json 2> $OUT/b.err; echo "bench rc=$?"
  python - <<'PY'
import json,os
p=os.path.join(os.environ["GRAFT_REPO_ROOT"],"gpurun_out/quick/b.json")
d=json.loads(open(p).read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step","host_launch_ms_per_step","host_wait_ms_per_step","pipelined_equals_serial")})
print("filter_dct", d["roofline_filter_dct"]["avg_ms_per_launch"], d["roofline_filter_dct"]["frac"])
print("roof", d["roofline"]["kernel"][:24], d["roofline"]["frac"], d["roofline"]["avg_ms_per_launch"])
print(" ".join("%s=%.3f/%.3f" % (k[:18], v["avg_ms_per_launch"], v.get("exclusive_avg_ms",0)) for k,v in d["kernels"].items()))
PY
done
