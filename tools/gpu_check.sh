#!/bin/bash
# The GPU-side check run through gpurun during development: all GPU tests,
# smoke(), and a bench line with its per-launch-group breakdown.
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_last.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_last.json'))
print(d["value"], d["ms_per_step"])
for k,v in d["kernels"].items(): print("  ", k, v["avg_ms_per_launch"], v["launches"], v["share_of_step"], v.get("achieved_GBs"))
PY
