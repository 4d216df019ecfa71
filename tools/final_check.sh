#!/bin/bash
# Round-end check on the GPU box: all GPU tests, smoke(), the default bench line.
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py 2>&1 | tail -1 > gpurun_out/bench_default.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_default.json'))
print(d["value"], d["ms_per_step"])
print(json.dumps(d["roofline"])[:1200])
for k,v in d["kernels"].items(): print("  ", k, v["avg_ms_per_launch"], v["launches"], v["share_of_step"])
print(d.get("cpu_baseline"))
print(d.get("speedup_vs_cpu_baseline"))
PY
