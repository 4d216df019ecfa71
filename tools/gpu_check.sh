timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py 2>&1 | tail -1 > gpurun_out/bench_last.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_last.json'))
print(d["value"], d["ms_per_step"], d["steps"], d["warmup"])
print(json.dumps(d["roofline"])[:400])
print(json.dumps(d.get("cpu_baseline"))[:300], d.get("speedup_vs_cpu_baseline"))
for k,v in d["kernels"].items(): print("  ", k, v["avg_ms_per_launch"], v["launches"], v["share_of_step"], v.get("achieved_GBs"))
PY
