timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py 2>&1 | tail -1 > gpurun_out/bench_default.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_default.json'))
print(d["value"], d["ms_per_step"])
print(json.dumps(d["roofline"])[:900])
print(d.get("speedup_vs_cpu_baseline"))
PY
