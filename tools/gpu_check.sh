timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_last.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_last.json'))
print(d["value"], d["ms_per_step"])
for k,v in d["kernels"].items(): print("  ", k, v["avg_ms_per_launch"], v["launches"], v["share_of_step"], v.get("achieved_GBs"))
PY
