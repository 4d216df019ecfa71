#!/bin/bash
# Round-2 GPU check (run through gpurun): the GPU test suite, smoke(), one default bench line.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2_check
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log
tail -15 $OUT/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/smoke.log
tail -3 $OUT/smoke.log
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
tail -5 $OUT/bench.err
python - <<'PY'
import json,os
p=os.path.join(os.environ["GRAFT_REPO_ROOT"],"gpurun_out/r2_check/bench.json")
try:
    d=json.loads(open(p).read().strip().splitlines()[-1])
    keep={k:d[k] for k in ("value","ms_per_step","host_launch_ms_per_step","host_wait_ms_total","pipelined_equals_serial","verified","theta_margin_reruns") if k in d}
    print(keep)
    print("roofline", {k:d["roofline"].get(k) for k in ("kernel","bound","achieved","peak","frac","avg_ms_per_launch","avg_ms_per_launch_in_step")})
    print("filter_dct", {k:d["roofline_filter_dct"].get(k) for k in ("achieved","frac","avg_ms_per_launch","copy_1GiB_GBs")})
    print("cpu", d.get("cpu_baseline",{}).get("value"), d.get("cpu_baseline_all_cores",{}).get("value"))
    print("verification", d.get("verification"))
    for k,v in d["kernels"].items(): print(k, v["avg_ms_per_launch"], v.get("exclusive_avg_ms"))
except Exception as e:
    print("bench parse failed", e)
PY
