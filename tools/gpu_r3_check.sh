#!/bin/bash
# Round-3 GPU check (run through gpurun): the GPU test suite, smoke(), one default bench
# line, then short A/B bench lines for the knobs named in $ABS ("VAR=val VAR=val;VAR=val").
set -u
TAG=${1:-r3_check}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
if [ "${SKIP_TESTS:-0}" != "1" ]; then
timeout 1200 python -m pytest tests -m gpu -x -q ${PYTEST_ARGS:-} > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log
tail -25 $OUT/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/smoke.log
tail -3 $OUT/smoke.log
fi
summ() {
python - "$1" <<'PY'
import json,sys
p=sys.argv[1]
try:
    d=json.loads(open(p).read().strip().splitlines()[-1])
    keep={k:d[k] for k in ("value","ms_per_step","host_launch_ms_per_step","pipelined_equals_serial","verified","theta_margin_reruns","price_margin_reruns") if k in d}
    print(keep)
    if "roofline" in d: print("roofline", {k:d["roofline"].get(k) for k in ("kernel","bound","frac","avg_ms_per_launch","avg_ms_per_launch_in_step")})
    if "roofline_filter_dct" in d: print("filter_dct", {k:d["roofline_filter_dct"].get(k) for k in ("achieved","frac","avg_ms_per_launch","copy_1GiB_GBs")})
    if d.get("cpu_baseline"): print("cpu", d.get("cpu_baseline",{}).get("value"), (d.get("cpu_baseline_all_cores") or {}).get("value"))
    for k,v in d.get("kernels",{}).items(): print("  ", k, v.get("avg_ms_per_launch"), v.get("exclusive_avg_ms"))
except Exception as e:
    print("bench parse failed", e)
PY
}
if [ "${SKIP_BENCH:-0}" != "1" ]; then
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
tail -5 $OUT/bench.err
summ $OUT/bench.json
fi
IFS=';' read -ra VARIANTS <<< "${ABS:-}"
i=0
for v in "${VARIANTS[@]}"; do
  i=$((i+1))
  echo "== variant $i: $v"
  env $v timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-streaming > $OUT/ab_$i.json 2> $OUT/ab_$i.err; echo "rc=$?"
  tail -2 $OUT/ab_$i.err
  summ $OUT/ab_$i.json | head -2
done
