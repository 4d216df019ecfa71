#!/bin/bash
# Round-5 GPU call: the whole GPU suite, then the default bench line (with the operating-point sweep and streaming_io).
set -u
TAG=${1:-r5_full}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  timeout ${TEST_TIMEOUT:-1500} python -m pytest tests -m gpu -x -q ${PYTEST_ARGS:-} > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log
  tail -15 $OUT/pytest.log
fi
if [ "${SKIP_BENCH:-0}" != "1" ]; then
  timeout 900 python bench.py ${BENCH_ARGS:-} > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
  tail -3 $OUT/bench.err
  python - $OUT/bench.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print({k:d.get(k) for k in ("value","ms_per_step","verified","pipelined_equals_serial","speedup_vs_cpu_baseline")})
    for k in ("roofline_filter_dct","roofline_filter_dct_chroma","roofline_inverse_luma","roofline_inverse_chroma","roofline_filter_dct_stage"):
        print(k, d[k].get("frac"), d[k].get("avg_ms_per_launch", d[k].get("ms_per_step_alone")))
    print("roofline", d["roofline"]["kernel"][:40], d["roofline"]["frac"], d["roofline"]["avg_ms_per_launch"])
    for k,v in d.get("kernels",{}).items(): print("   ", k, v.get("avg_ms_per_launch"), v.get("exclusive_avg_ms"))
    print("streaming_input", d.get("streaming_input",{}) and {k:d["streaming_input"].get(k) for k in ("value","ms_per_step","h2d_GBs")})
    print("streaming_io", d.get("streaming_io") and {k:d["streaming_io"].get(k) for k in ("value","ms_per_step","d2h_bytes_per_step","d2h_GBs")})
    for e in d.get("quality_sweep") or []:
        print("sweep", e["content"], e["quality"], "ms %.3f" % e["ms_per_step"], "value %.3e" % e["value"], "verified", e["verified"], "meanK", e.get("mean_k_per_band"), "reruns", e["theta_margin_reruns"], e["price_margin_reruns"])
except Exception as e:
    print("bench parse failed", repr(e))
PY
fi
