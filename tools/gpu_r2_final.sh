#!/bin/bash
# Round-2 evidence run (through gpurun): $1 = tests | bench | prof
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2_final
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
case "${1:-tests}" in
tests)
  timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log
  tail -6 $OUT/pytest.log
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/smoke.log
  tail -2 $OUT/smoke.log
  ;;
bench)
  timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"
  timeout 600 python bench.py --content natural > $OUT/bench_natural.json 2> $OUT/bench_natural.err; echo "natural rc=$?"
  timeout 600 python bench.py --chroma-noref > $OUT/bench_noref.json 2> $OUT/bench_noref.err; echo "noref rc=$?"
  timeout 300 python tools/microbench_dering.py > $OUT/microbench_dering.txt 2>&1; echo "dering rc=$?"
  timeout 900 python tools/e2e_integrated.py 3 > $OUT/e2e_integrated.txt 2>&1; echo "e2e rc=$?"
  python - <<'PY'
import json,os
for n in ("default","natural","noref"):
    p=os.path.join(os.environ["GRAFT_REPO_ROOT"],"gpurun_out/r2_final/bench_%s.json"%n)
    try:
        d=json.loads(open(p).read().strip().splitlines()[-1])
        print(n,{k:d.get(k) for k in ("value","ms_per_step","verified","pipelined_equals_serial","price_margin_reruns","speedup_vs_cpu_baseline")})
        print("  cpu",d.get("cpu_baseline",{}).get("value"),(d.get("cpu_baseline",{}).get("simd_build") or {}).get("value"),d.get("cpu_baseline_all_cores",{}).get("value"))
        print("  roof",d["roofline"]["kernel"][:28],d["roofline"]["frac"],"fd",d["roofline_filter_dct"]["frac"])
    except Exception as e:
        print(n,"parse failed",e)
PY
  tail -4 $OUT/microbench_dering.txt; tail -7 $OUT/e2e_integrated.txt
  ;;
prof)
  bash tools/profile_round.sh r2
  ;;
esac
