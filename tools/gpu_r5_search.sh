#!/bin/bash
# Round-5 search-stage call: the PVQ / pipeline tests, then short bench lines (kernel timings) for A/B.
set -u
TAG=${1:-r5_search}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  timeout 1200 python -m pytest ${TESTS:-tests/test_gpu_pvq_refbands.py tests/test_gpu_pvq_ref.py tests/test_gpu_pvq_bands.py tests/test_gpu_pipeline.py tests/test_gpu_fullsize.py} -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log
  tail -6 $OUT/pytest.log
fi
for rep in ${REPS:-1 2}; do
  timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-streaming ${BENCH_ARGS:-} > $OUT/bench_$rep.json 2> $OUT/bench_$rep.err
  python - $OUT/bench_$rep.json $rep <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], 'ms_per_step', round(d['ms_per_step'],3), 'value %.3e' % d['value'], 'pipe==serial', d.get('pipelined_equals_serial'))
    for k,v in d.get('kernels',{}).items(): print('   ', k, v.get('avg_ms_per_launch'), v.get('exclusive_avg_ms'))
    for k in ('roofline','roofline_noref_search','roofline_ref_search'):
        r=d.get(k)
        if r: print('   ', k, r['kernel'][:28], 'excl ms', r.get('avg_ms_per_launch'))
except Exception as e:
    print('bench parse failed', e)
PY
done
if [ "${TRACE:-0}" = "1" ]; then
  PO=$OUT/prof; rm -rf $PO; mkdir -p $PO
  ( cd /tmp && export TMPDIR=/tmp && ODHIP_PVQ_SERIAL=1 rocprofv3 --kernel-trace --stats -d $PO/trace -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-streaming --no-replay > $PO/out.json 2> $PO/trace.err )
  DB=$(find $PO/trace -name "*.db" | head -1)
  python tools/prof_summary.py $DB 0.0 > $OUT/prof_summary_serial.txt 2>&1
  head -40 $OUT/prof_summary_serial.txt
fi
