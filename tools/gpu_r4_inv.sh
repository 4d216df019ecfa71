#!/bin/bash
# (round 5) the A/B switches below exist in the experiments build of the library only
export ODHIP_LIB=${ODHIP_LIB:-$(cd "$(dirname "$0")/.." && pwd)/daala_amd/lib/libdaalahip_exp.so}
# Round-4 GPU call: the GPU test suite with the walking inverse kernels, then the filter + DCT stages
# timed alone for the kernel variants named by environment knobs, a kernel trace, and short bench lines.
set -u
TAG=${1:-r4_inv}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
if [ "${SKIP_TESTS:-0}" != "1" ]; then
timeout 900 python -m pytest tests -m gpu -x -q ${PYTEST_ARGS:-} > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log
tail -15 $OUT/pytest.log
fi
run() {  # name, env...
  local name=$1; shift
  ( export "$@" ODHIP_DUMMY=1; timeout 300 python tools/stage_times.py --tag "$name" ) > $OUT/stage_$name.json 2> $OUT/stage_$name.err
  echo "== $name: $*"; tail -1 $OUT/stage_$name.json | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read())
    print('  ', ' '.join('%s=%.1fus(%.2f)' % (k.replace('dequant_','').replace('forward_','').replace('image_copy_',''), v['ms']*1e3, v['frac']) for k,v in d['stages'].items()))
    print('   whole %.1f us frac %.3f  recon %s levels %s' % (d['whole_stage']['ms']*1e3, d['whole_stage']['frac'], d['recon_digest'], d['levels_digest']))
except Exception as e:
    print('   failed', e)
"
  tail -2 $OUT/stage_$name.err
}
IFS=';' read -ra VARIANTS <<< "${STAGE_VARIANTS:-new:ODHIP_X=0;old:ODHIP_INVERSE_OLD=1}"
for v in "${VARIANTS[@]}"; do
  name=${v%%:*}
  envs=${v#*:}
  run $name $envs
done
if [ "${SKIP_TRACE:-0}" != "1" ]; then
  PO=$OUT/prof; rm -rf $PO; mkdir -p $PO
  ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $PO/trace -o t -- python $GRAFT_REPO_ROOT/tools/stage_times.py --n 5 > $PO/out.json 2> $PO/trace.err )
  DB=$(find $PO/trace -name "*.db" | head -1)
  python tools/prof_summary.py $DB 0.0 > $OUT/prof_summary.txt 2>&1
  grep -i "inverse\|edge\|pyramid\|img" $OUT/prof_summary.txt | head -30
fi
if [ "${SKIP_BENCH:-0}" != "1" ]; then
  for m in new old; do
    if [ $m = old ]; then export ODHIP_INVERSE_OLD=1; else unset ODHIP_INVERSE_OLD; fi
    timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-streaming > $OUT/bench_$m.json 2> $OUT/bench_$m.err
    python - $OUT/bench_$m.json $m <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], 'ms_per_step', round(d['ms_per_step'],3), 'value %.3e' % d['value'], 'pipe==serial', d.get('pipelined_equals_serial'))
    for k,v in d.get('kernels',{}).items(): print('   ', k, v.get('avg_ms_per_launch'), v.get('exclusive_avg_ms'))
except Exception as e:
    print('bench parse failed', e)
PY
  done
fi
if [ "${PMC:-0}" = "1" ]; then
  # VALU / LDS counters of the filter + DCT kernels, new and old inverse kernels (one pass each)
  for m in new old; do
    if [ $m = old ]; then export ODHIP_INVERSE_OLD=1; else unset ODHIP_INVERSE_OLD; fi
    PO=$OUT/pmc_$m; rm -rf $PO; mkdir -p $PO
    ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES --kernel-trace --output-format csv -d $PO -o t -- python $GRAFT_REPO_ROOT/tools/stage_times.py --n 2 > $PO/out.json 2> $PO/err.log )
    python - $PO $m <<'PY'
import csv, collections, glob, re, sys
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1]+'/**/t_counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n=r['Kernel_Name'].replace('(anonymous namespace)::','').replace('void ','')
        n=re.sub(r'\(.*$','',n)
        if re.search('inverse|edge|pyramid', n):
            acc[(n, r.get('Grid_Size',''))][r['Counter_Name']].append(float(r['Counter_Value']))
print('== pmc', sys.argv[2])
for k,v in sorted(acc.items()):
    print('  %-50s %-10s' % (k[0][:50], k[1]), ' '.join('%s=%.0f' % (c.replace('SQ_',''), sum(x)/len(x)) for c,x in sorted(v.items())))
PY
  done
  unset ODHIP_INVERSE_OLD
fi
if [ "${FULLBENCH:-0}" = "1" ]; then
  timeout 600 python bench.py > $OUT/bench_full.json 2> $OUT/bench_full.err; echo "bench rc=$?"
  tail -3 $OUT/bench_full.err
  python - $OUT/bench_full.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print({k:d.get(k) for k in ("value","ms_per_step","verified","pipelined_equals_serial","speedup_vs_cpu_baseline")})
    print("verification", {k:d["verification"].get(k) for k in ("frames_compared","planes_levels_compared","mismatches")}, d["verification"].get("decisions",{}).get("bands_compared"), d["verification"].get("decisions",{}).get("mismatches"))
    for k in ("roofline_filter_dct","roofline_filter_dct_chroma","roofline_inverse_luma","roofline_inverse_chroma","roofline_filter_dct_stage"):
        print(k, d[k].get("frac"), d[k].get("avg_ms_per_launch", d[k].get("ms_per_step_alone")))
    print("roofline", d["roofline"]["kernel"][:40], d["roofline"]["frac"], d["roofline"]["avg_ms_per_launch"])
except Exception as e:
    print("bench parse failed", repr(e))
PY
fi
