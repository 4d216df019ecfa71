#!/bin/bash
# Round-5 inverse-stage call: a test subset, then the filter + DCT stages timed alone for kernel variants
# (name:ENV=..,ENV=..;name2:...), with digests.  Variants whose name starts with "exp" use the experiments library.
set -u
TAG=${1:-r5_inv}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
EXPLIB=$GRAFT_REPO_ROOT/daala_amd/lib/libdaalahip_exp.so
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  timeout 900 python -m pytest ${TESTS:-tests/test_gpu_parity.py tests/test_gpu_pipeline.py tests/test_gpu_fpr.py tests/test_gpu_decode_check.py} -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log
  tail -8 $OUT/pytest.log
fi
run() {  # name, env...
  local name=$1; shift
  ( export "$@" ODHIP_DUMMY=1; case $name in exp*) export ODHIP_LIB=$EXPLIB;; esac; timeout 300 python tools/stage_times.py --tag "$name" ${STAGE_ARGS:-} ) > $OUT/stage_$name.json 2> $OUT/stage_$name.err
  echo "== $name: $*"; tail -1 $OUT/stage_$name.json | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read())
    print('  ', ' '.join('%s=%.1fus(%.3f)' % (k.replace('dequant_','').replace('forward_','').replace('image_copy_',''), v['ms']*1e3, v['frac']) for k,v in d['stages'].items()))
    print('   whole %.1f us frac %.3f  recon %s levels %s' % (d['whole_stage']['ms']*1e3, d['whole_stage']['frac'], d['recon_digest'], d['levels_digest']))
except Exception as e:
    print('   failed', e)
"
  tail -2 $OUT/stage_$name.err
}
IFS=';' read -ra VARIANTS <<< "${STAGE_VARIANTS:-new:ODHIP_X=0;exp_leaf4_off:ODHIP_INVERSE_DBG=8}"
for v in "${VARIANTS[@]}"; do
  name=${v%%:*}
  envs=${v#*:}
  IFS=',' read -ra EV <<< "$envs"
  run $name "${EV[@]}"
done
if [ "${TRACE:-0}" = "1" ]; then
  PO=$OUT/prof; rm -rf $PO; mkdir -p $PO
  ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $PO/trace -o t -- python $GRAFT_REPO_ROOT/tools/stage_times.py --n 5 > $PO/out.json 2> $PO/trace.err )
  DB=$(find $PO/trace -name "*.db" | head -1)
  python tools/prof_summary.py $DB 0.0 > $OUT/prof_summary.txt 2>&1
  grep -i "inverse\|edge\|pyramid\|img" $OUT/prof_summary.txt | head -30
fi
if [ "${PMC:-0}" = "1" ]; then
  PO=$OUT/pmc; rm -rf $PO; mkdir -p $PO
  ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES --kernel-trace --output-format csv -d $PO -o t -- python $GRAFT_REPO_ROOT/tools/stage_times.py --n 2 > $PO/out.json 2> $PO/err.log )
  python - $PO <<'PY'
import csv, collections, glob, re, sys
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1]+'/**/t_counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n=r['Kernel_Name'].replace('(anonymous namespace)::','').replace('void ','')
        n=re.sub(r'\(.*$','',n)
        if re.search('inverse|edge|pyramid', n):
            acc[(n, r.get('Grid_Size',''))][r['Counter_Name']].append(float(r['Counter_Value']))
print('== pmc')
for k,v in sorted(acc.items()):
    print('  %-50s %-10s' % (k[0][:50], k[1]), ' '.join('%s=%.0f' % (c.replace('SQ_',''), sum(x)/len(x)) for c,x in sorted(v.items())))
PY
fi
