#!/bin/bash
# SQ occupancy / wait counters for kernels matching $1 (regex) in a short bench run.
cd /tmp && export TMPDIR=/tmp
export ODHIP_PVQ_SERIAL=1
O=$GRAFT_REPO_ROOT/gpurun_out/pmc_k; rm -rf $O; mkdir -p $O
B="python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline"
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" "GRBM_GUI_ACTIVE SQ_WAVES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/g$i -o t -- $B > $O/g$i.log 2>&1 || tail -3 $O/g$i.log
done
python - "$1" <<'PY'
import csv, collections, os, glob, re, sys
root=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/pmc_k'
pat=re.compile(sys.argv[1])
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(root+'/g*/t_counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        n=r['Kernel_Name'].replace('(anonymous namespace)::','').replace('void ','')
        n=re.sub(r'\(.*$','',n)
        if pat.search(n):
            acc[n][r['Counter_Name']].append(float(r['Counter_Value']))
names=sorted({c for k in acc.values() for c in k})
for k,v in sorted(acc.items()):
    print(k)
    for c in names:
        if c in v: print("    %-24s %16.0f"%(c, sum(v[c])/len(v[c])))
PY
