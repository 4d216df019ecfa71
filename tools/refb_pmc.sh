#!/bin/bash
# SQ counters of the with-reference kernels (one rocprofv3 --pmc pass, serialised streams)
cd /tmp && export TMPDIR=/tmp
export ODHIP_PVQ_SERIAL=1
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_refb; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $OUT -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/err.txt
python - <<'PY'
import csv, collections, os, re
root=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/pmc_refb"
c=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(root+"/t_counter_collection.csv")):
    n=re.sub(r"\(.*$","",r["Kernel_Name"].replace("(anonymous namespace)::","").replace("void ",""))
    if "refb" in n or "k_search" in n: c[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
names=["SQ_INSTS_VALU","SQ_WAVE_CYCLES","SQ_BUSY_CYCLES","SQ_WAIT_INST_ANY","SQ_ACTIVE_INST_VALU","SQ_INSTS_LDS","SQ_INSTS_VMEM_RD","SQ_INSTS_VMEM_WR"]
print("%-28s"%"kernel"+" ".join("%14s"%n[3:] for n in names))
for k,v in sorted(c.items()):
    print("%-28s"%k+" ".join("%14.0f"%(sum(v[n])/len(v[n]) if v.get(n) else float('nan')) for n in names))
PY
