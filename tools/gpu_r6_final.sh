#!/bin/bash
# Round-6 final GPU evidence: full GPU test suite + smoke, the rocprofv3 set (tools/profile_round.sh),
# the bench lines (default = the driver's command, natural content, chroma without reference),
# configs[2] / configs[3] microbenchmarks, configs[4] with one encoder process.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6_final
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
bash tools/profile_round.sh r6 > $OUT/profile_round.log 2>&1; echo "profile rc=$?"
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py gpurun_out/prof_r6 $OUT/r6_pmc_traffic.json > $OUT/r6_rocprofv3_summary.txt 2> $OUT/pmc_summary.err; echo "pmc summary rc=$?"
cp $OUT/r6_pmc_traffic.json profiles/r6_pmc_traffic.json     # so that the bench lines below carry the counters
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench default rc=$?"
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 3 > $OUT/bench_driver_cmd.json 2> $OUT/bench_driver_cmd.err; echo "bench driver rc=$?"
timeout 600 python bench.py --content natural --no-cpu-allcores > $OUT/bench_natural.json 2> $OUT/bench_natural.err; echo "bench natural rc=$?"
timeout 600 python bench.py --chroma-noref --no-cpu-allcores > $OUT/bench_noref.json 2> $OUT/bench_noref.err; echo "bench noref rc=$?"
timeout 600 python tools/microbench.py > $OUT/microbench_configs2_3.txt 2>&1; echo "microbench rc=$?"
# configs[4] at its stated size: 300 frames, encoder threads sized from the host quota, EVERY packet compared with the plain C encoder
timeout 1500 python bench.py --encode-frames 300 --encode-check 300 > $OUT/encode_mode_300frames.json 2> $OUT/encode_300.err; echo "encode 300 rc=$?"; tail -2 $OUT/encode_300.err
python - $OUT <<'PY'
import json,sys,os
o=sys.argv[1]
for n in ("bench_default","bench_driver_cmd","bench_natural","bench_noref"):
    try:
        d=json.loads(open(os.path.join(o,n+".json")).read().strip().splitlines()[-1])
        print(n, "value %.4g ms/step %.3f verified %s pipe==serial %s" % (d["value"], d["ms_per_step"], d.get("verified"), d.get("pipelined_equals_serial")),
              "roofline", d["roofline"]["kernel"][:24], d["roofline"]["bound"], d["roofline"]["frac"],
              "| fd", d["roofline_filter_dct"]["frac"], "fdc", d["roofline_filter_dct_chroma"]["frac"], "invL", d["roofline_inverse_luma"]["frac"], "invC", d["roofline_inverse_chroma"]["frac"], "stage", d["roofline_filter_dct_stage"]["frac"],
              "| cpu", (d.get("cpu_baseline") or {}).get("value"), "x", d.get("speedup_vs_cpu_baseline"))
    except Exception as e:
        print(n, "failed", repr(e))
try:
    d=json.loads(open(os.path.join(o,"encode_mode_300frames.json")).read().strip().splitlines()[-1])
    print("encode 300", d["value"], d["encoder_threads_per_process"], d["host_cpu_quota"], d["prefix_check"], d["rank0"])
except Exception as e:
    print("encode 300 failed", repr(e))
PY
tail -12 $OUT/microbench_configs2_3.txt
