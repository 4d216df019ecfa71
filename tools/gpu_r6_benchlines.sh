#!/bin/bash
# The four bench lines of the round (after profiles/r6_pmc_traffic.json is in place).
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6_final
mkdir -p $OUT; cd $GRAFT_REPO_ROOT
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench default rc=$?"
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 3 > $OUT/bench_driver_cmd.json 2> $OUT/bench_driver_cmd.err; echo "bench driver rc=$?"
timeout 600 python bench.py --content natural --no-cpu-allcores > $OUT/bench_natural.json 2> $OUT/bench_natural.err; echo "bench natural rc=$?"
timeout 600 python bench.py --chroma-noref --no-cpu-allcores > $OUT/bench_noref.json 2> $OUT/bench_noref.err; echo "bench noref rc=$?"
