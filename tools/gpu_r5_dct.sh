#!/bin/bash
# Round-5 call for the standalone transforms: parity tests of the per-size batch / plane transforms, then configs[2] / [3].
set -u
TAG=${1:-r5_dct}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log
tail -5 $OUT/pytest.log
timeout 600 python tools/microbench.py > $OUT/microbench.txt 2> $OUT/microbench.err
head -14 $OUT/microbench.txt
