#!/bin/bash
# Round-4 GPU call: the shim refactor under the drop-in / shard / bench-multi tests, then BASELINE
# configs[4] at its stated size: 300 frames of 1080p, P encoder processes sharing one GPU.
set -u
TAG=${1:-r4_encode}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
nproc > $OUT/nproc.txt; echo "host cores: $(cat $OUT/nproc.txt)"
if [ "${SKIP_TESTS:-0}" != "1" ]; then
timeout 1500 python -m pytest ${PYTEST_FILES:-tests/test_gpu_dropin_encoder.py tests/test_gpu_shard_encode.py tests/test_gpu_bench_multi.py tests/test_gpu_decode_check.py tests/test_encoder_example.py} -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log
tail -15 $OUT/pytest.log
fi
N=${JOB_FRAMES:-300}
if [ "${PS:-}" != "" ]; then
  t0=$(date +%s)
  python -c "import bench; bench.write_y4m('/tmp/job.y4m', $N)"; echo "y4m written in $(( $(date +%s) - t0 )) s"
  free -g | head -2
  for PT in $PS; do
    if [ "${PT%%:*}" = "nodist" ]; then export ODHIP_REFERENCE_LIB=$GRAFT_REPO_ROOT/oracle/_ref/libdaalaref.so; PT=${PT#*:}; else unset ODHIP_REFERENCE_LIB; fi
    P=${PT%%x*}; T=1; if [ "$PT" != "$P" ]; then T=${PT##*x}; fi
    timeout 1500 python bench.py --encode-frames $N --procs-per-gpu $P --threads-per-proc $T --y4m /tmp/job.y4m --encode-check ${CHECK:-6} > $OUT/encode_P${P}_T$T.json 2> $OUT/encode_P${P}_T$T.err; echo "P=$P T=$T rc=$? ref=${ODHIP_REFERENCE_LIB:-distglue}"
    tail -2 $OUT/encode_P${P}_T$T.err
    python - $OUT/encode_P${P}_T$T.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print({k:d.get(k) for k in ("value","frames","encoder_processes_per_gpu","encoder_threads_per_process","host_cores_available")}, d["seconds"], d["prefix_check"], {k:d["rank0"].get(k) for k in ("batched_gpu_pass_ms_per_frame","dering_cache_ms_per_frame","served_pvq_theta_ms_per_frame","od_compute_dist_served_per_frame")})
except Exception as e:
    print("parse failed", repr(e))
PY
  done
fi
