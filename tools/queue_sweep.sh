#!/bin/bash
# (round 5) the A/B switches below exist in the experiments build of the library only
export ODHIP_LIB=${ODHIP_LIB:-$(cd "$(dirname "$0")/.." && pwd)/daala_amd/lib/libdaalahip_exp.so}
# Development: the 16-frame step against the number of hardware queues and the per-chain stream forks (one box).
run() { python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-replay 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', round(d['ms_per_step'],3))"; }
for rep in 1 2; do
run default
ODHIP_PIPE_FORK=1 run fork_luma
ODHIP_PIPE_FORK=2 run fork_chroma
ODHIP_PIPE_FORK=3 run fork_both
GPU_MAX_HW_QUEUES=6 ODHIP_PIPE_FORK=1 run fork_luma_q6
GPU_MAX_HW_QUEUES=8 ODHIP_PIPE_FORK=1 run fork_luma_q8
done
