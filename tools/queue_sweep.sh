#!/bin/bash
# Development: the 16-frame step against the number of hardware queues and the per-chain stream forks (one box).
run() { python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-replay 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', round(d['ms_per_step'],3))"; }
for rep in 1 2; do
run default
ODHIP_PIPE_FORK=1 run fork
GPU_MAX_HW_QUEUES=2 run q2
GPU_MAX_HW_QUEUES=3 run q3
GPU_MAX_HW_QUEUES=6 run q6
GPU_MAX_HW_QUEUES=8 run q8
ODHIP_PIPE_FORK=1 GPU_MAX_HW_QUEUES=8 run fork_q8
done
