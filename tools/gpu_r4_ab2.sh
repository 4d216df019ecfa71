#!/bin/bash
# A/B of environment settings on the overlapped step: each argument is one setting ("" = default).
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  for rep in 1 2; do
  env $v timeout 120 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-streaming --no-replay 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('[$v] ms', round(d['ms_per_step'],3))"
  done
done
