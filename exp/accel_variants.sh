#!/bin/bash
# exp/accel_variants.sh "bench args": the grid's resolution and layout forced, one bench run each
for v in "ACCEL_LOG2=None ACCEL_BRICKS=None" "ACCEL_LOG2=7 ACCEL_BRICKS=True" "ACCEL_LOG2=8 ACCEL_BRICKS=False" "ACCEL_LOG2=8 ACCEL_BRICKS=True" "ACCEL_LOG2=6 ACCEL_BRICKS=False"; do
  python exp/bench_with.py $v -- --no-cpu-baseline --no-plain --no-other-configs $1 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$v', d['value'], d['ms_per_step'], d['kernel_ms'], flush=True)"
done
