#!/bin/bash
# A/B of one environment switch on the default bench (batch-1 AR step), alternating processes on one box:
#   tests/probes/ar_ab.sh VAR rounds value [value ...]        e.g.  ar_ab.sh VX_AR_NT 3 0 1
VAR=$1; N=$2; shift 2
for i in $(seq $N); do
  for v in "$@"; do
    env $VAR=$v python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.readline())
print('$VAR=$v', 'ar_step_us', r['ar_step_us'], 'nar_ms', r['nar_7stage_ms'], 'value', r['value'], flush=True)"
  done
done
