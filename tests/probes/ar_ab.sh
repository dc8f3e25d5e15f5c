#!/bin/bash
# A/B of one environment switch on the default bench (batch-1 AR step), alternating processes on one box:
#   tests/probes/ar_ab.sh VX_AR_NT 0 1 [rounds]
VAR=$1; A=$2; B=$3; N=${4:-3}
for i in $(seq $N); do
  for v in $A $B; do
    env $VAR=$v python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.readline())
print('$VAR=$v', 'ar_step_us', r['ar_step_us'], 'nar_ms', r['nar_7stage_ms'], 'value', r['value'], flush=True)"
  done
done
