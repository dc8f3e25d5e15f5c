#!/bin/bash
# A/B of one environment switch on the batched bench (BASELINE configs[2]: --batch 32), alternating processes on one box:
#   tests/probes/batch_ab.sh VAR rounds value [value ...]        e.g.  batch_ab.sh VX_BATCH_LNFUSE 2 0 1
VAR=$1; N=$2; shift 2
for i in $(seq $N); do
  for v in "$@"; do
    env $VAR=$v python bench.py --batch ${BATCH:-32} --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.readline())
print('$VAR=$v', 'ar_step_us', r['ar_step_us'], 'nar_ms', r['nar_7stage_ms'], 'prefill_ms', r['prefill_ms'], 'value', r['value'], 'frac', r['roofline']['frac'], flush=True)"
  done
done
