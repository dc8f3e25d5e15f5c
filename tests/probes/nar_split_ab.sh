#!/bin/bash
# A/B of the split-K factors of the batch-1 NAR out-projection / FFN2 (VX_SPLIT_D / VX_SPLIT_FF, engine.hip run_stack) on one box.
run() { env "$@" python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.readline())
print('$*', 'nar_ms', r['nar_7stage_ms'], 'prefill_ms', r['prefill_ms'], flush=True)"; }
for i in 1 2; do
  run VX_SPLIT_D=4 VX_SPLIT_FF=4
  run VX_SPLIT_D=2 VX_SPLIT_FF=4
  run VX_SPLIT_D=1 VX_SPLIT_FF=4
  run VX_SPLIT_D=1 VX_SPLIT_FF=4 VX_GEMM_ALG=1
  run VX_SPLIT_D=2 VX_SPLIT_FF=2
  run VX_SPLIT_D=4 VX_SPLIT_FF=2
done
