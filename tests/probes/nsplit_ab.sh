#!/bin/bash
# A/B of the decode attention's key splits per head (ATT_NSPLIT, a compile-time constant): variant libraries built with
# -DVX_ATT_NSPLIT=4 / 16 are swapped in for libvallex.so between bench processes on one box.
L=vall-e_amd/csrc/libvallex.so
cp $L /tmp/libvallex_ns8.so
run() { cp $1 $L; python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.readline())
print('$2', 'ar_step_us', r['ar_step_us'], 'value', r['value'], flush=True)"; }
for i in 1 2 3; do
  run /tmp/libvallex_ns8.so nsplit=8
  run tests/probes/libvallex_ns4.so nsplit=4
  run tests/probes/libvallex_ns16.so nsplit=16
done
cp /tmp/libvallex_ns8.so $L
