#!/bin/bash
# A/B of the 256^2 GEMM schedules on one box: alternating processes.  VX_GEMM_P8 = 1: 8-phase ping-pong (default), 2: the same with the
# 12/4/8/0 fragment-read schedule, 0: the 32-k ring.
# usage: tests/probes/gemm256_ab.sh OUT.log [rounds] [settings...]
out=${1:-gpurun_out/gemm256_ab.log}; rounds=${2:-2}; shift 2
set -- ${@:-1 0}
: > $out
for r in $(seq $rounds); do
  for p in "$@"; do
    VX_GEMM_P8=$p python3 tests/probes/gemm256_shapes.py $p >> $out 2>&1 || { echo "FAILED p8=$p" >> $out; exit 1; }
  done
done
cat $out
