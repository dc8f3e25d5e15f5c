#!/bin/bash
# A/B of the 256^2 GEMM schedules on one box: alternating processes, VX_GEMM_P8 = 1 (8-phase ping-pong) / 0 (32-k ring).
# usage: tests/probes/gemm256_ab.sh OUT.log [rounds]
out=${1:-gpurun_out/gemm256_ab.log}; rounds=${2:-2}
: > $out
for r in $(seq $rounds); do
  for p in 1 0; do
    VX_GEMM_P8=$p python3 tests/probes/gemm256_shapes.py $p >> $out 2>&1 || { echo "FAILED p8=$p" >> $out; exit 1; }
  done
done
cat $out
