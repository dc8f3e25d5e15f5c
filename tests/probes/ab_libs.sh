#!/bin/bash
# A/B of several library builds on one box through bench.py, alternating processes.  The package always loads libvallex.so, so the
# variant files (vall-e_amd/csrc/<name>.so, built by tests/probes/build_variant.sh) are copied over it in turn; the original is restored.
# usage: tests/probes/ab_libs.sh "<bench.py arguments>" rounds name [name ...]
args=$1; rounds=$2; shift 2
C=vall-e_amd/csrc
cp $C/libvallex.so $C/lib_orig.so
for r in $(seq $rounds); do
  for l in "$@"; do
    cp $C/$l.so $C/libvallex.so; touch $C/libvallex.so
    python3 bench.py $args --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$l', 'value', d['value'], 'ms_per_step', d['ms_per_step'], 'ar_step_us', d['ar_step_us'], 'nar_ms', d['nar_7stage_ms'], 'prefill_ms', d['prefill_ms'], flush=True)"
  done
done
cp $C/lib_orig.so $C/libvallex.so; touch $C/libvallex.so
