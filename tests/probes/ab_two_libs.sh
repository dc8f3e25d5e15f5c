#!/bin/bash
# A/B of two library builds on one box through bench.py: vall-e_amd/csrc/libvallex.so (current) against libvallex_head.so (built from
# `git archive HEAD` by hand), alternating processes; the package always loads libvallex.so, so the files are swapped in place.
# usage: tests/probes/ab_two_libs.sh "<bench.py arguments>" [rounds]
args=$1; rounds=${2:-2}
cd vall-e_amd/csrc && cp libvallex.so lib_cur.so && cd ../..
for r in $(seq $rounds); do
  for l in lib_cur libvallex_head; do
    cp vall-e_amd/csrc/$l.so vall-e_amd/csrc/libvallex.so; touch vall-e_amd/csrc/libvallex.so
    python3 bench.py $args --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$l', d['value'], d['ms_per_step'], d['ar_step_us'], d['nar_7stage_ms'], d['prefill_ms'])"
  done
done
cp vall-e_amd/csrc/lib_cur.so vall-e_amd/csrc/libvallex.so
