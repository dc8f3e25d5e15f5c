#!/bin/bash
# Round-3 measurement set on one box: the bench lines of the four workloads + rocprofv3 kernel stats of the default line.
#   bash tests/probes/r03_measure.sh   -> gpurun_out/r3m/*
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3m; mkdir -p $O
VX_BENCH_VERBOSE=1 timeout -k 10 300 python bench.py --steps 20 --warmup 3 > $O/bench_line_default.json 2> $O/bench_default.err && echo "default ok" &&
timeout -k 10 300 python bench.py --batch 32 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_line_batch32.json 2> $O/bench_batch32.err && echo "batch32 ok" &&
timeout -k 10 400 python bench.py --batch 64 --precision fp8nar --text-len 94 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_line_cfg4_fp8nar_b64.json 2> $O/bench_cfg4.err && echo "cfg4 ok" &&
timeout -k 10 300 python bench.py --precision fp32 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_line_fp32.json 2> $O/bench_fp32.err && echo "fp32 ok" &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_default -o t -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/prof_default.log 2>&1 && echo "prof ok"
find $O -name "*kernel_trace.csv" -size +20M -delete
ls -la $O $O/prof_default 2>/dev/null | head -30
