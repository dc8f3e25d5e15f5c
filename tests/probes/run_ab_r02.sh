mkdir -p gpurun_out/r2; rm -f gpurun_out/r2/ab_mx.log gpurun_out/r2/ab_wgs.log
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -s -k "mx_gemm or layernorm_mx" > gpurun_out/r2/t6.log 2>&1; grep -E "passed|failed" gpurun_out/r2/t6.log
for a in 0 1 0 1; do VX_MX_ALG=$a timeout -k 10 200 python bench.py --batch 64 --precision fp8nar --text-len 94 --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.readline()); print('VX_MX_ALG=$a', 'nar_ms', r['nar_7stage_ms'], 'gemm TF', r['roofline']['achieved'], 'gemm ms', r['roofline']['ms_per_launch'], 'value', r['value'], flush=True)" >> gpurun_out/r2/ab_mx.log 2>&1; done; cat gpurun_out/r2/ab_mx.log
for i in 1 2 3; do for w in none 1024x1024:64 1024x1024:128 1024x4096:128 1024x1024:64,1024x4096:128; do VX_AR_WGS=$w python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.readline()); print('VX_AR_WGS=$w', 'ar_step_us', r['ar_step_us'], flush=True)" >> gpurun_out/r2/ab_wgs.log; done; done; cat gpurun_out/r2/ab_wgs.log
