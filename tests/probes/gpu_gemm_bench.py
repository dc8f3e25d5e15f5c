"""Scratch: stand-alone timing of the MFMA GEMM variants through vx_op_gemm (bf16), TFLOP/s vs shape."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
ge.build()
from valle_amd import engine as E

def bench(M, N, K, iters=30):
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    W = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
    b = torch.randn(N, device="cuda")
    for _ in range(3):
        E.op_gemm(A, W, b, mfma=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        E.op_gemm(A, W, b, mfma=True)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    return us, 2.0 * M * N * K / us / 1e6

alg = os.environ.get("VX_GEMM_ALG", "0")
for (M, N, K) in [(1025, 3072, 1024), (1025, 4096, 1024), (1025, 1024, 1024), (1025, 1024, 4096), (2050, 3072, 1024), (4100, 3072, 1024), (8200, 3072, 1024), (4096, 4096, 4096), (33000, 3072, 1024), (33000, 1024, 4096), (33000, 4096, 1024), (33000, 1024, 1024)]:
    us, tf = bench(M, N, K)
    print(json.dumps(dict(alg=alg, M=M, N=N, K=K, us=round(us, 1), tflops=round(tf, 1))), flush=True)
