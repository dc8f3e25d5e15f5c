"""Scratch: tail split of the 256^2 GEMM on / off (VX_GEMM_TAIL, read once per process: run twice) on the shapes it applies to."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
ge.build()
from valle_amd import engine as E

def bench(M, N, K, iters=20):
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

for (M, N, K) in [(34816, 1024, 4096), (134144, 1024, 4096), (34816, 1024, 1024), (34816, 3072, 1024), (69632, 2048, 2048)]:
    us, tf = bench(M, N, K)
    print(json.dumps(dict(tail=os.environ.get("VX_GEMM_TAIL", "1"), M=M, N=N, K=K, us=round(us, 1), tflops=round(tf, 1))), flush=True)
