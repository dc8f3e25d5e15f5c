"""Scratch: the batched NAR GEMM shapes (34 816 concatenated rows = 32 x 1025 padded to 64-row segments; 134 k rows = configs[4])
through vx_op_gemm (bf16 operands, f32 output), TFLOP/s per shape; random operands (zero-filled ones read high)."""
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

tag = sys.argv[1] if len(sys.argv) > 1 else "?"
for (M, N, K) in [(4096, 4096, 4096), (8192, 8192, 8192), (34816, 3072, 1024), (34816, 1024, 1024), (34816, 4096, 1024), (34816, 1024, 4096), (65536, 4096, 1024)]:
    us, tf = bench(M, N, K)
    print(json.dumps(dict(p8=tag, M=M, N=N, K=K, us=round(us, 1), tflops=round(tf, 1))), flush=True)
