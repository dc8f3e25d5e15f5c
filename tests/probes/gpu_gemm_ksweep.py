"""Scratch: fixed cost vs per-K-step cost of the row GEMMs at the batch-1 NAR shape (M = 1025)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
ge.build()
from valle_amd import engine as E

def bench(M, N, K, iters=50):
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
    return e0.elapsed_time(e1) * 1e3 / iters

alg = os.environ.get("VX_GEMM_ALG", "0")
for (M, N) in [(1025, 3072), (1025, 1024), (1025, 4096), (1024, 3072), (128, 3072)]:
    row = {}
    for K in (256, 512, 1024, 2048, 4096):
        row[K] = round(bench(M, N, K), 1)
    print(json.dumps(dict(alg=alg, M=M, N=N, us_by_K=row)), flush=True)
