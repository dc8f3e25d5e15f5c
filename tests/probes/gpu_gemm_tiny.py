"""Scratch: fixed cost of one 128x128 GEMM workgroup (launch + prologue + epilogue) from kernel-trace durations."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
ge.build()
from valle_amd import engine as E

for (M, N, K) in [(128, 128, 64), (128, 128, 128), (128, 128, 256), (128, 128, 1024), (128, 3072, 64), (1025, 3072, 64), (1025, 3072, 128)]:
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    W = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
    b = torch.randn(N, device="cuda")
    for _ in range(20):
        E.op_gemm(A, W, b, mfma=True)
    torch.cuda.synchronize()
    print(json.dumps(dict(M=M, N=N, K=K)), flush=True)
