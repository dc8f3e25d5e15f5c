"""Scratch: MXFP8 GEMM (vx_op_gemm_mx) over shapes / K; run under `rocprofv3 --kernel-trace` and read the mx256* kernel durations
(the op quantises its operands first: those kernels are separate dispatches).  usage: python tests/probes/mx_gemm_sweep.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
ge.build()
from valle_amd import engine as eng

M = int(sys.argv[1]) if len(sys.argv) > 1 else 34816
g = torch.Generator().manual_seed(0)
for (N, K) in [(3072, 256), (3072, 512), (3072, 1024), (3072, 2048), (4096, 1024), (1024, 1024), (1024, 4096)]:
    A = torch.randn(M, K, generator=g).cuda()
    W = (torch.randn(N, K, generator=g) * K ** -0.5).cuda()
    b = torch.randn(N, generator=g).cuda()
    for out_mx in (False, True):
        for _ in range(3):
            eng.op_gemm_mx(A, W, b, relu=True, out_mx=out_mx)
    torch.cuda.synchronize()
    print("done", M, N, K, flush=True)
