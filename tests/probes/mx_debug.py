"""Scratch: crafted operands through vx_op_gemm_mx to localise an error (exact integers / powers of two survive the MXFP8 quantiser)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
ge.build()
from valle_amd import engine as eng

def run(name, A, W, relu=False):
    C = eng.op_gemm_mx(A.cuda(), W.cuda(), None).cpu()
    ref = A.double() @ W.double().t()
    bad = (C.double() - ref).abs() > 1e-3 * ref.abs().clamp_min(1)
    rows = bad.any(1).nonzero().flatten().tolist(); cols = bad.any(0).nonzero().flatten().tolist()
    print(f"{name}: M,N,K={A.shape[0]},{W.shape[0]},{A.shape[1]} bad {int(bad.sum())}/{bad.numel()} rows {len(rows)} {rows[:12]} cols {len(cols)} {cols[:12]}")
    if bad.any():
        i, j = bad.nonzero()[0].tolist()
        print("   first bad", (i, j), "got", float(C[i, j]), "want", float(ref[i, j]), "| row0 got", C[0, :6].tolist(), "want", ref[0, :6].tolist())

for (M, N, K) in [(256, 256, 128), (256, 256, 256), (512, 512, 128)]:
    one = torch.ones
    m = torch.arange(M).float()[:, None]; n = torch.arange(N).float()[:, None]; k = torch.arange(K).float()[None, :]
    run("ones", one(M, K), one(N, K))
    run("W scale by n", one(M, K), (2.0 ** (n % 5)) * one(N, K))
    run("A scale by m", (2.0 ** (m % 7)) * one(M, K), one(N, K))
    run("A k-block", one(M, 1) * torch.where(k < 64, 1.0, 4.0), one(N, K))
    run("W k-block", one(M, K), one(N, 1) * torch.where((k // 32) % 2 == 0, 1.0, 8.0))
    run("A k pattern", one(M, 1) * (1 + (k % 3)), one(N, K))
    g = torch.Generator().manual_seed(0)
    run("rand ints", torch.randint(-2, 3, (M, K), generator=g).float(), torch.randint(-2, 3, (N, K), generator=g).float())
    run("A onehot k", (k == 37).float() * one(M, 1), one(N, 1) * (1 + (k % 4)))
