"""GPU: every HIP kernel, called through the C ABI (vx_op_*), against the torch op the oracle
uses for the same step.  Integer/index outputs are compared bit-exactly; floating point within
the tolerance written next to each check."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    import __graft_entry__ as ge

    ge.build()
    from valle_amd import engine

    engine.load_library()
    assert torch.cuda.is_available(), "gpu tests need the MI355X"
    return engine


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


@pytest.mark.parametrize("rows,d", [(1, 1024), (273, 1024), (37, 256), (5, 128)])
@pytest.mark.parametrize("adaptive", [False, True])
def test_layernorm(eng, rows, d, adaptive):
    x, g, b = _rand(rows, d, seed=1, scale=3.0), 1 + 0.1 * _rand(d, seed=2), 0.1 * _rand(d, seed=3)
    w, c = (1 + 0.1 * _rand(d, seed=4), 0.1 * _rand(d, seed=5)) if adaptive else (None, None)
    ref = F.layer_norm(x, (d,), g, b, 1e-5)
    if adaptive:
        ref = w * ref + c
    dev = lambda t: None if t is None else t.cuda()
    out = eng.op_layernorm(dev(x), dev(g), dev(b), dev(w), dev(c), torch.float32).cpu()
    assert (out - ref).abs().max() <= 2e-5  # fp32: differs from torch only in reduction order
    outb = eng.op_layernorm(dev(x), dev(g), dev(b), dev(w), dev(c), torch.bfloat16).cpu().float()
    assert (outb - ref).abs().max() <= 2 ** -7 * ref.abs().max()  # one bf16 rounding


@pytest.mark.parametrize("N,K", [(3072, 1024), (1024, 4096), (1025, 1024), (4096, 1024), (768, 256), (1024, 256),
                                 (384, 128), (128, 512), (1025, 128)])
@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_gemv(eng, N, K, prec):
    W, b, x = _rand(N, K, seed=1, scale=K ** -0.5), _rand(N, seed=2), _rand(K, seed=3)
    Wd = W.cuda() if prec == "fp32" else W.cuda().to(torch.bfloat16)
    Wr = W if prec == "fp32" else W.to(torch.bfloat16).float()  # same stored values, fp32 arithmetic
    ref = F.linear(x.double(), Wr.double(), b.double()).float()
    y = eng.op_gemv(Wd, b.cuda(), x.cuda()).cpu()
    assert (y - ref).abs().max() <= 2e-5 * max(1.0, float(ref.abs().max()))
    yr = eng.op_gemv(Wd, b.cuda(), x.cuda(), relu=True).cpu()
    assert (yr - ref.clamp_min(0)).abs().max() <= 2e-5 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("M,N,K", [(272, 3072, 1024), (1025, 1024, 4096), (100, 384, 128), (753, 1024, 1024), (65, 256, 256)])
def test_gemm_fp32(eng, M, N, K):
    A, W, b = _rand(M, K, seed=1), _rand(N, K, seed=2, scale=K ** -0.5), _rand(N, seed=3)
    ref = F.linear(A.double(), W.double(), b.double()).float()
    out = eng.op_gemm(A.cuda(), W.cuda(), b.cuda()).cpu()
    assert (out - ref).abs().max() <= 1e-4


@pytest.mark.parametrize("M,N,K", [(272, 3072, 1024), (1025, 1024, 4096), (100, 384, 128), (753, 1024, 1024), (128, 128, 64),
                                   (1, 256, 256), (4133, 512, 256), (4096, 768, 1024), (4700, 256, 64)])
@pytest.mark.parametrize("mfma", [False, True])
def test_gemm_bf16(eng, M, N, K, mfma):
    A, W, b = _rand(M, K, seed=1).to(torch.bfloat16), _rand(N, K, seed=2, scale=K ** -0.5).to(torch.bfloat16), _rand(N, seed=3)
    ref = F.linear(A.double(), W.double(), b.double()).float()  # exact products of the bf16 values
    out = eng.op_gemm(A.cuda(), W.cuda(), b.cuda(), mfma=mfma).cpu()
    assert (out - ref).abs().max() <= 2e-4 * max(1.0, float(ref.abs().max()))  # fp32 accumulation order only
    outr = eng.op_gemm(A.cuda(), W.cuda(), b.cuda(), relu=True, mfma=mfma).cpu()
    assert (outr - ref.clamp_min(0)).abs().max() <= 2e-4 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("M,N,K", [(16500, 1024, 1024), (16400, 1280, 128), (20000, 768, 384), (66000, 256, 256), (34700, 1024, 4096),
                                   (16500, 1024, 192), (20000, 512, 320)])  # the last two: K % 128 = 64 -> the 32-k ring kernel (mfma256_kernel)
def test_gemm_bf16_persistent_256_tiles(eng, M, N, K):
    """More 256^2 tiles than CUs: every persistent workgroup walks two or more tiles (operand stream running on across the tile
    boundary, counted waits behind an epilogue's stores, ragged last row tile) - the 8-phase kernel, and the 32-k ring that serves
    the K it cannot take.  Reference: fp64 products of the same bf16 values."""
    g = torch.Generator().manual_seed(5)
    A = torch.randn(M, K, generator=g).to(torch.bfloat16).cuda()
    W = (torch.randn(N, K, generator=g) * K ** -0.5).to(torch.bfloat16).cuda()
    b = torch.randn(N, generator=g).cuda()
    ref = F.linear(A.double(), W.double(), b.double()).float()
    for relu in (False, True):
        out = eng.op_gemm(A, W, b, relu=relu, mfma=True)
        r = ref.clamp_min(0) if relu else ref
        err = float((out - r).abs().max())
        assert err <= 2e-4 * max(1.0, float(r.abs().max())), (relu, err)
    assert torch.equal(eng.op_gemm(A, W, b, mfma=True), eng.op_gemm(A, W, b, mfma=True))  # bitwise reproducible


def test_gemm_bf16_tail_split():
    """VX_GEMM_TAIL=1 (off by default: position-dependent rounding): the last, mostly idle round of 256^2 tiles is split 8 / 4 ways
    along K and summed by a second launch.  The switch is read once per process: one child process, both shapes."""
    import subprocess, sys, os
    from conftest import ROOT

    code = f"""
import sys, torch
sys.path.insert(0, {ROOT!r})
import __graft_entry__ as ge
ge.build()
from valle_amd import engine as E
for M, N, K in ((34700, 1024, 4096), (66000, 1024, 2048)):
    g = torch.Generator().manual_seed(5)
    A = torch.randn(M, K, generator=g).to(torch.bfloat16).cuda()
    W = (torch.randn(N, K, generator=g) * K ** -0.5).to(torch.bfloat16).cuda()
    b = torch.randn(N, generator=g).cuda()
    ref = torch.nn.functional.linear(A.double(), W.double(), b.double()).float()
    for relu in (False, True):
        out = E.op_gemm(A, W, b, relu=relu, mfma=True)
        r = ref.clamp_min(0) if relu else ref
        err = float((out - r).abs().max())
        assert err <= 2e-4 * max(1.0, float(r.abs().max())), (M, relu, err)
    assert torch.equal(E.op_gemm(A, W, b, mfma=True), E.op_gemm(A, W, b, mfma=True))
    del A, W, b, ref
print("ok")
"""
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, VX_GEMM_TAIL="1"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("M,N,K", [(1025, 3072, 1024), (4100, 768, 256), (8704, 3072, 1024), (34816, 1024, 1024), (20000, 1024, 4096)])
def test_gemm_rows_forms(eng, M, N, K):
    """The epilogues the row path itself launches, at M below and above the 256^2 kernel's threshold: bf16 output (+ ReLU) with the
    V^T copy of the last third of the columns (QKV), and the fp32 residual update (out-projection / FFN2) - the latter applied
    exactly once (a tile written twice would add its product twice)."""
    g = torch.Generator().manual_seed(11)
    A = torch.randn(M, K, generator=g).to(torch.bfloat16).cuda()
    W = (torch.randn(N, K, generator=g) * K ** -0.5).to(torch.bfloat16).cuda()
    b = torch.randn(N, generator=g).cuda()
    ref = F.linear(A.double(), W.double(), b.double())
    scale = max(1.0, float(ref.abs().max()))
    vc = N // 3 // 64 * 64 + (N % 64)  # the copy starts at a multiple of 64 columns (the engine's is 2 d)
    C, vt = eng.op_gemm_rows(A, W, b, vt_cols=vc)
    assert (C.double() - ref).abs().max() <= 2 ** -8 * scale  # one bf16 rounding of an fp32 sum
    assert torch.equal(vt[:, :M], C[:, N - vc:].t())  # the transposed copy holds the same bf16 values
    Cr, _ = eng.op_gemm_rows(A, W, b, relu=True)
    assert (Cr.double() - ref.clamp_min(0)).abs().max() <= 2 ** -8 * scale
    X0 = torch.randn(M, N, generator=g).cuda()
    X = eng.op_gemm_rows(A, W, b, resid=X0.clone())
    assert (X.double() - (X0.double() + ref)).abs().max() <= 3e-4 * scale


def test_mfma_gemm_layout_asymmetric(eng):
    """A = I against an asymmetric W catches a transposed C write (cdna guide §3)."""
    K = 128
    A = torch.eye(K).to(torch.bfloat16)
    W = (torch.arange(256 * K).reshape(256, K) % 251).float().to(torch.bfloat16)
    out = eng.op_gemm(A.cuda(), W.cuda(), None, mfma=True).cpu()
    assert torch.equal(out, W.float().t().contiguous())


def _attn_ref(qkv, H, text_len):
    """fp64 SDPA with the reference's prefix mask (valle.py:1019-1033), on the qkv tensor's device, 16 heads at a time."""
    N, d3 = qkv.shape
    d = d3 // 3
    q, k, v = qkv.double().chunk(3, -1)
    sp = lambda t: t.reshape(N, H, d // H).transpose(0, 1)
    m = None
    if text_len >= 0:
        m = torch.zeros(N, N, dtype=torch.bool, device=qkv.device)
        m[:text_len, text_len:] = True
        A = N - text_len
        m[text_len:, text_len:] = torch.triu(torch.ones(A, A, dtype=torch.bool, device=qkv.device), 1)
    outs = []
    for h0 in range(0, H, 16):
        s = sp(q)[h0 : h0 + 16] @ sp(k)[h0 : h0 + 16].transpose(1, 2) / (d // H) ** 0.5
        if m is not None:
            s = s.masked_fill(m, float("-inf"))
        outs.append(torch.softmax(s, -1) @ sp(v)[h0 : h0 + 16])
    return torch.cat(outs).transpose(0, 1).reshape(N, d).float()


# (1800, 16, *): configs[4]'s segment length on the two-key-group kernel; (1100, 128, *): 18 x 128 = 2304 (query block, head) pairs
# put the stand-alone op on the four-query-wave kernel of the batched stages (plain + masked tile loops, XCD-aware block order)
@pytest.mark.parametrize("N,H,text_len", [(272, 16, 47), (1025, 16, -1), (70, 4, 10), (63, 2, -1), (130, 4, 0), (9, 2, 9),
                                          (1800, 16, -1), (1800, 16, 94), (1100, 128, -1), (1100, 128, 94), (1100, 128, 1100)])
def test_attention_rows(eng, N, H, text_len):
    big = N * H > 20000  # reference on the GPU (plain torch fp64), the fp32 scalar kernel only at the small shapes
    qkv = _rand(N, 3 * H * 64, seed=5)
    if not big:
        ref = _attn_ref(qkv, H, text_len)
        out = eng.op_attention(qkv.cuda(), H, text_len).cpu()
        assert (out - ref).abs().max() <= 2e-5
    qb = qkv.to(torch.bfloat16)
    refb = _attn_ref(qb.float().cuda() if big else qb.float(), H, text_len).cpu()
    for mfma in (False, True) if not big else (True,):
        outb = eng.op_attention(qb.cuda(), H, text_len, mfma=mfma).cpu().float()
        assert (outb - refb).abs().max() <= 2e-2  # bf16 output rounding (|out| <~ 3) (+ bf16 P in the MFMA path)


@pytest.mark.parametrize("top_k,temp", [(-100, 1.0), (1, 1.0), (10, 1.0), (50, 0.7), (64, 1.0), (65, 0.9), (100, 1.0),
                                        (1024, 1.0), (1025, 1.3), (2000, 1.0), (3, 2.0)])
def test_sampling_matches_oracle_bit_exact(eng, top_k, temp):
    """Index selection (argmax, top-k set, sampled token) must equal the oracle's exactly."""
    from oracle import valle_oracle as vo

    for seed in range(12):
        logits = _rand(1, 1025, seed=seed, scale=3.0)
        if seed % 3 == 0:  # exact ties around the k-th value and at the maximum
            logits[0, 5] = logits[0, 900] = logits[0].topk(10)[0][-1]
            logits[0, 77] = logits[0].max()
        g = torch.Generator().manual_seed(100 + seed)
        q = torch.empty(1, 1025).exponential_(1, generator=g)
        want = int(vo.topk_sampling(logits.clone(), top_k, temp, q))
        got, am = eng.op_sample(logits[0].cuda(), top_k, temp, q[0].cuda())
        assert am == int(torch.argmax(logits, -1))
        assert got == want, (seed, top_k, temp)


@pytest.mark.parametrize("top_k", [-1, 7, 80])
def test_sampling_single_wave_variant_large_vocab(eng, top_k):
    """V = 1500 > 1088 routes vx_op_sample to the single-wave kernel (32 keys per lane): same exactness."""
    from oracle import valle_oracle as vo

    for seed in range(6):
        logits = _rand(1, 1500, seed=50 + seed, scale=2.5)
        g = torch.Generator().manual_seed(500 + seed)
        q = torch.empty(1, 1500).exponential_(1, generator=g)
        want = int(vo.topk_sampling(logits.clone(), top_k, 0.9, q))
        got, am = eng.op_sample(logits[0].cuda(), top_k, 0.9, q[0].cuda())
        assert am == int(torch.argmax(logits, -1)) and got == want


# ---- MXFP8 kernels of VX_PREC_FP8_NAR (mx_kernels.hpp) against the host emulation of the same quantiser (tests/mx_ref.py) ----
@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (300, 512, 1024), (4100, 3072, 1024), (513, 1024, 4096),
                                   (20000, 1024, 512), (16700, 1280, 256),  # more tiles than CUs (two per persistent workgroup)
                                   (17000, 1024, 384)])                      # K % 256 = 128: the 32-k ring kernel (mx256_kernel), two tiles per workgroup
def test_mx_gemm_matches_host_emulation(eng, M, N, K):
    """Quantiser: bit-exact bytes and scales.  GEMM: both operands dequantise to exact fp32 values, so the matrix core's result
    differs from an fp64 evaluation of the same dequantised operands only by its internal accumulation: tolerance stated as a
    fraction of the magnitude budget sum |a||w| of each output."""
    from mx_ref import mx_dequant, mx_gemm_ref, mx_quant, mx_spos, scales_by_row

    A, W, b = _rand(M, K, seed=1, scale=1.7), _rand(N, K, seed=2, scale=K ** -0.5), _rand(N, seed=3)
    A[3, 5] = 300.0  # an outlier block and an all-zero block
    A[min(7, M - 1), 32:64] = 0.0
    ld = (M + 255) // 256 * 256
    C, qa, sa = eng.op_gemm_mx(A.cuda(), W.cuda(), b.cuda(), return_quant=True)
    q_ref, s_ref = mx_quant(A)
    assert torch.equal(qa.cpu(), q_ref)
    assert torch.equal(scales_by_row(sa.cpu(), M), s_ref)
    unused = torch.ones(ld, dtype=torch.bool)
    unused[mx_spos(M)] = False
    assert int(sa.cpu()[:, unused].sum()) == 0  # pad entries stay zero (finite scales for the tile loader's clamped rows)
    ref = mx_gemm_ref(A, W) + b
    budget = mx_dequant(*mx_quant(A)).abs().double() @ mx_dequant(*mx_quant(W)).abs().double().t()
    ratio = float(((C.cpu() - ref).abs().double() / (budget + 1e-3)).max())
    print(M, N, K, "max |C - ref| / sum|a||w| = %.2e" % ratio)
    # measured 3e-5 .. 2e-4 on MI355X: the block-scaled matrix core does not accumulate like a chain of correctly rounded fp32
    # fmas (the bf16 / f32 MFMAs stay within 1e-6 of the same budget); three orders of magnitude below the e4m3 rounding itself
    assert ratio <= 5e-4
    Cr = eng.op_gemm_mx(A.cuda(), W.cuda(), b.cuda(), relu=True).cpu()
    assert float(((Cr - ref.clamp_min(0)).abs().double() / (budget + 1e-3)).max()) <= 5e-4
    # the quantisation error itself, for the record: relative to the fp32 product
    exact = F.linear(A.double(), W.double(), b.double()).float()
    print(M, N, K, "mxfp8 vs fp32 GEMM: rel. error %.4f of the output rms" % float((ref - exact).pow(2).mean().sqrt() / exact.pow(2).mean().sqrt()))


def test_mx_gemm_quantised_output(eng):
    """FFN1's epilogue: ReLU(C + bias) leaves the GEMM as e4m3 bytes with one scale per 32 columns (the A operand of FFN2).  Checked
    against the same kernel's fp32 output x = ReLU(C + bias): every dequantised value within e4m3's rounding of x under its block's
    scale (half an ulp = x / 16 for normals, 2^-10 scale units for subnormals, up to x / 8 where the block maximum's mantissa
    exceeds 1.75 and the top clips to 448), and the scales equal to the host quantiser's on x (a block whose maximum sits within
    rounding of a power of two may differ by one)."""
    from mx_ref import mx_dequant, mx_quant, scales_by_row

    M, N, K = 700, 1024, 256
    A, W, b = _rand(M, K, seed=4), _rand(N, K, seed=5, scale=K ** -0.5), _rand(N, seed=6)
    (c8, sc) = eng.op_gemm_mx(A.cuda(), W.cuda(), b.cuda(), out_mx=True)
    x = eng.op_gemm_mx(A.cuda(), W.cuda(), b.cuda(), relu=True).cpu()
    sc_rows = scales_by_row(sc.cpu(), M)
    got = mx_dequant(c8.cpu(), sc_rows)
    amax = x.reshape(M, N // 32, 32).abs().amax(-1).repeat_interleave(32, dim=1)
    assert bool(((got - x).abs() <= torch.maximum(0.1251 * x.abs(), amax * 2.0 ** -17) + 1e-7).all())
    q_ref, s_ref = mx_quant(x)
    assert float((sc_rows == s_ref).float().mean()) >= 0.999 and int((sc_rows.int() - s_ref.int()).abs().max()) <= 1
    same_scale = (sc_rows == s_ref).repeat_interleave(32, dim=1)
    assert float((c8.cpu() == q_ref)[same_scale].float().mean()) >= 0.999  # same fp32 sums -> same bytes (different ones: see above)


@pytest.mark.parametrize("rows,d", [(300, 1024), (37, 256)])
@pytest.mark.parametrize("adaptive", [False, True])
def test_layernorm_mx(eng, rows, d, adaptive):
    from mx_ref import mx_quant, scales_by_row

    x, g, b = _rand(rows, d, seed=1, scale=3.0), 1 + 0.1 * _rand(d, seed=2), 0.1 * _rand(d, seed=3)
    w, c = (1 + 0.1 * _rand(d, seed=4), 0.1 * _rand(d, seed=5)) if adaptive else (None, None)
    dev = lambda t: None if t is None else t.cuda()
    f32 = eng.op_layernorm(dev(x), dev(g), dev(b), dev(w), dev(c), torch.float32).cpu()  # same arithmetic, unquantised
    q, sc = eng.op_layernorm_mx(dev(x), dev(g), dev(b), dev(w), dev(c))
    q_ref, s_ref = mx_quant(f32)
    assert torch.equal(scales_by_row(sc.cpu(), rows), s_ref)
    assert torch.equal(q.cpu(), q_ref)
