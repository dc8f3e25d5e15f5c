"""Host emulation of the engine's MXFP8 quantiser (vall-e_amd/csrc/mx_kernels.hpp: mx_block_scale / mx_pack4), used by the GPU
parity tests of the VX_PREC_FP8_NAR kernels.  OCP MX block scaling: 32 consecutive k of a row share one E8M0 scale
2^(floor(log2 amax) - 8); the scaled values are clamped to +-448 and rounded to nearest even into OCP e4m3 (torch.float8_e4m3fn)."""
import torch


def mx_quant(x: torch.Tensor):
    """x (..., K) fp32, K % 32 == 0 -> (q uint8 (..., K) e4m3 bit patterns, scale uint8 (..., K/32) E8M0)."""
    K = x.shape[-1]
    xb = x.float().reshape(*x.shape[:-1], K // 32, 32)
    amax = xb.abs().amax(-1)
    E = (amax.contiguous().view(torch.int32) >> 23) & 0xFF
    byte = torch.clamp(E, min=8) - 8
    inv = ((254 - byte) << 23).to(torch.int32).view(torch.float32)
    q = (xb * inv[..., None]).clamp(-448.0, 448.0).to(torch.float8_e4m3fn)
    return q.reshape(x.shape).view(torch.uint8), byte.to(torch.uint8)


def mx_dequant(q: torch.Tensor, scale: torch.Tensor) -> torch.Tensor:
    """inverse of mx_quant: exact fp32 values the matrix core multiplies."""
    K = q.shape[-1]
    v = q.view(torch.float8_e4m3fn).float().reshape(*q.shape[:-1], K // 32, 32)
    s = torch.pow(2.0, scale.float() - 127.0)
    return (v * s[..., None]).reshape(q.shape)


def mx_gemm_ref(A: torch.Tensor, W: torch.Tensor) -> torch.Tensor:
    """C = dq(A) . dq(W)^T in fp64 (the products of the dequantised operands are exact in fp32; only the accumulation order differs)."""
    a = mx_dequant(*mx_quant(A)).double()
    w = mx_dequant(*mx_quant(W)).double()
    return (a @ w.t()).float()


def mx_spos(n_rows: int) -> torch.Tensor:
    """position of row r's scale inside a k-block's run of the engine's scale arrays (mx_kernels.hpp mx_spos)."""
    r = torch.arange(n_rows)
    return (r & ~127) | ((r & 31) << 2) | ((r >> 5) & 3)


def scales_by_row(sc: torch.Tensor, n_rows: int) -> torch.Tensor:
    """engine scale array (K/32, ld) -> (n_rows, K/32) in row order."""
    return sc[:, mx_spos(n_rows)].t().contiguous()
