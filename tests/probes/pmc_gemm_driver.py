"""Torch-free driver: one bf16 GEMM through vx_op_gemm for rocprofv3 --pmc (traffic / MFMA counters).
usage: python3 tests/probes/pmc_gemm_driver.py M N K [iters] [form]   (VX_GEMM_ALG selects the kernel)
form (optional): "bf16" = bf16 output + bias, "qkv" = the same + V^T copy of the last third, "relu" = bf16 output + ReLU, "resid" = fp32
residual update (vx_op_gemm_rows: the forms the engine's row path launches); default: fp32 output + bias (vx_op_gemm)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hip = C.CDLL("libamdhip64.so")
lib = C.CDLL(os.path.join(ROOT, "vall-e_amd", "csrc", "libvallex.so"))
lib.vx_last_error.restype = C.c_char_p
M, N, K = (int(v) for v in sys.argv[1:4])
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 3
form = sys.argv[5] if len(sys.argv) > 5 else "f32"


def dmalloc(nbytes, fill=0x3c):
    p = C.c_void_p()
    assert hip.hipMalloc(C.byref(p), C.c_size_t(nbytes)) == 0
    assert hip.hipMemset(p, fill, C.c_size_t(nbytes)) == 0  # 0x3c3c = bf16 0.0115: finite, non-zero
    return p


def drandom_bf16(n, seed):
    """bf16 values of random sign and mantissa with magnitudes in [2^-4, 2): constant operands let the clock run higher than real
    data does (cdna guide, methodology rule 25), and MFMA-busy fractions are quoted against elapsed cycles."""
    import numpy as np

    rng = np.random.default_rng(seed)
    bits = (rng.integers(0, 2, n, dtype=np.uint16) << 15) | (rng.integers(0x3D80, 0x4000, n, dtype=np.uint16))
    p = C.c_void_p()
    assert hip.hipMalloc(C.byref(p), C.c_size_t(n * 2)) == 0
    assert hip.hipMemcpy(p, bits.ctypes.data_as(C.c_void_p), C.c_size_t(n * 2), 1) == 0
    return p


if os.environ.get("VX_PMC_CONST"):
    A, W = dmalloc(M * K * 2), dmalloc(N * K * 2)
else:
    A, W = drandom_bf16(M * K, 1), drandom_bf16(N * K, 2)
bias, Cm = dmalloc(N * 4, 0), dmalloc(M * N * 4, 0)
lib.vx_op_gemm.argtypes = [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int32] * 4 + [C.c_void_p]
lib.vx_op_gemm_rows.argtypes = [C.c_int32] + [C.c_void_p] * 4 + [C.c_int32] * 4 + [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
ld = (M + 255) // 256 * 256
vt = dmalloc((N // 3) * ld * 2, 0) if form == "qkv" else None
if form == "mx":  # MXFP8 GEMM (vx_op_gemm_mx quantises fp32 operands on the device, then mx256p_kernel): fp32 copies of the operands
    import numpy as np

    def dev_f32(n, seed, scale):
        rng = np.random.default_rng(seed)
        h = (rng.standard_normal(n, dtype=np.float32) * scale).astype(np.float32)
        p = C.c_void_p()
        assert hip.hipMalloc(C.byref(p), C.c_size_t(n * 4)) == 0
        assert hip.hipMemcpy(p, h.ctypes.data_as(C.c_void_p), C.c_size_t(n * 4), 1) == 0
        return p

    Af, Wf = dev_f32(M * K, 3, 1.0), dev_f32(N * K, 4, K ** -0.5)
    lib.vx_op_gemm_mx.argtypes = [C.c_void_p] * 5 + [C.c_int32] * 5 + [C.c_void_p] * 3
for _ in range(iters):
    if form == "mx":
        rc = lib.vx_op_gemm_mx(Af, Wf, bias, Cm, None, M, N, K, 0, 0, None, None, None)
    elif form == "f32":
        rc = lib.vx_op_gemm(1, 1, A, W, bias, Cm, M, N, K, 0, None)
    elif form == "resid":
        rc = lib.vx_op_gemm_rows(1, A, W, bias, Cm, M, N, K, 0, None, 0, 0, None)
    else:
        rc = lib.vx_op_gemm_rows(0, A, W, bias, Cm, M, N, K, 1 if form == "relu" else 0, vt, N - N // 3 if vt else 0, ld if vt else 0, None)
    assert rc == 0, lib.vx_last_error()
assert hip.hipDeviceSynchronize() == 0
print("ok", M, N, K, form)
