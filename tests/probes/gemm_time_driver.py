"""Torch-free timing of one bf16 GEMM through vx_op_gemm of a given library build (A/B of two builds on one box).
usage: python3 tests/probes/gemm_time_driver.py LIB.so M N K [iters] [form]     form: f32 (vx_op_gemm, default) | bf16 | relu | qkv | resid
(vx_op_gemm_rows: the row path's own forms)"""
import ctypes as C
import os
import sys

import numpy as np

hip = C.CDLL("libamdhip64.so")
lib = C.CDLL(os.path.abspath(sys.argv[1]))
lib.vx_last_error.restype = C.c_char_p
M, N, K = (int(v) for v in sys.argv[2:5])
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 20
form = sys.argv[6] if len(sys.argv) > 6 else "f32"


def dev(arr):
    p = C.c_void_p()
    assert hip.hipMalloc(C.byref(p), C.c_size_t(arr.nbytes)) == 0
    assert hip.hipMemcpy(p, arr.ctypes.data_as(C.c_void_p), C.c_size_t(arr.nbytes), 1) == 0
    return p


rng = np.random.default_rng(1)
rnd = lambda n: ((rng.integers(0, 2, n, dtype=np.uint16) << 15) | rng.integers(0x3D80, 0x4000, n, dtype=np.uint16))
A, W = dev(rnd(M * K)), dev(rnd(N * K))
bias, Cm = dev(np.zeros(N, np.float32)), dev(np.zeros(M * N, np.float32))
lib.vx_op_gemm.argtypes = [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int32] * 4 + [C.c_void_p]
if form != "f32":
    lib.vx_op_gemm_rows.argtypes = [C.c_int32] + [C.c_void_p] * 4 + [C.c_int32] * 4 + [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
ld = (M + 255) // 256 * 256
vt = dev(np.zeros((N // 3) * ld, np.uint16)) if form == "qkv" else None


def launch():
    if form == "f32":
        return lib.vx_op_gemm(1, 1, A, W, bias, Cm, M, N, K, 0, None)
    if form == "resid":
        return lib.vx_op_gemm_rows(1, A, W, bias, Cm, M, N, K, 0, None, 0, 0, None)
    return lib.vx_op_gemm_rows(0, A, W, bias, Cm, M, N, K, 1 if form == "relu" else 0, vt, N - N // 3 if vt else 0, ld if vt else 0, None)


e0, e1 = C.c_void_p(), C.c_void_p()
hip.hipEventCreate(C.byref(e0)); hip.hipEventCreate(C.byref(e1))
for _ in range(3):
    assert launch() == 0, lib.vx_last_error()
hip.hipDeviceSynchronize()
hip.hipEventRecord(e0, None)
for _ in range(iters):
    launch()
hip.hipEventRecord(e1, None)
hip.hipEventSynchronize(e1)
ms = C.c_float()
hip.hipEventElapsedTime(C.byref(ms), e0, e1)
us = ms.value * 1e3 / iters
out = np.empty(8, np.float32)
hip.hipMemcpy(out.ctypes.data_as(C.c_void_p), Cm, C.c_size_t(32), 2)
print(os.path.basename(sys.argv[1]), form, M, N, K, "us %.1f" % us, "TF/s %.1f" % (2.0 * M * N * K / us / 1e6), "C[0,:4]", out[:4])
