"""Scratch: phase stamps inside mfma_attn_kernel (libvallex_stamps.so = `python vall-e_amd/csrc/build.py --stamps`), workgroup 0,
10 ns ticks: 8 entry (Q fragments loaded), 9 first tile in LDS, 10.. start of iteration 0.., 22 loop done, 23 end.
usage: python3 tests/probes/attn_stamps.py [rows] [heads]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hip = C.CDLL("libamdhip64.so")
lib = C.CDLL(os.path.join(ROOT, "vall-e_amd", "csrc", "libvallex_stamps.so"))
lib.vx_last_error.restype = C.c_char_p
lib.vx_op_attention.argtypes = [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p] + [C.c_int32] * 4 + [C.c_void_p]
lib.vx_debug_read_stamps.argtypes = [C.POINTER(C.c_uint64), C.c_int32]
M = int(sys.argv[1]) if len(sys.argv) > 1 else 1025
H = int(sys.argv[2]) if len(sys.argv) > 2 else 16
d = H * 64


def dmalloc(nbytes, fill=0x3c):
    p = C.c_void_p()
    assert hip.hipMalloc(C.byref(p), C.c_size_t(nbytes)) == 0
    assert hip.hipMemset(p, fill, C.c_size_t(nbytes)) == 0
    return p


qkv, out = dmalloc(M * 3 * d * 2), dmalloc(M * d * 2, 0)
for _ in range(4):
    assert lib.vx_op_attention(1, 1, qkv, out, M, H, 64, -1, None) == 0, lib.vx_last_error()
st = (C.c_uint64 * 24)()
assert lib.vx_debug_read_stamps(st, 24) == 0, lib.vx_last_error()
t = [int(v) for v in st]
us = lambda a, b: round((t[b] - t[a]) * 0.01, 2)
niter = (((M + 63) // 64) + 1) // 2
print("rows", M, "heads", H, "iterations", niter)
print("entry -> first tile in LDS", us(8, 9), "us")
print("iterations:", [us(10 + i, 11 + i) for i in range(min(niter, 12) - 1)], "us each (start to start)")
print("loop total", us(9, 22), "us; merge + store", us(22, 23), "us; kernel", us(8, 23), "us")
