"""Scratch: phase stamps inside the 128x128 GEMM kernel (libvallex_stamps.so = `python vall-e_amd/csrc/build.py --stamps`).
Stamps of workgroup 0, 10 ns ticks: 0 entry, 1 prologue loads issued, 2 first tile in LDS, 3 after three K tiles,
4 K loop done, 5 epilogue stores issued.   usage: python3 tests/probes/gemm_stamps.py"""
import ctypes as C
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hip = C.CDLL("libamdhip64.so")
lib = C.CDLL(os.path.join(ROOT, "vall-e_amd", "csrc", "libvallex_stamps.so"))  # build.py --stamps
lib.vx_last_error.restype = C.c_char_p
lib.vx_op_gemm.argtypes = [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int32] * 4 + [C.c_void_p]
lib.vx_debug_read_stamps.argtypes = [C.POINTER(C.c_uint64), C.c_int32]
os.environ["VX_GEMM_ALG"] = "1"


def dmalloc(nbytes, fill=0x3c):
    p = C.c_void_p()
    assert hip.hipMalloc(C.byref(p), C.c_size_t(nbytes)) == 0
    assert hip.hipMemset(p, fill, C.c_size_t(nbytes)) == 0
    return p


for (M, N, K) in [(128, 128, 64), (128, 128, 1024), (1025, 3072, 256), (1025, 3072, 1024), (1025, 4096, 1024)]:
    A, W, b, Cm = dmalloc(M * K * 2), dmalloc(N * K * 2), dmalloc(N * 4, 0), dmalloc(M * N * 4, 0)
    for _ in range(5):
        assert lib.vx_op_gemm(1, 1, A, W, b, Cm, M, N, K, 0, None) == 0, lib.vx_last_error()
    st = (C.c_uint64 * 6)()
    assert lib.vx_debug_read_stamps(st, 6) == 0, lib.vx_last_error()
    t = [int(v) for v in st]
    d = [round((t[i + 1] - t[i]) * 0.01, 2) for i in range(5)]
    print(json.dumps(dict(M=M, N=N, K=K, us=dict(issue_prologue=d[0], first_tile_in_lds=d[1], first_3_tiles=d[2], rest_of_k_loop=d[3], epilogue=d[4],
                                              total=round((t[5] - t[0]) * 0.01, 2)))), flush=True)
    for p in (A, W, b, Cm):
        hip.hipFree(p)
