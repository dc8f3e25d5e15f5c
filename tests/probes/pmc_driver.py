"""Torch-free driver of the C ABI for counter collection (rocprofv3 --pmc crashes inside torch's
start-up on this image): random weights of the cfg1 geometry, one prefill, N eager AR steps.
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d OUT -- python3 tests/probes/pmc_driver.py 40
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lib = C.CDLL(os.path.join(ROOT, "vall-e_amd", "csrc", "libvallex.so"))
lib.vx_last_error.restype = C.c_char_p
NSTEP = int(sys.argv[1]) if len(sys.argv) > 1 else 40
NAR = len(sys.argv) > 2 and sys.argv[2] == "nar"
d, H, L, Q = 1024, 16, 12, 8


def ck(rc):
    if rc:
        raise RuntimeError(lib.vx_last_error().decode())


class Cfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("struct_size d_model nhead num_layers nar_d_model nar_nhead nar_num_layers "
                                         "num_quantizers prefix_mode prepend_bos precision max_text max_audio device flags max_batch").split()]


class Dec(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("top_k", C.c_int32), ("temperature", C.c_float), ("max_new_tokens", C.c_int32),
                ("exp_noise", C.c_void_p), ("noise_rows", C.c_int64), ("seed", C.c_uint64), ("forced", C.c_void_p), ("n_forced", C.c_int32)]


def keys():
    def enc(pre, n, ada):
        for i in range(n):
            p = f"{pre}.layers.{i}"
            yield p + ".self_attn.in_proj_weight", (3 * d, d)
            yield p + ".self_attn.in_proj_bias", (3 * d,)
            yield p + ".self_attn.out_proj.weight", (d, d)
            yield p + ".self_attn.out_proj.bias", (d,)
            yield p + ".linear1.weight", (4 * d, d)
            yield p + ".linear1.bias", (4 * d,)
            yield p + ".linear2.weight", (d, 4 * d)
            yield p + ".linear2.bias", (d,)
            for nn in ("norm1", "norm2"):
                if ada:
                    yield f"{p}.{nn}.project_layer.weight", (2 * d, d)
                    yield f"{p}.{nn}.project_layer.bias", (2 * d,)
                    yield f"{p}.{nn}.norm.weight", (d,)
                    yield f"{p}.{nn}.norm.bias", (d,)
                else:
                    yield f"{p}.{nn}.weight", (d,)
                    yield f"{p}.{nn}.bias", (d,)
        if ada:
            yield pre + ".norm.project_layer.weight", (2 * d, d)
            yield pre + ".norm.project_layer.bias", (2 * d,)
            yield pre + ".norm.norm.weight", (d,)
            yield pre + ".norm.norm.bias", (d,)
        else:
            yield pre + ".norm.weight", (d,)
            yield pre + ".norm.bias", (d,)

    yield "ar_text_embedding.word_embeddings.weight", (512, d)
    yield "nar_text_embedding.word_embeddings.weight", (512, d)
    yield "ar_audio_embedding.word_embeddings.weight", (1025, d)
    yield "ar_text_position.alpha", (1,)
    yield "ar_audio_position.alpha", (1,)
    yield from enc("ar_decoder", L, False)
    yield "ar_predict_layer.weight", (1025, d)
    yield "nar_audio_embeddings.0.word_embeddings.weight", (1025, d)
    for j in range(1, Q):
        yield f"nar_audio_embeddings.{j}.word_embeddings.weight", (1024, d)
    yield "nar_text_position.alpha", (1,)
    yield "nar_audio_position.alpha", (1,)
    yield from enc("nar_decoder", L, True)
    for j in range(Q - 1):
        yield f"nar_predict_layers.{j}.weight", (1024, d)
    for j in range(Q - 1):
        yield f"nar_stage_embeddings.{j}.word_embeddings.weight", (1, d)


c = Cfg(C.sizeof(Cfg), d, H, L, d, H, L, Q, 1, 0, 1, 64, 1024, 0, 2, 0)  # bf16, flags: VX_FLAG_NO_GRAPH
h = C.c_void_p()
ck(lib.vx_create(C.byref(c), C.byref(h)))
rng = np.random.default_rng(0)
for k, shp in keys():
    if k.endswith("alpha"):
        t = np.ones(1, np.float32)
    elif len(shp) == 2 and "embedding" not in k:
        t = (rng.standard_normal(shp, dtype=np.float32) / np.sqrt(shp[1])).astype(np.float32)
    elif len(shp) == 2:
        t = rng.standard_normal(shp, dtype=np.float32)
    elif k.endswith("weight"):
        t = np.ones(shp, np.float32)
    else:
        t = np.zeros(shp, np.float32)
    if k == "ar_predict_layer.weight":
        t[1024] = 0
    s = (C.c_int64 * len(shp))(*shp)
    ck(lib.vx_set_weight(h, k.encode(), t.ctypes.data_as(C.c_void_p), s, len(shp)))
ck(lib.vx_finalize_weights(h))
text = rng.integers(3, 100, 47).astype(np.int64)
prom = rng.integers(0, 1024, (225, 8)).astype(np.int64)
cb0 = np.ascontiguousarray(prom[:, 0])
ck(lib.vx_ar_prefill(h, text.ctypes.data_as(C.c_void_p), 47, cb0.ctypes.data_as(C.c_void_p), 225, None))
p = Dec(C.sizeof(Dec), 10, 1.0, NSTEP, None, 0, 1234, None, 0)
ck(lib.vx_ar_decode(h, C.byref(p), None))
n = C.c_int32()
ck(lib.vx_ar_result(h, None, 0, C.byref(n), None, None))
toks = np.zeros(n.value, np.int64)
ck(lib.vx_ar_result(h, toks.ctypes.data_as(C.c_void_p), n.value, C.byref(n), None, None))
print("decoded", n.value, "tokens", toks[:8])
if NAR:
    full = rng.integers(0, 1024, 753).astype(np.int64)
    codes = np.zeros((753, 8), np.int64)
    ck(lib.vx_nar(h, text.ctypes.data_as(C.c_void_p), 47, prom.ctypes.data_as(C.c_void_p), 225,
                  full.ctypes.data_as(C.c_void_p), 753, codes.ctypes.data_as(C.c_void_p), None))
    print("nar ok", codes[:2])
lib.vx_destroy(h)
