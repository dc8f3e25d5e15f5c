"""ctypes binding of libvallex.so (the C ABI in include/vallex.h).

There is deliberately no fallback: if the HIP library is missing or fails to load, importing
the symbols raises, and every product entry point that needs the GPU fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libvallex.so")

VX_PREC_F32, VX_PREC_BF16, VX_PREC_FP8_NAR = 0, 1, 2
BMAX = 64  # slots per engine (csrc/batch_kernels.hpp)
VX_FLAG_TRACE_LOGITS, VX_FLAG_NO_GRAPH, VX_FLAG_SIMPLE_ROWS, VX_FLAG_POST_NORM, VX_FLAG_PRENET, VX_FLAG_VALLF = 1, 2, 4, 8, 16, 32
STOP_REASONS = {0: "none", 1: "eos_argmax", 2: "eos_sample", 3: "length", 4: "max_new"}


class VxConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "struct_size", "d_model", "nhead", "num_layers", "nar_d_model", "nar_nhead", "nar_num_layers",
        "num_quantizers", "prefix_mode", "prepend_bos", "precision", "max_text", "max_audio", "device", "flags", "max_batch")]


class VxDecodeParams(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int32), ("top_k", C.c_int32), ("temperature", C.c_float), ("max_new_tokens", C.c_int32),
        ("exp_noise", C.c_void_p), ("noise_rows", C.c_int64), ("seed", C.c_uint64),
        ("forced", C.c_void_p), ("n_forced", C.c_int32),
    ]


class VxError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"vallex error {code}: {msg}")
        self.code = code


_lib: Optional[C.CDLL] = None

_SIGS = {
    "vx_last_error": (C.c_char_p, []),
    "vx_create": (C.c_int, [C.POINTER(VxConfig), C.POINTER(C.c_void_p)]),
    "vx_destroy": (None, [C.c_void_p]),
    "vx_set_weight": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.POINTER(C.c_int64), C.c_int32]),
    "vx_set_sine_table": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_int64]),
    "vx_finalize_weights": (C.c_int, [C.c_void_p]),
    "vx_ar_prefill": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]),
    "vx_ar_decode": (C.c_int, [C.c_void_p, C.POINTER(VxDecodeParams), C.c_void_p]),
    "vx_ar_result": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "vx_nar": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "vx_nar_ex": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                            C.c_void_p, C.c_int32, C.c_void_p]),
    "vx_nar_continual": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "vx_batch_prefill": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]),
    "vx_batch_prefill_all": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_int32), C.POINTER(C.c_void_p),
                                       C.POINTER(C.c_int32), C.c_void_p]),
    "vx_batch_decode": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(VxDecodeParams), C.c_void_p]),
    "vx_batch_result": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "vx_nar_batch": (C.c_int, [C.c_void_p, C.c_int32] + [C.c_void_p] * 7 + [C.c_void_p]),
    "vx_nar_batch_ex": (C.c_int, [C.c_void_p, C.c_int32] + [C.c_void_p] * 8 + [C.c_void_p]),
    "vx_get_timings": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.c_int32]),
    "vx_read_buffer": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64, C.c_int64]),
    "vx_op_layernorm": (C.c_int, [C.c_int32] + [C.c_void_p] * 6 + [C.c_int32, C.c_int32, C.c_void_p]),
    "vx_op_gemv": (C.c_int, [C.c_int32] + [C.c_void_p] * 4 + [C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "vx_op_gemm": (C.c_int, [C.c_int32, C.c_int32] + [C.c_void_p] * 4 + [C.c_int32] * 4 + [C.c_void_p]),
    "vx_op_gemm_rows": (C.c_int, [C.c_int32] + [C.c_void_p] * 4 + [C.c_int32] * 4 + [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "vx_op_gemm_mx": (C.c_int, [C.c_void_p] * 5 + [C.c_int32] * 5 + [C.c_void_p] * 3),
    "vx_op_layernorm_mx": (C.c_int, [C.c_void_p] * 7 + [C.c_int32, C.c_int32, C.c_void_p]),
    "vx_op_attention": (C.c_int, [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p] + [C.c_int32] * 4 + [C.c_void_p]),
    "vx_op_sample": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_void_p, C.POINTER(C.c_int32), C.c_void_p]),
    "vx_op_convert_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
}

# measurement probes (csrc/probes.h): exported by the probe builds only (`csrc/build.py --probes|--stamps`), never by libvallex.so
_PROBE_SIGS = {
    "vx_debug_launch_floor": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_double)]),
    "vx_debug_stage_chain": (C.c_int, [C.c_int32] * 5 + [C.POINTER(C.c_double)]),
    "vx_debug_l2_fill": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_int32, C.POINTER(C.c_double)]),
    "vx_debug_read_stamps": (C.c_int, [C.POINTER(C.c_uint64), C.c_int32]),
    "vx_debug_kstamps": (C.c_int, [C.c_void_p, C.c_int64, C.POINTER(C.c_int32)]),
}


def declared_symbols():
    return sorted(_SIGS)


def load_library(path: str = LIB_PATH) -> C.CDLL:
    """Loads libvallex.so and attaches the prototypes.  Raises if the library is absent: build it
    with ``python vall-e_amd/csrc/build.py`` (or ``__graft_entry__.build()``)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(path):
        raise FileNotFoundError(f"{path} not built — run `python vall-e_amd/csrc/build.py`; there is no CPU fallback")
    lib = C.CDLL(path)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    for name, (res, args) in _PROBE_SIGS.items():  # present in the probe builds only
        fn = getattr(lib, name, None)
        if fn is not None:
            fn.restype = res
            fn.argtypes = args
    _lib = lib
    return lib


def load_probe_library(stamps: bool = False) -> C.CDLL:
    """Measurement scripts (tests/probes) only: makes the probe build the process's library.  Must be called before anything
    else loads libvallex.so."""
    assert _lib is None, "a library is already loaded in this process"
    return load_library(os.path.join(_HERE, "csrc", "libvallex_stamps.so" if stamps else "libvallex_probes.so"))


def _check(code: int):
    if code != 0:
        raise VxError(code, load_library().vx_last_error().decode())


def _ptr(t) -> Optional[int]:
    if t is None:
        return None
    if isinstance(t, torch.Tensor):
        assert t.is_contiguous()
        return t.data_ptr()
    if isinstance(t, np.ndarray):
        assert t.flags["C_CONTIGUOUS"]
        return t.ctypes.data
    raise TypeError(type(t))


def current_stream_ptr(device) -> Optional[int]:
    s = torch.cuda.current_stream(device).cuda_stream
    return s or None


class Engine:
    """One model replica on one GPU (see include/vallex.h for the contract of each call)."""

    def __init__(self, cfg, precision: str = "bf16", max_text: int = 256, max_audio: int = 2048, device: int = 0,
                 trace_logits: bool = False, no_graph: bool = False, simple_rows: bool = False, max_batch: int = 0):
        self.lib = load_library()
        self.cfg = cfg
        self.device = int(device)
        self.precision = precision
        c = VxConfig()
        c.struct_size = C.sizeof(VxConfig)
        c.d_model, c.nhead, c.num_layers = cfg.decoder_dim, cfg.nhead, cfg.num_decoder_layers
        c.nar_d_model, c.nar_nhead, c.nar_num_layers = cfg.nar_dim, cfg.nar_nhead, cfg.nar_layers
        c.num_quantizers, c.prefix_mode, c.prepend_bos = cfg.num_quantizers, cfg.prefix_mode, int(cfg.prepend_bos)
        c.precision = {"fp32": VX_PREC_F32, "f32": VX_PREC_F32, "bf16": VX_PREC_BF16, "fp8nar": VX_PREC_FP8_NAR}[precision]
        c.max_text, c.max_audio, c.device = max_text, max_audio, self.device
        c.flags = (VX_FLAG_TRACE_LOGITS if trace_logits else 0) | (VX_FLAG_NO_GRAPH if no_graph else 0) | \
                  (VX_FLAG_SIMPLE_ROWS if simple_rows else 0) | (0 if getattr(cfg, "norm_first", True) else VX_FLAG_POST_NORM) | \
                  (VX_FLAG_PRENET if getattr(cfg, "add_prenet", False) else 0) | \
                  (VX_FLAG_VALLF if getattr(cfg, "is_vallf", False) else 0)
        c.max_batch = int(max_batch)
        self.max_text, self.max_audio, self.trace_logits, self.max_batch = max_text, max_audio, trace_logits, int(max_batch)
        self.mfma_rows = c.precision != VX_PREC_F32 and not simple_rows
        h = C.c_void_p()
        _check(self.lib.vx_create(C.byref(c), C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.lib.vx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- weights -----------------------------------------------------------------------------
    def load_state_dict(self, sd, sine_tables: bool = True):
        from .weights import expected_keys, sine_table

        keys = expected_keys(self.cfg)
        for k, shape in keys.items():
            if k.endswith("num_batches_tracked"):  # BatchNorm's step counter: not read by the forward pass
                continue
            t = sd[k].detach().to(torch.float32).contiguous()
            assert tuple(t.shape) == tuple(shape), (k, tuple(t.shape), shape)
            shp = (C.c_int64 * t.dim())(*t.shape)
            _check(self.lib.vx_set_weight(self.h, k.encode(), _ptr(t), shp, t.dim()))
        if sine_tables:
            rows = max(4000, self.max_text + self.max_audio)
            pe = sine_table(rows, self.cfg.decoder_dim)
            _check(self.lib.vx_set_sine_table(self.h, 0, _ptr(pe), rows, self.cfg.decoder_dim))
            if self.cfg.num_quantizers > 1:
                pe = sine_table(rows, self.cfg.nar_dim)
                _check(self.lib.vx_set_sine_table(self.h, 1, _ptr(pe), rows, self.cfg.nar_dim))
        _check(self.lib.vx_finalize_weights(self.h))

    # -- hot path ----------------------------------------------------------------------------
    def ar_prefill(self, text: torch.Tensor, prompt_cb0: torch.Tensor, stream=None):
        text = text.to(torch.int64).contiguous()
        prompt_cb0 = prompt_cb0.to(torch.int64).contiguous()
        self._keep = (text, prompt_cb0)
        _check(self.lib.vx_ar_prefill(self.h, _ptr(text), text.numel(), _ptr(prompt_cb0), prompt_cb0.numel(), stream))

    def ar_decode(self, top_k: int = -100, temperature: float = 1.0, exp_noise: Optional[torch.Tensor] = None,
                  seed: int = 0, max_new_tokens: int = -1, forced: Optional[torch.Tensor] = None, stream=None):
        p = VxDecodeParams()
        p.struct_size = C.sizeof(VxDecodeParams)
        p.top_k, p.temperature, p.max_new_tokens, p.seed = int(top_k), float(temperature), int(max_new_tokens), int(seed)
        keep = []
        if exp_noise is not None:
            exp_noise = exp_noise.to(torch.float32).contiguous()
            assert exp_noise.dim() == 2 and exp_noise.shape[1] == 1025
            p.exp_noise, p.noise_rows = _ptr(exp_noise), exp_noise.shape[0]
            keep.append(exp_noise)
        if forced is not None:
            forced = forced.to(torch.int64).contiguous()
            p.forced, p.n_forced = (_ptr(forced) if forced.numel() else _ptr(torch.zeros(1, dtype=torch.int64))), forced.numel()
            keep.append(forced)
        _check(self.lib.vx_ar_decode(self.h, C.byref(p), stream))

    def ar_result(self):
        n, reason, npass = C.c_int32(), C.c_int32(), C.c_int32()
        _check(self.lib.vx_ar_result(self.h, None, 0, C.byref(n), C.byref(reason), C.byref(npass)))
        toks = torch.empty(n.value, dtype=torch.int64)
        _check(self.lib.vx_ar_result(self.h, _ptr(toks), n.value, C.byref(n), C.byref(reason), C.byref(npass)))
        return toks, reason.value, npass.value

    def nar(self, text_nar: torch.Tensor, prompts: torch.Tensor, ar_tokens: torch.Tensor, out_device=None, stream=None,
            continual: bool = False, forced_codes: Optional[torch.Tensor] = None, stage_logits: bool = False):
        """prompts: (P, Q); returns codes (T, Q) int64 on ``out_device`` (default: prompts' device).  ``continual``:
        the NAR body of VALLE.continual (vx_nar_continual).  Parity-test options (vx_nar_ex): ``forced_codes`` (T, Q) feeds
        every stage the given codes of the earlier stages; ``stage_logits=True`` also returns the (Q-1, T, 1024) logits."""
        text_nar = text_nar.to(torch.int64).contiguous()
        prompts = prompts.to(torch.int64).contiguous()
        ar_tokens = ar_tokens.to(torch.int64).contiguous()
        T, Q = ar_tokens.numel(), self.cfg.num_quantizers
        out = torch.empty((T, Q), dtype=torch.int64, device=out_device if out_device is not None else prompts.device)
        if forced_codes is None and not stage_logits:
            fn = self.lib.vx_nar_continual if continual else self.lib.vx_nar
            _check(fn(self.h, _ptr(text_nar), text_nar.numel(), _ptr(prompts), prompts.shape[0], _ptr(ar_tokens), T, _ptr(out), stream))
            return out
        fc = None if forced_codes is None else forced_codes.to(torch.int64).contiguous()
        assert fc is None or tuple(fc.shape) == (T, Q)
        lg = torch.empty((max(Q - 1, 0), T, 1024), dtype=torch.float32) if stage_logits else None
        _check(self.lib.vx_nar_ex(self.h, _ptr(text_nar), text_nar.numel(), _ptr(prompts), prompts.shape[0], _ptr(ar_tokens), T,
                                  _ptr(out), _ptr(fc), _ptr(lg), int(continual), stream))
        return (out, lg) if stage_logits else out

    # -- batched decode (BASELINE configs[2]) -------------------------------------------------------
    def batch_prefill(self, slot: int, text: torch.Tensor, prompt_cb0: torch.Tensor, stream=None):
        text = text.to(torch.int64).contiguous()
        prompt_cb0 = prompt_cb0.to(torch.int64).contiguous()
        _check(self.lib.vx_batch_prefill(self.h, slot, _ptr(text), text.numel(), _ptr(prompt_cb0), prompt_cb0.numel(), stream))

    def batch_prefill_all(self, texts, prompts_cb0, stream=None):
        """Slots 0..n-1 prefilled in one pass over the concatenated rows (bf16 engines)."""
        n = len(texts)
        texts = [t.to(torch.int64).contiguous() for t in texts]
        proms = [p.to(torch.int64).contiguous() for p in prompts_cb0]
        tp = (C.c_void_p * n)(*[_ptr(t) for t in texts])
        pp = (C.c_void_p * n)(*[_ptr(p) for p in proms])
        S = (C.c_int32 * n)(*[t.numel() for t in texts])
        P = (C.c_int32 * n)(*[p.numel() for p in proms])
        _check(self.lib.vx_batch_prefill_all(self.h, n, tp, S, pp, P, stream))

    def batch_decode(self, n_slots: int, top_k=-100, temperature=1.0, seeds=None, exp_noise=None, forced=None,
                     max_new_tokens=-1, stream=None):
        """exp_noise / forced: optional per-slot lists of DEVICE tensors (kept alive here for the call)."""
        arr = (VxDecodeParams * n_slots)()
        keep = []
        for b in range(n_slots):
            p = arr[b]
            p.struct_size = C.sizeof(VxDecodeParams)
            p.top_k, p.temperature, p.max_new_tokens = int(top_k), float(temperature), int(max_new_tokens)
            p.seed = int(seeds[b]) if seeds is not None else b + 1
            if exp_noise is not None and exp_noise[b] is not None:
                t = exp_noise[b].to(torch.float32).contiguous()
                assert t.is_cuda and t.shape[1] == 1025
                p.exp_noise, p.noise_rows = _ptr(t), t.shape[0]
                keep.append(t)
            if forced is not None and forced[b] is not None:
                t = forced[b].to(torch.int64).contiguous()
                assert t.is_cuda
                p.forced, p.n_forced = _ptr(t), t.numel()
                keep.append(t)
        _check(self.lib.vx_batch_decode(self.h, n_slots, arr, stream))

    def batch_result(self, slot: int):
        n, reason = C.c_int32(), C.c_int32()
        _check(self.lib.vx_batch_result(self.h, slot, None, 0, C.byref(n), C.byref(reason)))
        toks = torch.empty(n.value, dtype=torch.int64)
        _check(self.lib.vx_batch_result(self.h, slot, _ptr(toks), n.value, C.byref(n), C.byref(reason)))
        return toks, reason.value

    def nar_batch(self, texts, prompts, tokens, out_device=None, stream=None, forced_codes=None):
        """lists of per-utterance tensors (as for ``nar``) -> list of (T_i, Q) int64 code tensors.  ``forced_codes``: optional
        list of (T_i, Q) tensors, per-stage teacher forcing as in ``nar`` (vx_nar_batch_ex)."""
        n, Q = len(texts), self.cfg.num_quantizers
        texts = [t.to(torch.int64).contiguous() for t in texts]
        prompts = [p.to(torch.int64).contiguous() for p in prompts]
        tokens = [t.to(torch.int64).contiguous() for t in tokens]
        outs = [torch.empty((t.numel(), Q), dtype=torch.int64, device=out_device if out_device is not None else p.device)
                for t, p in zip(tokens, prompts)]
        ptrs = lambda ts: (C.c_void_p * n)(*[_ptr(t) for t in ts])
        ints = lambda vs: (C.c_int32 * n)(*vs)
        fc = None
        if forced_codes is not None:
            forced_codes = [f.to(torch.int64).contiguous() for f in forced_codes]
            fc = ptrs(forced_codes)
        _check(self.lib.vx_nar_batch_ex(self.h, n, ptrs(texts), ints([t.numel() for t in texts]), ptrs(prompts),
                                        ints([p.shape[0] for p in prompts]), ptrs(tokens), ints([t.numel() for t in tokens]),
                                        ptrs(outs), fc, stream))
        return outs

    def timings(self):
        buf = (C.c_double * 9)()
        _check(self.lib.vx_get_timings(self.h, buf, 9))
        return dict(prefill_ms=buf[0], decode_ms=buf[1], nar_ms=buf[2], n_pass=int(buf[3]), launches=int(buf[4]),
                    batch_decode_ms=buf[5], batch_launches=int(buf[6]), nar_gemm_ms=buf[7], nar_gemm_flops=buf[8])

    def read(self, name: str, shape, dtype=torch.float32, offset_bytes: int = 0) -> torch.Tensor:
        out = torch.empty(shape, dtype=dtype)
        _check(self.lib.vx_read_buffer(self.h, name.encode(), _ptr(out), offset_bytes, out.numel() * out.element_size()))
        return out


# ---- kernel-level ops (parity tests call the HIP kernels through the same C ABI) -------------------
def _prec(t_or_name) -> int:
    if isinstance(t_or_name, str):
        return VX_PREC_BF16 if t_or_name == "bf16" else VX_PREC_F32
    return VX_PREC_BF16 if t_or_name.dtype == torch.bfloat16 else VX_PREC_F32


def op_layernorm(x, gamma, beta, ada_w=None, ada_b=None, out_dtype=torch.float32):
    lib = load_library()
    rows, d = x.shape
    out = torch.empty((rows, d), dtype=out_dtype, device=x.device)
    _check(lib.vx_op_layernorm(_prec(out), _ptr(x), _ptr(gamma), _ptr(beta), _ptr(ada_w), _ptr(ada_b), _ptr(out), rows, d,
                               current_stream_ptr(x.device)))
    return out


def op_gemv(W, bias, x, relu=False):
    lib = load_library()
    N, K = W.shape
    y = torch.empty(N, dtype=torch.float32, device=x.device)
    _check(lib.vx_op_gemv(_prec(W), _ptr(W), _ptr(bias), _ptr(x), _ptr(y), N, K, int(relu), current_stream_ptr(x.device)))
    return y


def op_gemm(A, W, bias=None, relu=False, mfma=False):
    lib = load_library()
    M, K = A.shape
    N = W.shape[0]
    Cm = torch.empty((M, N), dtype=torch.float32, device=A.device)
    _check(lib.vx_op_gemm(_prec(A), int(mfma), _ptr(A), _ptr(W), _ptr(bias), _ptr(Cm), M, N, K, int(relu),
                          current_stream_ptr(A.device)))
    return Cm


def op_gemm_rows(A, W, bias, relu=False, resid=None, vt_cols=0):
    """The engine's row-path GEMM forms (vx_op_gemm_rows) on bf16 A (M, K), W (N, K): resid=None -> (C bf16 (M, N), V^T copy of
    the last vt_cols columns (vt_cols, ld) or None); resid = fp32 (M, N) -> updated in place (resid += A.W^T + bias), returned."""
    lib = load_library()
    M, K = A.shape
    N = W.shape[0]
    if resid is not None:
        _check(lib.vx_op_gemm_rows(1, _ptr(A), _ptr(W), _ptr(bias), _ptr(resid), M, N, K, 0, None, 0, 0, current_stream_ptr(A.device)))
        return resid
    Cm = torch.empty((M, N), dtype=torch.bfloat16, device=A.device)
    ld = (M + 255) // 256 * 256
    vt = torch.zeros((vt_cols, ld), dtype=torch.bfloat16, device=A.device) if vt_cols else None
    _check(lib.vx_op_gemm_rows(0, _ptr(A), _ptr(W), _ptr(bias), _ptr(Cm), M, N, K, int(relu), _ptr(vt), N - vt_cols, ld,
                               current_stream_ptr(A.device)))
    return Cm, vt


def op_gemm_mx(A, W, bias=None, relu=False, out_mx=False, return_quant=False):
    """MXFP8 GEMM (vx_op_gemm_mx): fp32 A (M, K), W (N, K) on the GPU.  out_mx=False -> C (M, N) fp32; True -> (e4m3 bytes (M, N),
    E8M0 scales (N/32, ld)) of ReLU(C + bias).  return_quant adds the engine's quantisation of A: (bytes (M, K), scales (K/32, ld))."""
    lib = load_library()
    M, K = A.shape
    N = W.shape[0]
    ld = (M + 255) // 256 * 256
    dev = A.device
    if out_mx:
        c = torch.empty((M, N), dtype=torch.uint8, device=dev)
        sc = torch.empty((N // 32, ld), dtype=torch.uint8, device=dev)
    else:
        c, sc = torch.empty((M, N), dtype=torch.float32, device=dev), None
    qa = torch.empty((M, K), dtype=torch.uint8, device=dev) if return_quant else None
    sa = torch.empty((K // 32, ld), dtype=torch.uint8, device=dev) if return_quant else None
    _check(lib.vx_op_gemm_mx(_ptr(A), _ptr(W), _ptr(bias), _ptr(c), _ptr(sc), M, N, K, int(relu), 2 if out_mx else 0, _ptr(qa), _ptr(sa),
                             current_stream_ptr(dev)))
    res = (c, sc) if out_mx else c
    return (res, qa, sa) if return_quant else res


def op_layernorm_mx(x, gamma, beta, ada_w=None, ada_b=None):
    lib = load_library()
    rows, d = x.shape
    ld = (rows + 255) // 256 * 256
    q = torch.empty((rows, d), dtype=torch.uint8, device=x.device)
    sc = torch.empty((d // 32, ld), dtype=torch.uint8, device=x.device)
    _check(lib.vx_op_layernorm_mx(_ptr(x), _ptr(gamma), _ptr(beta), _ptr(ada_w), _ptr(ada_b), _ptr(q), _ptr(sc), rows, d,
                                  current_stream_ptr(x.device)))
    return q, sc


def op_attention(qkv, nhead, text_len=-1, mfma=False):
    lib = load_library()
    rows, d3 = qkv.shape
    d = d3 // 3
    out = torch.empty((rows, d), dtype=qkv.dtype, device=qkv.device)
    _check(lib.vx_op_attention(_prec(qkv), int(mfma), _ptr(qkv), _ptr(out), rows, nhead, d // nhead, text_len,
                               current_stream_ptr(qkv.device)))
    return out


def op_sample(logits, top_k, temperature, exp_noise):
    lib = load_library()
    out = (C.c_int32 * 2)()
    _check(lib.vx_op_sample(_ptr(logits), logits.numel(), int(top_k), float(temperature), _ptr(exp_noise), out,
                            current_stream_ptr(logits.device)))
    return out[0], out[1]


def launch_floor(n_kernels=62, grid=256, block=256, iters=200):  # the three probes below need load_probe_library()
    lib = load_library()
    out = (C.c_double * 2)()
    _check(lib.vx_debug_launch_floor(n_kernels, grid, block, iters, out))
    return dict(graph_us_per_kernel=out[0], eager_us_per_kernel=out[1])


def stage_chain(nwg=256, stages=60, rows=12, mode=2, iters=20):
    lib = load_library()
    out = (C.c_double * 4)()
    _check(lib.vx_debug_stage_chain(nwg, stages, rows, mode, iters, out))
    return dict(us_per_launch=out[0], us_per_stage=out[1], max_err=out[2], spin_timeout=int(out[3]))


def l2_fill(grid=256, threads=256, unroll=8, region_bytes=2 << 20, iters=200):
    lib = load_library()
    out = (C.c_double * 3)()
    _check(lib.vx_debug_l2_fill(grid, threads, unroll, region_bytes, iters, out))
    return dict(gbs=out[0], bytes_per_clk_per_cu=out[1], ghz=out[2])
