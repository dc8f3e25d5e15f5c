"""Phase anatomy of the fused QKV + attention launch of the decode step (ar_fused.hpp), from in-kernel stamps.

    python vall-e_amd/csrc/build.py --stamps && python tests/probes/fq_stamps.py [out.json]

Every workgroup's thread 0 records s_memrealtime (100 MHz) at: 0 entry, 1 q/k/v published, 2 head gather done, 3 partial
published, 4 (combiner workgroups) partial gather done, 5 output stored.  Reported per phase: min / median / max over the
workgroups of the time since the launch's first entry stamp, averaged over the layers of the last decode step.
"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

import valle_amd  # noqa: F401
from valle_amd.engine import load_probe_library

lib = load_probe_library(stamps=True)
lib.vx_debug_fqstamps.argtypes = [C.c_void_p, C.c_int64]
from valle_amd.config import ModelConfig
from valle_amd.models import VALLE
from valle_amd.weights import synthetic_inputs, synthetic_state_dict

L = 12
cfg = ModelConfig(decoder_dim=1024, nhead=16, num_decoder_layers=L, prefix_mode=1)
m = VALLE(1024, 16, L, prefix_mode=1, precision="bf16", max_text=64, max_audio=1024, print_eos=False)
m.load_state_dict(synthetic_state_dict(cfg, seed=0))
m.to("cuda:0").eval()
x, xl, y = synthetic_inputs(47, 225, 8, seed=1)
x, xl, y = x.cuda(), xl.cuda(), y.cuda()
for i in range(2):
    torch.manual_seed(1234 + i)
    m.inference(x, xl, y, None, top_k=10)
res = {}
for n_new in (100, 400, 700):
    assert lib.vx_debug_fqstamps(None, 0) == 0
    torch.manual_seed(7)
    m.inference(x, xl, y, None, top_k=10, max_new_tokens=n_new)
    t = m.engine().timings()
    buf = np.zeros((2, 16, 256, 16), dtype=np.uint64)[0]
    assert lib.vx_debug_fqstamps(buf.ctypes.data_as(C.c_void_p), buf.nbytes) == 0
    st = buf[:L].astype(np.int64)
    t0 = st[:, :, 0].min(axis=1, keepdims=True)
    rows = {}
    for ph, name in enumerate(["entry", "published", "gathered", "partial_out", "comb_gathered", "out_stored"]):
        v = st[:, :, ph]
        ok = v > 0
        rel = np.where(ok, v - t0, 0) * 0.01
        per_layer = [(rel[l][ok[l]].min(), np.median(rel[l][ok[l]]), rel[l][ok[l]].max()) for l in range(L) if ok[l].any()]
        a = np.array(per_layer)
        rows[name] = dict(min=round(float(a[:, 0].mean()), 2), med=round(float(a[:, 1].mean()), 2), max=round(float(a[:, 2].mean()), 2))
    nxt = (st[1:, :, 0].min(axis=1) - st[:-1, :, 0].min(axis=1)) * 0.01  # entry to next layer's entry
    res[n_new] = dict(step_us=round(1e3 * t["decode_ms"] / t["launches"], 2), layer_period_us=round(float(np.median(nxt)), 2), phases=rows)
    print(n_new, json.dumps(res[n_new]), flush=True)
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], "w"), indent=1)
