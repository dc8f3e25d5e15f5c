"""Phase anatomy of the XCD-sharded decode step (ar_tp.hpp) from in-kernel stamps (s_memrealtime, 100 MHz).

    python vall-e_amd/csrc/build.py --stamps && python tests/probes/tp_stamps.py [out.json]

Attention half: 0 row loads issued, 1 LN done, 2 q/k/v published, 3 head gathered, 4 partial published, 5 partials gathered,
6 out-projection slice stored.  Feed-forward half: 0 row loads issued, 1 LN done, 2 hidden units published, 3 gathered,
4 linear2 slice stored.  Per phase: min / median / max over the 256 workgroups of the time since the launch's first stamp,
averaged over layers 1..11 of the last decode step; `period` = first stamp of a launch to the first stamp of the next.
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
NAMES = [["loads_issued", "ln_done", "qkv_published", "head_gathered", "partial_published", "partials_gathered", "out_stored", "row_arrived",
          "att_start", "att_loop_done", "att_wave_merged", "att_barrier", "x_arrived"],
         ["loads_issued", "ln_done", "hidden_published", "hidden_gathered", "out_stored", None, None, "row_arrived"]]
res = {}
for n_new in (100, 700):
    assert lib.vx_debug_fqstamps(None, 0) == 0
    torch.manual_seed(7)
    m.inference(x, xl, y, None, top_k=10, max_new_tokens=n_new)
    t = m.engine().timings()
    buf = np.zeros((2, 16, 256, 16), dtype=np.uint64)
    assert lib.vx_debug_fqstamps(buf.ctypes.data_as(C.c_void_p), buf.nbytes) == 0
    st = buf[:, :L].astype(np.int64)
    first = st[:, :, :, 0].min(axis=2)  # (kind, layer)
    out = dict(step_us=round(1e3 * t["decode_ms"] / t["launches"], 2))
    out["period_attn_us"] = round(float(np.median((first[1, 1:] - first[0, 1:]) * 0.01)), 2)
    out["period_ffn_us"] = round(float(np.median((first[0, 2:] - first[1, 1:-1]) * 0.01)), 2)
    for kind in (0, 1):
        rows = {}
        for ph, name in enumerate(NAMES[kind]):
            if name is None:
                continue
            rel = (st[kind, 1:, :, ph] - first[kind, 1:, None]) * 0.01
            rows[name] = [round(float(rel.min(axis=1).mean()), 2), round(float(np.median(rel, axis=1).mean()), 2), round(float(rel.max(axis=1).mean()), 2)]
        out["attn" if kind == 0 else "ffn"] = rows
    res[n_new] = out
    print(n_new, json.dumps(out), flush=True)
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], "w"), indent=1)
# per key-split (j = local index % 16) view of the attention half's compute phase, last run
wg = np.arange(256)
jj = (wg >> 3) & 15
att = (st[0, 1:, :, 4] - st[0, 1:, :, 3]) * 0.01   # head gathered -> partial published
lnq = (st[0, 1:, :, 2] - st[0, 1:, :, 1]) * 0.01   # LN done -> q/k/v published
print("attention compute by split j:", [round(float(att[:, jj == j].mean()), 2) for j in range(16)])
print("LN -> publish by split j:   ", [round(float(lnq[:, jj == j].mean()), 2) for j in range(16)])
print("attention compute by XCD:    ", [round(float(att[:, (wg & 7) == x].mean()), 2) for x in range(8)])
