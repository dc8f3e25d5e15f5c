"""Per-kernel anatomy of the batch-1 AR decode step INSIDE the hipGraph replay (no profiler in the loop).

    python vall-e_amd/csrc/build.py --stamps && python tests/probes/ar_step_stamps.py [out.json]

Runs the bench workload (BASELINE configs[1]: d=1024 L=12 bf16, S=47, P=225, top-k 10) on libvallex_stamps.so, whose decode
kernels record s_memrealtime (100 MHz) into a 16-pass ring (common.hpp VX_KSTAMP): one wave of workgroup 0 its (slightly late)
start, and in mode 2 one wave of the last workgroup its end.  period = next kernel's entry - this kernel's entry (mode 1, the
least intrusive); body / gap = the split of that period by the mode-2 exit stamps.  The periods sum to the step, checked against
the HIP-event step time of the same run (vx_get_timings) and quoted next to the un-stamped library's step time (bench.py).
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import ctypes as C

import torch

import valle_amd  # noqa: F401
from valle_amd.engine import load_probe_library

lib = load_probe_library(stamps=True)
from valle_amd.config import ModelConfig
from valle_amd.models import VALLE
from valle_amd.weights import synthetic_inputs, synthetic_state_dict

L = 12
cfg = ModelConfig(decoder_dim=1024, nhead=16, num_decoder_layers=L, prefix_mode=1)
m = VALLE(1024, 16, L, prefix_mode=1, precision="bf16", max_text=64, max_audio=1024, print_eos=False)
m.load_state_dict(synthetic_state_dict(cfg, seed=0))
m.to("cuda:0").eval()
eng = m.engine()
x, xl, y = synthetic_inputs(47, 225, 8, seed=1)
x, xl, y = x.cuda(), xl.cuda(), y.cuda()
for i in range(2):  # warm-up (graph capture, clocks)
    torch.manual_seed(1234 + i)
    m.inference(x, xl, y, None, top_k=10)

names = ["sample+embed"] + [f"L{l}.{k}" for l in range(L) for k in ("qkv", "attn", "out", "ffn1", "ffn2")] + ["head"]
NK = len(names)  # 62


def collect(n_new, seed):
    """one decode stopped after n_new tokens with the stamps armed; the ring then holds its last 16 passes."""
    assert lib.vx_debug_kstamps(None, 0, None) == 0
    torch.manual_seed(seed)
    m.inference(x, xl, y, None, top_k=10, max_new_tokens=n_new)
    t = eng.timings()
    ring = np.zeros((16, 64), dtype=np.uint64)
    assert lib.vx_debug_kstamps(ring.ctypes.data_as(C.c_void_p), ring.nbytes, None) == 0
    entry = ring[:, :NK].astype(np.int64)  # (16 passes, NK kernels)
    ok = (entry > 0).all(axis=1)
    order = [p for p in np.argsort(entry[:, 0]) if ok[p]][2:-2]  # time order; drop the oldest two and the newest two (the last step stops early)
    return entry, order, 1e3 * t["decode_ms"] / t["launches"]


CTX = (200, 400, 600, 753)
per, hip = [], []
for rep, n_new in enumerate(CTX):
    entry, order, hip_us = collect(n_new, 77 + rep)
    seq = np.array([np.concatenate([entry[order[i]], entry[order[i + 1]][:1]]) for i in range(len(order) - 1)])
    per.append(np.diff(seq, axis=1).mean(0) * 0.01)  # us: this kernel's stamp -> the next kernel's stamp
    hip.append(hip_us)
period = np.mean(per, axis=0)

kinds = {}
for i, n in enumerate(names):
    d = kinds.setdefault(n.split(".")[-1], dict(count=0, period_us=0.0))
    d["count"] += 1
    d["period_us"] += float(period[i])
for d in kinds.values():
    d["period_us_each"] = round(d["period_us"] / d["count"], 3)
    d["period_us"] = round(d["period_us"], 2)
out = dict(
    what=("AR decode step, batch 1, cfg1 (d=1024 L=12 bf16), inside the hipGraph replay: s_memrealtime stamps of the 62 kernels (one extra idle "
          "workgroup per kernel records the time), mean over passes near ctx 472 / 672 / 872 / 1025.  period = this kernel's stamp to the next "
          "kernel's stamp = the kernel's body + the boundary behind it; the periods sum to the step"),
    kernels=NK, sum_period_us=round(float(period.sum()), 2), hip_event_step_us=round(float(np.mean(hip)), 2),
    by_kind=kinds, per_kernel=[dict(name=n, period_us=round(float(p_), 3)) for n, p_ in zip(names, period)],
    per_context=[dict(ctx_end=47 + 225 + c, hip_event_step_us=round(h, 2), sum_period_us=round(float(p_.sum()), 2)) for c, h, p_ in zip(CTX, hip, per)],
)
s = json.dumps(out, indent=1)
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write(s + "\n")
print(json.dumps({k: out[k] for k in ("sum_period_us", "hip_event_step_us", "by_kind")}))
