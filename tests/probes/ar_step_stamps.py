"""Per-kernel anatomy of the batch-1 AR decode step INSIDE the hipGraph replay (no profiler in the loop).

    python vall-e_amd/csrc/build.py --stamps && python tests/probes/ar_step_stamps.py [out.json]

Runs the bench workload (BASELINE configs[1]: d=1024 L=12 bf16, S=47, P=225, top-k 10) on libvallex_stamps.so, whose decode
kernels record s_memrealtime (100 MHz) at wave start and wave end into a 16-pass ring (common.hpp VX_KSTAMP).  Per kernel of
the step: entry = first wave's start, exit = last wave's end; body = exit - entry; gap = next kernel's entry - this exit (the
launch boundary as the kernels see it).  The sum over the 62 kernels + 62 gaps is the step period, checked against the HIP-event
step time of the same run (vx_get_timings) and quoted next to the un-stamped library's step time (bench.py).
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
res = []
for rep, n_new in enumerate((200, 400, 600, 753)):  # stop the decode at different context lengths: the ring holds its last 16 passes
    assert lib.vx_debug_kstamps(None, 0, None) == 0
    torch.manual_seed(77 + rep)
    m.inference(x, xl, y, None, top_k=10, max_new_tokens=n_new)
    t = eng.timings()
    ring = np.zeros((16, 64, 1024, 2), dtype=np.uint64)
    assert lib.vx_debug_kstamps(ring.ctypes.data_as(C.c_void_p), ring.nbytes, None) == 0
    ring = ring[:, :NK].astype(np.int64)
    valid = ring[..., 0] > 0
    entry = np.where(valid, ring[..., 0], np.iinfo(np.int64).max).min(axis=2)  # (16, NK)
    exit_ = np.where(valid, ring[..., 1], 0).max(axis=2)
    ok = valid.any(axis=2).all(axis=1)
    # passes in time order; drop the oldest two and the newest two (the last step stops early)
    order = [p for p in np.argsort(entry[:, 0]) if ok[p]][2:-2]
    body = np.array([(exit_[p] - entry[p]) for p in order]) * 0.01          # us
    gap_in = np.array([(entry[p][1:] - exit_[p][:-1]) for p in order]) * 0.01
    period = np.diff(np.array([entry[p][0] for p in order])) * 0.01
    step_gap = np.array([entry[order[i + 1]][0] - exit_[order[i]][NK - 1] for i in range(len(order) - 1)]) * 0.01
    res.append(dict(ctx_end=47 + 225 + n_new, passes_used=len(order), hip_event_step_us=1e3 * t["decode_ms"] / t["launches"],
                    period_us=float(period.mean()), body_us=body.mean(0).tolist(), gap_us=gap_in.mean(0).tolist() + [float(step_gap.mean())]))

# aggregate over the four context lengths
body = np.mean([r["body_us"] for r in res], axis=0)
gap = np.mean([r["gap_us"] for r in res], axis=0)
kinds = {}
for i, n in enumerate(names):
    k = n.split(".")[-1]
    kinds.setdefault(k, dict(count=0, body_us=0.0, gap_after_us=0.0))
    kinds[k]["count"] += 1
    kinds[k]["body_us"] += float(body[i])
    kinds[k]["gap_after_us"] += float(gap[i])
for k in kinds.values():
    k["body_us_each"] = round(k["body_us"] / k["count"], 3)
    k["gap_after_us_each"] = round(k["gap_after_us"] / k["count"], 3)
    k["body_us"] = round(k["body_us"], 2)
    k["gap_after_us"] = round(k["gap_after_us"], 2)
out = dict(
    what="AR decode step, batch 1, cfg1 (d=1024 L=12 bf16): in-graph s_memrealtime stamps of the 62 kernels, mean over passes near ctx 472 / 672 / 872 / 1025",
    kernels=NK, sum_body_us=round(float(body.sum()), 2), sum_gap_us=round(float(gap.sum()), 2),
    sum_us=round(float(body.sum() + gap.sum()), 2),
    stamped_period_us=round(float(np.mean([r["period_us"] for r in res])), 2),
    stamped_hip_event_step_us=round(float(np.mean([r["hip_event_step_us"] for r in res])), 2),
    by_kind=kinds, per_kernel=[dict(name=n, body_us=round(float(b), 3), gap_after_us=round(float(g), 3)) for n, b, g in zip(names, body, gap)],
    per_context=[dict(ctx_end=r["ctx_end"], period_us=round(r["period_us"], 2), hip_event_step_us=round(r["hip_event_step_us"], 2)) for r in res],
)
s = json.dumps(out, indent=1)
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write(s + "\n")
print(json.dumps({k: out[k] for k in ("sum_body_us", "sum_gap_us", "sum_us", "stamped_period_us", "stamped_hip_event_step_us", "by_kind")}))
