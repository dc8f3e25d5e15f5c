"""Per-stage NAR logit statistics of the UNMODIFIED reference for selected fixtures -> tests/golden/narstats/<name>.npz.

Build-container only (imports /root/reference through oracle/ref_harness.py).
Usage:  PYTHONDONTWRITEBYTECODE=1 python -m oracle.gen_nar_stats [fixture ...]   (default: the list below)

The committed fixtures keep the reference's codes and a few logits rows.  The bf16 / fp8 engines cannot be compared
free-running (one flipped argmax changes every later stage's input), so the GPU tests teacher-force every NAR stage with the
reference's own codes of the earlier stages (what the reference itself fed that stage, valle.py:1133-1134) and apply the
north-star rule per row: the argmax must equal the reference's wherever the reference's top-2 margin exceeds twice the
stated tolerance.  For that rule this script records, from forward hooks on the reference's own nar_predict_layers
(valle.py:1128), for EVERY generated row of EVERY stage: the top-1 / top-2 logit values and the row's largest magnitude,
plus the full 1024 logits of 16 evenly spaced rows per stage (checked against the engine's within the tolerance).
It re-runs the fixture's case through the reference and asserts that the codes equal the committed fixture first.
"""
from __future__ import annotations

import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

import valle_amd  # noqa: E402,F401
from valle_amd.config import ModelConfig  # noqa: E402
from valle_amd.weights import synthetic_inputs, synthetic_state_dict  # noqa: E402
from oracle.ref_harness import build_reference_model  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "narstats")
DEFAULT = ["tiny_mode0", "cfg0_topk10", "cfg1_topk10"]
N_ROWS = 16


def run(name: str):
    from oracle.gen_golden import CASES
    kw, S, P, top_k, temp, sseed, enroll, _ = CASES[name]
    cfg = ModelConfig(**kw)
    sd = synthetic_state_dict(cfg, seed=0)
    x, x_lens, y = synthetic_inputs(S, P, 8, seed=1)
    enroll_x_lens = None if enroll is None else torch.tensor([enroll], dtype=torch.int32)
    ref = build_reference_model(cfg, sd)
    logs = {}
    for si, layer in enumerate(ref.nar_predict_layers):
        layer.register_forward_hook(lambda m, i, o, si=si: logs.__setitem__(si, o.detach()[0].clone()))
    if sseed is not None:
        torch.manual_seed(sseed)
    t0 = time.time()
    with torch.no_grad():
        codes = ref.inference(x, x_lens, y, enroll_x_lens=enroll_x_lens, top_k=top_k, temperature=temp)
    print(f"[{name}] reference {tuple(codes.shape)} in {time.time() - t0:.1f}s", flush=True)
    fx = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    assert np.array_equal(fx["codes"].astype(np.int64), codes.numpy()), "reference run differs from the committed fixture"
    write_stats(name, codes, torch.stack([logs[i] for i in sorted(logs)]))


def write_stats(name: str, codes: torch.Tensor, st: torch.Tensor):
    """st: (Q-1, T, 1024) logits of the reference's own nar_predict_layers."""
    T = codes.shape[1]
    assert torch.equal(st.argmax(-1), codes[0, :, 1:].t())  # the reference's own argmax (valle.py:1130)
    top2 = st.topk(2, dim=-1)[0]
    rows = np.unique(np.linspace(0, T - 1, N_ROWS).round().astype(np.int64))
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, name + ".npz"),
                        top1=top2[..., 0].numpy(), top2=top2[..., 1].numpy(), absmax=st.abs().amax(-1).numpy(),
                        rows=rows.astype(np.int32), row_logits=st[:, rows].numpy())
    print(f"[{name}] wrote narstats ({os.path.getsize(os.path.join(OUT, name + '.npz')) / 1024:.0f} KiB)", flush=True)


if __name__ == "__main__":
    torch.set_num_threads(int(os.environ.get("GOLDEN_THREADS", "8")))
    for n in (sys.argv[1:] or DEFAULT):
        run(n)
