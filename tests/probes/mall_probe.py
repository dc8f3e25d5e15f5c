"""Does the batch-1 decode step run faster per layer when ALL its weights fit the 256 MB Infinity Cache?  Same geometry (d = 1024,
16 heads), 12 / 9 / 6 / 3 layers: weights 304 / 228 / 152 / 76 MB + the KV rows.  If the per-layer time (difference quotient)
drops once the working set fits, a weight prefetch into the memory-side cache has that much to gain; if not, HBM is not what
the launches wait for."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import valle_amd  # noqa
from valle_amd.config import ModelConfig
from valle_amd.models import VALLE
from valle_amd.weights import synthetic_inputs, synthetic_state_dict

x, xl, y = synthetic_inputs(47, 225, 8, seed=1)
x, xl, y = x.cuda(), xl.cuda(), y.cuda()
res = {}
for L in (12, 9, 6, 3):
    cfg = ModelConfig(decoder_dim=1024, nhead=16, num_decoder_layers=L, prefix_mode=1)
    m = VALLE(1024, 16, L, prefix_mode=1, precision="bf16", max_text=64, max_audio=1024, print_eos=False)
    m.load_state_dict(synthetic_state_dict(cfg, seed=0))
    m.to("cuda:0").eval()
    eng = m.engine()
    best = None
    for i in range(3):
        torch.manual_seed(1 + i)
        m.inference(x, xl, y, None, top_k=10)
        torch.cuda.synchronize()
        t = eng.timings()
        us = t["decode_ms"] * 1e3 / max(1, t["launches"])
        best = us if best is None else min(best, us)
    res[L] = best
    print(f"layers {L:2d}: AR step {best:7.2f} us  ({t['launches']} steps)", flush=True)
    m._drop_engine()
    del m, eng
Ls = sorted(res)
for a, b in zip(Ls[:-1], Ls[1:]):
    print(f"per layer between {a} and {b} layers: {(res[b] - res[a]) / (b - a):.2f} us", flush=True)
