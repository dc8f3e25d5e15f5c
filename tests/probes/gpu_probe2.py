"""Scratch: per-layer AR step time vs model size (does Infinity-Cache residency of the weights help?)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
ge.build()
from valle_amd.config import ModelConfig
from valle_amd.models import VALLE
from valle_amd.weights import synthetic_inputs, synthetic_state_dict

for L in (3, 6, 9, 12):
    cfg = ModelConfig(decoder_dim=1024, nhead=16, num_decoder_layers=L, prefix_mode=1, num_quantizers=1)
    sd = synthetic_state_dict(cfg, 0)
    for nop in (True,):
        m = VALLE(1024, 16, L, prefix_mode=1, num_quantizers=1, precision="bf16", max_text=64, max_audio=1024, print_eos=False, no_prefetch=nop)
        m.load_state_dict(sd); m.to("cuda:0").eval()
        x, xl, y = synthetic_inputs(47, 225)
        for it in range(3):
            torch.manual_seed(1)
            m.inference(x.cuda(), xl.cuda(), y.cuda(), None, top_k=10)
            t = m.engine().timings()
        us = 1e3 * t["decode_ms"] / t["launches"]
        print(json.dumps(dict(L=L, no_prefetch=nop, us_per_step=round(us, 1), us_per_layer=round((us) / L, 2), weights_MB=round(L*12*1024*1024*2/1e6))), flush=True)
        m._drop_engine()
