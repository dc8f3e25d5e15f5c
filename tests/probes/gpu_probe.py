"""Scratch GPU experiments (not a test): decode-time stability across engines in one process."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import valle_amd  # noqa
from valle_amd.engine import load_probe_library
load_probe_library()  # libvallex_probes.so: `python vall-e_amd/csrc/build.py --probes`
from conftest import Golden
from valle_amd.models import VALLE

g = Golden("cfg0_topk10")
sd = g.state_dict()

def mk(precision, **kw):
    c = g.cfg
    m = VALLE(c.decoder_dim, c.nhead, c.num_decoder_layers, prefix_mode=c.prefix_mode, precision=precision, max_text=128,
              max_audio=1280, print_eos=False, **kw)
    m.load_state_dict(sd)
    return m.to("cuda:0").eval()

def run(m, tag, n=3):
    for i in range(n):
        m.inference(g.x.cuda(), g.x_lens.cuda(), g.y.cuda(), None, top_k=10, exp_noise=g.exp_noise)
        t = m.engine().timings()
        print(json.dumps(dict(tag=tag, it=i, decode_ms=round(t["decode_ms"], 2), us_per_step=round(1e3 * t["decode_ms"] / t["launches"], 1),
                              prefill_ms=round(t["prefill_ms"], 3), nar_ms=round(t["nar_ms"], 3))), flush=True)

from valle_amd.engine import launch_floor
for grid, block in ((256, 256), (1, 64), (128, 256), (1024, 256)):
    print(json.dumps(dict(floor=launch_floor(62, grid, block, 200), grid=grid, block=block)), flush=True)
seq = sys.argv[1:] or ["fp32", "bf16m", "fp32ng"]
for i, s in enumerate(seq):
    if s == "fp32": m = mk("fp32")
    elif s == "fp32ng": m = mk("fp32", no_graph=True)
    elif s == "bf16s": m = mk("bf16", simple_rows=True)
    elif s == "bf16sng": m = mk("bf16", simple_rows=True, no_graph=True)
    elif s == "bf16m": m = mk("bf16")
    run(m, f"{i}:{s}")
    m._drop_engine()
