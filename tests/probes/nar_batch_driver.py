"""Batched NAR pass alone (no AR decode), for rocprofv3 passes over its kernels: d = 1024, 16 heads, L layers (default 2: the kernels
of a layer-stage do not depend on the layer count), B utterances of S text / 225 prompt / 16 S + 1 generated rows, random tokens.
usage: python3 tests/probes/nar_batch_driver.py [B=32] [S=47] [L=2] [calls=2] [precision=bf16]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import valle_amd  # noqa
from valle_amd.config import ModelConfig
from valle_amd.models import VALLE
from valle_amd.weights import synthetic_inputs, synthetic_state_dict

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
S = int(sys.argv[2]) if len(sys.argv) > 2 else 47
L = int(sys.argv[3]) if len(sys.argv) > 3 else 2
calls = int(sys.argv[4]) if len(sys.argv) > 4 else 2
prec = sys.argv[5] if len(sys.argv) > 5 else "bf16"
T = 16 * S + 1
cfg = ModelConfig(decoder_dim=1024, nhead=16, num_decoder_layers=L, prefix_mode=1)
m = VALLE(1024, 16, L, prefix_mode=1, precision=prec, max_text=max(64, S), max_audio=((225 + T + 1 + 63) // 64) * 64, print_eos=False, max_batch=B)
m.load_state_dict(synthetic_state_dict(cfg, seed=0))
m.to("cuda:0").eval()
eng = m.engine()
g = torch.Generator().manual_seed(0)
texts, prompts, tokens = [], [], []
for b in range(B):
    x, xl, y = synthetic_inputs(S, 225, 8, seed=1 + b)
    texts.append(x[0].cuda()); prompts.append(y[0].cuda()); tokens.append(torch.randint(0, 1024, (T,), generator=g).cuda())
for _ in range(calls):
    out = eng.nar_batch(texts, prompts, tokens)
    torch.cuda.synchronize()
    print("nar_batch", B, "x", S + 225 + T, "rows:", round(eng.timings()["nar_ms"], 2), "ms", flush=True)

# probe build (libvallex_stamps.so copied over libvallex.so): phases of iteration 3 of the attention kernel's workgroup 0, last launch
import ctypes as C
from valle_amd.engine import load_library
lib = load_library()
if hasattr(lib, "vx_debug_read_stamps"):
    st = (C.c_uint64 * 32)()
    lib.vx_debug_read_stamps.argtypes = [C.POINTER(C.c_uint64), C.c_int32]
    if lib.vx_debug_read_stamps(st, 32) == 0:
        t = [int(v) for v in st]
        us = lambda a, b: round((t[b] - t[a]) * 0.01, 2)
        print("attention wg 0: entry->first tile", us(8, 9), "us; iterations (start to start)", [us(10 + i, 11 + i) for i in range(11)])
        print("iteration 3: S^T + max", us(24, 25), "| exp + PV", us(25, 26), "| loads landed + stored", us(26, 27), "| barrier", us(27, 28), "us; loop", us(9, 22), "kernel", us(8, 23))
