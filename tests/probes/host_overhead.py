"""Where does one VALLE.inference() call spend host wall time beyond the device-timed phases?  (bench workload, batch 1)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import valle_amd  # noqa
from valle_amd.config import ModelConfig
from valle_amd.models import VALLE
from valle_amd.weights import synthetic_inputs, synthetic_state_dict

cfg = ModelConfig(decoder_dim=1024, nhead=16, num_decoder_layers=12, prefix_mode=1)
m = VALLE(1024, 16, 12, prefix_mode=1, precision="bf16", max_text=64, max_audio=1024, print_eos=False)
m.load_state_dict(synthetic_state_dict(cfg, seed=0))
m.to("cuda:0").eval()
x, xl, y = synthetic_inputs(47, 225, 8, seed=1)
x, xl, y = x.cuda(), xl.cuda(), y.cuda()
eng = m.engine()
for i in range(3):
    torch.manual_seed(1 + i)
    m.inference(x, xl, y, None, top_k=10)
torch.cuda.synchronize()
for i in range(3):
    torch.manual_seed(10 + i)
    t0 = time.perf_counter()
    m.inference(x, xl, y, None, top_k=10)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    t = eng.timings()
    print("wall %.2f ms | device: prefill %.2f decode %.2f nar %.2f | launches %d" % ((t1 - t0) * 1e3, t["prefill_ms"], t["decode_ms"], t["nar_ms"], t["launches"]), flush=True)
# phase by phase
text, prompts = x[0], y[0, :, :8].contiguous()
for i in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eng.ar_prefill(text, prompts[:, 0].contiguous()); torch.cuda.synchronize(); t1 = time.perf_counter()
    eng.ar_decode(top_k=10, temperature=1.0, exp_noise=None, seed=5, max_new_tokens=-1); torch.cuda.synchronize(); t2 = time.perf_counter()
    toks, reason, n_pass = eng.ar_result(); t3 = time.perf_counter()
    codes = eng.nar(text, prompts, toks); torch.cuda.synchronize(); t4 = time.perf_counter()
    t = eng.timings()
    print("prefill %.2f (dev %.2f) | decode %.2f (dev %.2f) | result %.2f | nar %.2f (dev %.2f)" % ((t1 - t0) * 1e3, t["prefill_ms"], (t2 - t1) * 1e3, t["decode_ms"], (t3 - t2) * 1e3, (t4 - t3) * 1e3, t["nar_ms"]), flush=True)
