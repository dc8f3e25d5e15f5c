import sys, os, torch, faulthandler
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from test_gpu_batch import _setup, _utts
cfg, sd, m = _setup(max_batch=4)
u = _utts([(6, 30), (9, 12), (4, 55), (5, 8)])
eng = m.engine()
print("prefill_all", flush=True)
eng.batch_prefill_all([x[0][0] for x in u[:3]], [x[2][0, :, 0].contiguous() for x in u[:3]])
print("decode", flush=True)
eng.batch_decode(3, top_k=5, seeds=[11, 22, 33])
print("decoded", [eng.batch_result(b)[0].numel() for b in range(3)], flush=True)
a = m.inference_batch([u[0], u[1], u[2]], top_k=5, seeds=[11, 22, 33])
print("ok", [t.shape for t in a], flush=True)
