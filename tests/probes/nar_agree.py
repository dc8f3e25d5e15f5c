import sys, os, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from conftest import Golden
from test_gpu_batch import _setup, _utts
g = Golden("cfg0_topk10")
cfg, sd, m = _setup(max_batch=4)
eng = m.engine()
u = _utts([(6, 30), (9, 12), (4, 55)])
toks = [torch.randint(0, 1024, (16 * x.shape[1] + 1,), generator=torch.Generator().manual_seed(i)) for i, (x, _, _) in enumerate(u)]
texts = [x[0] for x, _, _ in u] + [g.x[0]]
proms = [y[0].contiguous() for _, _, y in u] + [g.y[0].contiguous()]
tks = toks + [g.codes[0, :, 0].contiguous()]
single = [eng.nar(t, p, k).cpu() for t, p, k in zip(texts, proms, tks)]
batched = [c.cpu() for c in eng.nar_batch(texts, proms, tks)]
for a, b in zip(single, batched):
    print("agree", [round((a[:, q] == b[:, q]).float().mean().item(), 3) for q in range(8)])
print("vs ref batched", [(batched[3][:, q] == g.codes[0][:, q]).float().mean().item() for q in range(1, 8)])
print("vs ref single ", [(single[3][:, q] == g.codes[0][:, q]).float().mean().item() for q in range(1, 8)])
