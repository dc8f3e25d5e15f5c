"""CPU: the N>1 utterance-sharding path with 2 gloo ranks — scatter, per-rank decode, gather must
return exactly what a single rank produces, in utterance order, for ragged S / P / T."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _fake_decode(x, x_lens, y):
    """Deterministic stand-in for model.inference with data-dependent output length."""
    T = int(x.sum()) % 7 + 3
    base = (y[0, :, :].sum(0) % 1024).reshape(1, 1, -1)
    return (base + torch.arange(T).reshape(1, T, 1) * int(x_lens[0])) % 1024


def _utts(n):
    g = torch.Generator().manual_seed(0)
    out = []
    for i in range(n):
        S, P = 4 + (i * 3) % 5, 6 + (i * 5) % 7
        out.append((torch.randint(3, 100, (1, S), generator=g), torch.tensor([S], dtype=torch.int32),
                    torch.randint(0, 1024, (1, P, 8), generator=g)))
    return out


def _worker(rank, world, port, per_rank, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from valle_amd.sharding import infer_sharded

    utts = _utts(world * per_rank) if rank == 0 else None
    res = infer_sharded(_fake_decode, utts, per_rank, torch.device("cpu"), world, rank)
    if rank == 0:
        q.put([r.clone() for r in res])
    else:
        assert res is None
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_matches_single_rank():
    world, per_rank = 2, 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    import socket

    with socket.socket() as sk:  # a free port, so concurrent test runs cannot collide
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = [ctx.Process(target=_worker, args=(r, world, port, per_rank, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = [_fake_decode(*u) for u in _utts(world * per_rank)]
    assert len(got) == len(want)
    for a, b in zip(got, want):
        assert torch.equal(a, b)


def test_single_rank_passthrough():
    from valle_amd.sharding import infer_sharded

    utts = _utts(3)
    res = infer_sharded(_fake_decode, utts, 3, torch.device("cpu"), 1, 0)
    for a, u in zip(res, utts):
        assert torch.equal(a, _fake_decode(*u))
