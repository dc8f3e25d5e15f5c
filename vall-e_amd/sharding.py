"""Utterance sharding across the GPUs of one node.

The hot path is batch-1 and utterances share no state (valle.py:989), so N GPUs run N model
replicas.  The only exchange is at the edges: rank 0 scatters the padded id tensors
(KB-scale) and gathers the code matrices back — ``torch.distributed`` scatter / gather, i.e.
RCCL over xGMI with backend "nccl", gloo on CPU in the tests.  No collective touches the data
path in between.  Utterance u = step * world + rank runs on ``rank``.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

Utt = Tuple[torch.Tensor, torch.Tensor, torch.Tensor]  # x (1,S) int64, x_lens (1,) int32, y (1,P,Q) int64


def _pack(utts: Sequence[Utt], smax: int, pmax: int, q: int):
    n = len(utts)
    text = torch.zeros((n, smax), dtype=torch.int64)
    prom = torch.zeros((n, pmax, q), dtype=torch.int64)
    lens = torch.zeros((n, 2), dtype=torch.int64)
    for i, (x, x_lens, y) in enumerate(utts):
        s, p = x.shape[1], y.shape[1]
        text[i, :s] = x[0]
        prom[i, :p] = y[0]
        lens[i, 0], lens[i, 1] = s, p
    return text, prom, lens


def scatter_utterances(all_utts: Optional[Sequence[Utt]], per_rank: int, device, world: int, rank: int) -> List[Utt]:
    """rank 0 passes ``world * per_rank`` utterances ordered [step][rank]; every rank returns its own
    ``per_rank`` utterances on ``device``."""
    if world == 1:
        assert all_utts is not None and len(all_utts) == per_rank
        return [(x.to(device), xl.to(device), y.to(device)) for x, xl, y in all_utts]
    hdr = torch.zeros(3, dtype=torch.int64, device=device)
    if rank == 0:
        assert all_utts is not None and len(all_utts) == world * per_rank
        hdr[0] = max(u[0].shape[1] for u in all_utts)
        hdr[1] = max(u[2].shape[1] for u in all_utts)
        hdr[2] = all_utts[0][2].shape[2]
    dist.broadcast(hdr, src=0)
    smax, pmax, q = (int(v) for v in hdr)
    text = torch.empty((per_rank, smax), dtype=torch.int64, device=device)
    prom = torch.empty((per_rank, pmax, q), dtype=torch.int64, device=device)
    lens = torch.empty((per_rank, 2), dtype=torch.int64, device=device)
    lists = [None, None, None]
    if rank == 0:
        packed = [_pack([all_utts[s * world + r] for s in range(per_rank)], smax, pmax, q) for r in range(world)]
        lists = [[p[i].to(device) for p in packed] for i in range(3)]
    for buf, lst in zip((text, prom, lens), lists):
        dist.scatter(buf, scatter_list=lst, src=0)
    out = []
    for i in range(per_rank):
        s, p = int(lens[i, 0]), int(lens[i, 1])
        out.append((text[i : i + 1, :s].contiguous(), torch.tensor([s], dtype=torch.int32, device=device),
                    prom[i : i + 1, :p].contiguous()))
    return out


def gather_codes(codes: Sequence[torch.Tensor], device, world: int, rank: int) -> Optional[List[torch.Tensor]]:
    """Every rank passes its ``(1, T_i, Q)`` code tensors; rank 0 gets all of them back in utterance
    order (u = step * world + rank), other ranks get None."""
    if world == 1:
        return list(codes)
    n = len(codes)
    q = codes[0].shape[2] if n else 0
    meta = torch.tensor([max([c.shape[1] for c in codes], default=0), q], dtype=torch.int64, device=device)
    dist.all_reduce(meta, op=dist.ReduceOp.MAX)
    tmax, q = int(meta[0]), int(meta[1])
    buf = torch.full((n, tmax, q), -1, dtype=torch.int64, device=device)
    lens = torch.zeros(n, dtype=torch.int64, device=device)
    for i, c in enumerate(codes):
        buf[i, : c.shape[1]] = c[0].to(device)
        lens[i] = c.shape[1]
    got = [torch.empty_like(buf) for _ in range(world)] if rank == 0 else None
    got_l = [torch.empty_like(lens) for _ in range(world)] if rank == 0 else None
    dist.gather(buf, gather_list=got, dst=0)
    dist.gather(lens, gather_list=got_l, dst=0)
    if rank != 0:
        return None
    out = []
    for s in range(n):
        for r in range(world):
            out.append(got[r][s : s + 1, : int(got_l[r][s])])
    return out


def infer_sharded(run_one, all_utts: Optional[Sequence[Utt]], per_rank: int, device, world: int, rank: int,
                  run_many=None):
    """scatter -> decode the local utterances -> gather.  ``run_one(x, x_lens, y) -> (1,T,Q)`` decodes them one
    at a time (batch-1, the reference's mode); ``run_many(list_of_utts) -> list of (1,T,Q)`` decodes them as one
    padded batch (``VALLE.inference_batch``, BASELINE configs[3]: 256 utterances = 32 per GPU on 8 GPUs)."""
    mine = scatter_utterances(all_utts, per_rank, device, world, rank)
    outs = run_many(mine) if run_many is not None else [run_one(*u) for u in mine]
    return gather_codes(outs, device, world, rank)
