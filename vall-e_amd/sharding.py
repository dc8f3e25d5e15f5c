"""Utterance sharding across the GPUs of one node.

The hot path is batch-1 and utterances share no state (valle.py:989), so N GPUs run N model
replicas.  The only exchange is at the edges: rank 0 scatters the padded id tensors
(KB-scale) and gathers the code matrices back — ``torch.distributed`` scatter / gather, i.e.
RCCL over xGMI with backend "nccl", gloo on CPU in the tests.  No collective touches the data
path in between.

Partition (``plan_partition``): a decode costs about 16 x S steps (the reference's stop rule,
valle.py:1047), so rank 0 sorts the utterances by text length, cuts the sorted list into contiguous
groups of ``group`` utterances (one padded batch each: neighbours in length, so the batch's step count
is set by utterances of similar length) and deals the groups to the ranks in snake order
(0..W-1, W-1..0, ...), which evens out the summed cost per rank.  The gather restores utterance order.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

Utt = Tuple[torch.Tensor, torch.Tensor, torch.Tensor]  # x (1,S) int64, x_lens (1,) int32, y (1,P,Q) int64


def plan_partition(lengths: Sequence[int], world: int, group: int = 1) -> List[List[int]]:
    """-> per rank, the utterance indices it decodes, ``group`` consecutive entries forming one batch.
    ``len(lengths)`` must be a multiple of ``world * group`` (every rank gets the same number of groups)."""
    n = len(lengths)
    if n % (world * group):
        raise ValueError(f"{n} utterances do not split into {world} ranks x groups of {group}")
    order = sorted(range(n), key=lambda i: (-int(lengths[i]), i))
    groups = [order[k : k + group] for k in range(0, n, group)]
    per_rank: List[List[int]] = [[] for _ in range(world)]
    for gi, g in enumerate(groups):
        r = gi % (2 * world)
        per_rank[r if r < world else 2 * world - 1 - r] += g
    return per_rank


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


def scatter_lists(per_rank_utts: Optional[Sequence[Sequence[Utt]]], per_rank: int, device, world: int, rank: int) -> List[Utt]:
    """rank 0 passes one list of ``per_rank`` utterances for every rank; every rank returns its own on ``device``."""
    if world == 1:
        assert per_rank_utts is not None and len(per_rank_utts) == 1 and len(per_rank_utts[0]) == per_rank
        return [(x.to(device), xl.to(device), y.to(device)) for x, xl, y in per_rank_utts[0]]
    hdr = torch.zeros(3, dtype=torch.int64, device=device)
    if rank == 0:
        assert per_rank_utts is not None and len(per_rank_utts) == world and all(len(l) == per_rank for l in per_rank_utts)
        flat = [u for l in per_rank_utts for u in l]
        hdr[0] = max(u[0].shape[1] for u in flat)
        hdr[1] = max(u[2].shape[1] for u in flat)
        hdr[2] = flat[0][2].shape[2]
    dist.broadcast(hdr, src=0)
    smax, pmax, q = (int(v) for v in hdr)
    text = torch.empty((per_rank, smax), dtype=torch.int64, device=device)
    prom = torch.empty((per_rank, pmax, q), dtype=torch.int64, device=device)
    lens = torch.empty((per_rank, 2), dtype=torch.int64, device=device)
    lists = [None, None, None]
    if rank == 0:
        packed = [_pack(l, smax, pmax, q) for l in per_rank_utts]
        lists = [[p[i].to(device) for p in packed] for i in range(3)]
    for buf, lst in zip((text, prom, lens), lists):
        dist.scatter(buf, scatter_list=lst, src=0)
    out = []
    for i in range(per_rank):
        s, p = int(lens[i, 0]), int(lens[i, 1])
        out.append((text[i : i + 1, :s].contiguous(), torch.tensor([s], dtype=torch.int32, device=device),
                    prom[i : i + 1, :p].contiguous()))
    return out


def gather_lists(codes: Sequence[torch.Tensor], device, world: int, rank: int) -> Optional[List[List[torch.Tensor]]]:
    """Every rank passes its ``(1, T_i, Q)`` code tensors (the same count on every rank); rank 0 gets one list per
    rank back, other ranks get None."""
    if world == 1:
        return [list(codes)]
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
    return [[got[r][s : s + 1, : int(got_l[r][s])] for s in range(n)] for r in range(world)]


def scatter_utterances(all_utts: Optional[Sequence[Utt]], per_rank: int, device, world: int, rank: int) -> List[Utt]:
    """Fixed dealing, no sort: rank 0 passes ``world * per_rank`` utterances ordered [step][rank] (utterance
    u = step * world + rank runs on ``rank``)."""
    lists = None
    if rank == 0:
        assert all_utts is not None and len(all_utts) == world * per_rank
        lists = [[all_utts[s * world + r] for s in range(per_rank)] for r in range(world)]
    return scatter_lists(lists, per_rank, device, world, rank)


def gather_codes(codes: Sequence[torch.Tensor], device, world: int, rank: int) -> Optional[List[torch.Tensor]]:
    """Inverse of ``scatter_utterances``: rank 0 gets all code tensors back in utterance order u = step * world + rank."""
    got = gather_lists(codes, device, world, rank)
    if got is None:
        return None
    return [got[r][s] for s in range(len(codes)) for r in range(world)]


def infer_sharded(run_one, all_utts: Optional[Sequence[Utt]], per_rank: int, device, world: int, rank: int,
                  run_many=None, group: int = 1):
    """scatter -> decode the local utterances -> gather; rank 0 returns the codes in the order of ``all_utts``.
    ``run_one(x, x_lens, y) -> (1,T,Q)`` decodes one utterance at a time (batch-1, the reference's mode);
    ``run_many(list_of_utts) -> list of (1,T,Q)`` decodes ``group`` of them as one padded batch
    (``VALLE.inference_batch``, BASELINE configs[3]: 256 utterances = 32 per GPU on 8 GPUs).  ``per_rank`` must be a
    multiple of ``group``.  The partition is ``plan_partition`` over the text lengths."""
    plan = None
    lists = None
    if rank == 0:
        assert all_utts is not None and len(all_utts) == world * per_rank
        plan = plan_partition([u[0].shape[1] for u in all_utts], world, group)
        lists = [[all_utts[i] for i in idx] for idx in plan]
    mine = scatter_lists(lists, per_rank, device, world, rank)
    if run_many is not None:
        outs = []
        for k in range(0, len(mine), group):
            outs += list(run_many(mine[k : k + group]))
    else:
        outs = [run_one(*u) for u in mine]
    got = gather_lists(outs, device, world, rank)
    if got is None:
        return None
    res: List[Optional[torch.Tensor]] = [None] * (world * per_rank)
    for r, idx in enumerate(plan):
        for j, i in enumerate(idx):
            res[i] = got[r][j]
    return res
