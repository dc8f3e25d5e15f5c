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
    res_b = infer_sharded(None, utts, per_rank, torch.device("cpu"), world, rank, group=3,
                          run_many=lambda us: [_fake_decode(*u) for u in us])  # one padded batch of 3 per rank
    if rank == 0:
        assert all(torch.equal(a, b) for a, b in zip(res, res_b))
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


def test_partition_sorts_by_length_and_balances():
    """sort by S, contiguous groups (tight batches), summed cost evened out over the ranks."""
    from valle_amd.sharding import plan_partition

    lens = [40 + (i * 7) % 15 for i in range(64)]
    plan = plan_partition(lens, world=4, group=4)
    assert sorted(i for idx in plan for i in idx) == list(range(64))  # a permutation
    assert all(len(idx) == 16 for idx in plan)
    for idx in plan:
        for k in range(0, 16, 4):  # every group is a contiguous run of the sorted order: spread <= what 4 neighbours span
            g = [lens[i] for i in idx[k : k + 4]]
            assert max(g) - min(g) <= 1
    cost = [sum(lens[i] for i in idx) for idx in plan]
    assert max(cost) - min(cost) <= 4
    import pytest

    with pytest.raises(ValueError):
        plan_partition(lens[:10], world=4, group=4)


def test_bench_launcher_spawns_ranks_on_gloo():
    """`python bench.py --gpus 2` (no WORLD_SIZE in the environment) must start two ranks itself and print ONE line with
    n_gpus == 2; --dry-run-cpu swaps the engine for a stand-in so that this runs without a GPU (gloo).  The 2 x 4 x 3
    utterances go through plan_partition / scatter / gather exactly as on the GPUs."""
    import json
    import subprocess

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "4",
                        "--dry-run-cpu"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 2 and rec["scaling"] == "weak" and rec["data"].startswith("dry-run")
    frames = sum(16 * (40 + (i % 15)) + 1 for i in range(8, 24))  # the 16 utterances of the two timed steps
    assert abs(rec["value"] * rec["ms_per_step"] * 2e-3 - frames) < 0.01 * frames
    # a rank count that disagrees with --gpus is refused
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run-cpu"], env=dict(env, WORLD_SIZE="1", RANK="0"),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0


def test_bench_batch32_two_ranks_returns_codes_in_utterance_order(tmp_path):
    """BASELINE configs[3]'s code path at world size 2 on gloo: `bench.py --gpus 2 --batch 32` (ragged S in [40, 54]) with the
    stand-in decoder - plan_partition (sort by S, contiguous groups of 32, snake deal) -> scatter -> inference_batch per rank ->
    gather.  Rank 0 must end up with every utterance's codes at the utterance's own index: length 16 S + 1 and the stand-in's
    checksum of THAT utterance's inputs (a permutation error of the plan / scatter / gather would move them)."""
    import json
    import subprocess

    sys.path.insert(0, ROOT)
    import bench
    from valle_amd.weights import synthetic_inputs

    dump = tmp_path / "codes.json"
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["VX_BENCH_DUMP"] = str(dump)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--batch", "32",
                        "--dry-run-cpu"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    rec = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    got = json.load(open(dump))
    assert rec["n_gpus"] == 2 and len(got) == 64
    model = bench._DryRunModel()
    total = 0
    for i, (T, chk) in enumerate(got):
        S = bench.S_TEXT - 7 + (i % 15)
        x, xl, y = synthetic_inputs(S, bench.P_PROMPT, 8, seed=1 + i)
        want = model.inference(x, xl, y, None)
        assert T == 16 * S + 1 == want.shape[1] and chk == int(want.sum()), i
        total += T
    assert abs(rec["value"] * rec["ms_per_step"] * 1e-3 - total) < 0.01 * total
