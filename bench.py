#!/usr/bin/env python3
"""Headline benchmark: BASELINE.json configs[1] — d=1024 nhead=16 L=12 bf16, batch-1 AR top-k(10)
decode + 7 NAR stages of a 10 s utterance (S=47 phonemes, 3 s prompt -> 753 frames x 8 codebooks).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one whole VALLE.inference() (prefill + 753 AR passes + 7 NAR stages) of one utterance
per GPU.  Utterances are independent, so N GPUs run N replicas (weak scaling): rank 0 owns the
inputs, scatters them over RCCL, every rank decodes its own, codes are gathered back; scatter and
gather are inside the timed region.  Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

S_TEXT, P_PROMPT, TOP_K, TEMP = 47, 225, 10, 1.0
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def ar_bytes_per_token(d: int, L: int, ctx: float, bpe: int = 2) -> float:
    """weights streamed once per token + KV cache read + KV write."""
    weights = (L * (12 * d * d + 13 * d) + 2 * d + 1025 * d) * bpe
    return weights + 2 * L * d * bpe * (ctx + 1)


def measured_traffic(ctx_mean: float):
    """HBM bytes per AR step from the committed PMC profile (profiles/r01_pmc_ar_step.json: FETCH_SIZE x2 as the
    microarch guide prescribes for gfx950, split into the ctx-independent GEMV part and the per-cached-row
    attention part so that it can be quoted at this run's mean context).  None if the profile is absent."""
    off = os.environ.get("VX_AR_PREFETCH", "1") == "0"  # the engine's weight warm-up doubles the fabric requests (see the profile's "reading")
    path = os.path.join(ROOT, "profiles", "r01_pmc_ar_step_prefetch_off.json" if off else "r01_pmc_ar_step.json")
    if not os.path.isfile(path):
        return None
    p = json.load(open(path))
    return int(p["gemv_bytes_per_step_corrected"] + p["attn_bytes_per_ctx_row_corrected"] * ctx_mean)


def cpu_baseline(sd, cfg, x, x_lens, y, n_tokens: int):
    """The reference algorithm (no KV cache, fp32, oracle/valle_oracle.inference_faithful — a checked
    port, kind="port") on a bounded sample of the same workload, timed on this box's host cores."""
    from oracle import valle_oracle as vo  # checker / baseline only — never the product path

    m = vo.OracleModel(sd, cfg.decoder_dim, cfg.nhead, cfg.num_decoder_layers, cfg.prefix_mode, cfg.prepend_bos,
                       cfg.num_quantizers)
    threads = torch.get_num_threads()
    t0 = time.time()
    codes = vo.inference_faithful(m, x, x_lens, y, None, TOP_K, TEMP, exp_noise=torch.ones(n_tokens + 2, 1025),
                                  max_new_tokens=n_tokens)
    dt = time.time() - t0
    return {
        "value": round(codes.shape[1] / dt, 3), "unit": "codec-tokens/s", "cores": threads, "kind": "port",
        "sample": f"first {n_tokens} of 753 AR steps (no KV cache, ctx {S_TEXT + P_PROMPT}..{S_TEXT + P_PROMPT + n_tokens}) + "
                  f"7 NAR stages over {S_TEXT + P_PROMPT + n_tokens} rows, fp32, {dt:.1f} s wall; later AR steps cost more "
                  "(O(T^2)), so the full-length CPU rate is lower than this",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-tokens", type=int, default=16)
    ap.add_argument("--batch", type=int, default=1, help="utterances decoded together per GPU per step (1 = the headline "
                    "batch-1 workload, BASELINE configs[1]; 32 = configs[2]/[3])")
    ap.add_argument("--no-graph", action="store_true", help="launch the AR step kernel by kernel (rocprofv3 --pmc cannot follow hipGraph replays)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)  # RCCL

    import __graft_entry__ as ge

    if rank == 0:
        ge.build()
    if world > 1:
        dist.barrier()
    from valle_amd.config import ModelConfig
    from valle_amd.models import VALLE
    from valle_amd.sharding import gather_codes, scatter_utterances
    from valle_amd.weights import synthetic_inputs, synthetic_state_dict

    cfg = ModelConfig(decoder_dim=1024, nhead=16, num_decoder_layers=12, prefix_mode=1)
    sd = synthetic_state_dict(cfg, seed=0)
    model = VALLE(1024, 16, 12, prefix_mode=1, precision=args.precision, max_text=64, max_audio=1024, print_eos=False,
                  no_graph=args.no_graph, max_batch=args.batch if args.batch > 1 else 0)
    model.load_state_dict(sd)
    model.to(dev).eval()
    eng = model.engine()

    n_total = args.warmup + args.steps
    # rank 0 owns every utterance of the job: (world * n_total) independent inputs
    utts = None
    if rank == 0:
        utts = [synthetic_inputs(S_TEXT, P_PROMPT, 8, seed=1 + i) for i in range(world * n_total * args.batch)]

    def run_phase(step_ids):
        """scatter -> decode -> gather for the utterances of these steps; returns #frames produced here."""
        Bt = args.batch
        mine = scatter_utterances([utts[(s * world + r) * Bt + j] for s in step_ids for j in range(Bt) for r in range(world)]
                                  if rank == 0 else None, len(step_ids) * Bt, dev, world, rank)
        outs, frames = [], 0
        tms = []
        if Bt > 1:
            for i in range(len(step_ids)):
                group = mine[i * Bt : (i + 1) * Bt]
                res = model.inference_batch(group, top_k=TOP_K, temperature=TEMP,
                                            seeds=[1234 + (step_ids[i] * world + rank) * Bt + j for j in range(Bt)])
                outs += res
                frames += sum(c.shape[1] for c in res)
                tms.append(eng.timings())
        else:
            for i, (x, x_lens, y) in enumerate(mine):
                torch.manual_seed(1234 + step_ids[i] * world + rank)  # seeds the on-device sampler
                codes = model.inference(x, x_lens, y, None, top_k=TOP_K, temperature=TEMP)
                outs.append(codes)
                frames += codes.shape[1]
                tms.append(eng.timings())
        gather_codes(outs, dev, world, rank)
        return frames, outs, tms

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    run_phase(list(range(args.warmup)))
    fence()
    t0 = time.perf_counter()
    frames, outs, tms = run_phase(list(range(args.warmup, n_total)))
    fence()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt, float(frames)], dtype=torch.float64, device=dev)
    if world > 1:
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt, frames = float(tmax[0]), float(tsum[1])
    # HIP-event device times (engine stream) of every timed step on this rank: totals for the AR step, p50 for NAR
    import statistics
    tm = dict(tms[-1])
    for k in ("decode_ms", "launches", "batch_decode_ms", "batch_launches"):
        tm[k] = sum(t[k] for t in tms)
    tm["nar_ms"] = statistics.median(t["nar_ms"] for t in tms)
    tm["prefill_ms"] = statistics.median(t["prefill_ms"] for t in tms)

    if rank == 0:
        T = outs[-1].shape[1]
        ctx_mean = S_TEXT + P_PROMPT + (T - 1) / 2.0
        bpe = 2 if args.precision == "bf16" else 4
        Bt = args.batch
        if Bt > 1:  # batched step: weights once + Bt KV streams
            step_s = tm["batch_decode_ms"] * 1e-3 / max(1, tm["batch_launches"])
            step_bytes = ar_bytes_per_token(1024, 12, 0, bpe) + Bt * 2 * 12 * 1024 * bpe * ctx_mean
            tm = dict(tm, decode_ms=tm["batch_decode_ms"], launches=tm["batch_launches"] * Bt)
        else:
            step_s = tm["decode_ms"] * 1e-3 / max(1, tm["launches"])
            step_bytes = ar_bytes_per_token(1024, 12, ctx_mean, bpe)
        achieved = step_bytes / step_s / 1e9
        out = {
            "metric": "AR codec-tokens/sec/GPU + NAR 7-stage p50 latency, d=1024 L=12 10s utterance",
            "value": round(frames / dt, 2), "unit": "codec-tokens/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt * 1e3 / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": (f"BASELINE configs[1]: d=1024 nhead=16 L=12, batch=1 AR top-k(10) + 7 NAR stages, "
                                    f"S={S_TEXT} P={P_PROMPT} -> T={T} frames x 8 codebooks, one utterance per GPU per step")
                       if Bt == 1 else
                       (f"BASELINE configs[2]{'' if Bt == 32 else ' geometry at another batch size'}: d=1024 nhead=16 L=12, batch={Bt} concurrent utterances per GPU (padded KV, "
                        f"hipGraph step, one batched prefill and one batched NAR pass), S={S_TEXT} P={P_PROMPT} -> T={T} x 8"),
                       "parallelism": f"replica x{world} (utterance sharding, RCCL scatter/gather)"},
            "ar_tokens_per_s": round(tm["launches"] / (tm["decode_ms"] * 1e-3), 1),
            "ar_step_us": round(step_s * 1e6, 2),
            "prefill_ms": round(tm["prefill_ms"], 3),
            "nar_7stage_ms": round(tm["nar_ms"], 3),  # p50 over the timed steps
            "roofline": {"bound": "hbm", "kernel": "AR decode step (hipGraph of 62 kernels = 1 token)" if Bt == 1 else
                         f"batched AR decode step (hipGraph of 87 kernels = {Bt} tokens)",
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": measured_traffic(ctx_mean) if (args.precision == "bf16" and Bt == 1) else None,
                         "bytes_per_launch": int(step_bytes)},
        }
        if Bt == 1 and os.environ.get("VX_AR_PREFETCH", "1") != "0":
            out["roofline"]["traffic_note"] = ("fabric read requests (FETCH_SIZE x2), ~2x algorithmic by design: each GEMV also requests the weights of the GEMV "
                                               "two places ahead so that they are served from the Infinity Cache; VX_AR_PREFETCH=0: 1.05x, -8% tokens/s")
        if world == 1 and not args.no_cpu_baseline:
            x, x_lens, y = utts[0]
            out["cpu_baseline"] = cpu_baseline(sd, cfg, x, x_lens, y, args.cpu_tokens)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
