#!/usr/bin/env python3
"""Headline benchmark: BASELINE.json configs[1] — d=1024 nhead=16 L=12 bf16, batch-1 AR top-k(10)
decode + 7 NAR stages of a 10 s utterance (S=47 phonemes, 3 s prompt -> 753 frames x 8 codebooks).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]

``--gpus N`` with N > 1 starts N ranks itself (one fresh process per GPU, RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_* in their environment, before this process has touched HIP) and relays rank 0's line; under
``python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`` the ranks already exist and
this file runs as one of them.  Either way WORLD_SIZE must equal --gpus.

One "step" = one whole VALLE.inference() (prefill + 753 AR passes + 7 NAR stages) of one utterance per GPU;
with ``--batch B`` one step = B utterances per GPU decoded together (BASELINE configs[2]: B = 32, ragged
S in [40, 54]; configs[3] = ``--batch 32 --gpus 8``, 256 utterances per step).  Utterances are independent,
so N GPUs run N replicas (weak scaling): rank 0 owns the inputs, sorts them by length, scatters them over RCCL,
every rank decodes its own, codes are gathered back; scatter and gather are inside the timed region.
Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import glob
import json
import os
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

S_TEXT, P_PROMPT, TOP_K, TEMP = 47, 225, 10, 1.0
D_MODEL, N_HEAD, N_LAYER = 1024, 16, 12
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
METRIC = "AR codec-tokens/sec/GPU + NAR 7-stage p50 latency, d=1024 L=12 10s utterance"


def weight_bytes(d: int, L: int, bpe: int) -> int:
    """parameters one AR token streams (12 d^2 + 13 d per layer, final norm, 1025-way head)."""
    return (L * (12 * d * d + 13 * d) + 2 * d + 1025 * d) * bpe


def kv_bytes_per_row(d: int, L: int, bpe: int) -> int:
    return 2 * L * d * bpe  # K and V of every layer: 49 152 B at cfg1 in bf16


def decode_bytes(launches: int, lens, bpe: int) -> float:
    """Algorithmic bytes of `launches` decode steps over the utterances `lens` = [(S, P, T), ...] that share them: the
    weights once per step + every live utterance's cached rows read and one row written (ctx = S + P + t at step t)."""
    kv = kv_bytes_per_row(D_MODEL, N_LAYER, bpe)
    total = float(launches) * weight_bytes(D_MODEL, N_LAYER, bpe)
    for S, P, T in lens:
        total += kv * (T * (S + P + 1) + T * (T - 1) / 2.0)
    return total


def committed_traffic():
    """HBM traffic per batch-1 AR step from the newest committed PMC summary (profiles/r*_pmc_ar_step.json: FETCH_SIZE x2 per the
    microarch guide's gfx950 correction, a SEPARATE rocprofv3 --pmc pass, not this run).  (value at ctx, source) or (None, None)."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_ar_step.json")))
    if not files or os.environ.get("VX_AR_TP", "") == "0":
        return None, None
    p = json.load(open(files[-1]))
    return p, os.path.relpath(files[-1], ROOT)


def cpu_model_name() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(sd, cfg, x, x_lens, y, budget_s: float):
    """The reference algorithm (no KV cache, fp32, oracle/valle_oracle.inference_faithful — a checked port, kind="port") on a
    bounded sample of the same workload, timed on this box's host cores: once on all the cores this process may use and once
    on ONE thread (what the reference's CLI does: torch.set_num_threads(1), valle/bin/infer.py:262-263).  The sample (number of
    AR steps in front of the 7 NAR stages) is sized from one calibration pass so that each leg takes about `budget_s`."""
    import torch
    from oracle import valle_oracle as vo  # checker / baseline only — never the product path

    m = vo.OracleModel(sd, cfg.decoder_dim, cfg.nhead, cfg.num_decoder_layers, cfg.prefix_mode, cfg.prepend_bos,
                       cfg.num_quantizers)
    ctx0 = S_TEXT + P_PROMPT

    def run(n, skip_nar=False):
        t0 = time.time()
        codes = vo.inference_faithful(m, x, x_lens, y, None, TOP_K, TEMP, exp_noise=torch.ones(n + 2, 1025), max_new_tokens=n,
                                      skip_nar=skip_nar)
        return codes.shape[1], time.time() - t0

    def leg(threads):
        torch.set_num_threads(threads)
        _, t1 = run(1, skip_nar=True)          # two forward passes over ~273 rows: the calibration
        per_pass = t1 / 2.0
        # n AR tokens cost n + 1 passes, the 7 NAR stages about 7 x 1.4 passes (the NAR stack has 1.4 x the AR stack's parameters)
        n = int(max(1, min(16, budget_s / per_pass - 11)))
        print(f"bench.py: cpu baseline on {threads} thread(s): {per_pass:.2f} s per pass, sampling {n} AR steps + NAR", file=sys.stderr, flush=True)
        frames, dt = run(n)
        return frames / dt, dt, n

    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))  # a one-GPU box's CPU share is 16 cores; more threads than that only oversubscribe its quota
    before = torch.get_num_threads()
    v_all, dt_all, n_all = leg(cores)
    v_one, dt_one, n_one = leg(1)
    torch.set_num_threads(before)
    tail = "fp32, no KV cache; later AR steps cost more (O(T^2)), so the full-length CPU rate is lower than this"
    return {
        "value": round(v_all, 3), "unit": "codec-tokens/s", "cores": cores, "kind": "port", "cpu_model": cpu_model_name(),
        "sample": f"first {n_all} of 753 AR steps (ctx {ctx0}..{ctx0 + n_all}) + 7 NAR stages over {ctx0 + n_all} rows, "
                  f"{dt_all:.1f} s wall on {cores} threads; {tail}",
        "one_thread": {"value": round(v_one, 3), "unit": "codec-tokens/s", "cores": 1,
                       "sample": f"first {n_one} AR steps + 7 NAR stages over {ctx0 + n_one} rows, {dt_one:.1f} s wall, "
                                 "torch.set_num_threads(1) as valle/bin/infer.py:262-263 does"},
    }


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--precision", default="bf16", help="bf16 | fp32 (the token-exact parity mode) | fp8nar (bf16 AR, fp8 NAR GEMMs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=12.0, help="seconds of CPU work per leg of the cpu_baseline sample")
    ap.add_argument("--batch", type=int, default=1, help="utterances decoded together per GPU per step (1 = the headline "
                    "batch-1 workload, BASELINE configs[1]; 32 = configs[2], with --gpus 8 configs[3]; 64 + --precision fp8nar + "
                    "--text-len 94 = configs[4])")
    ap.add_argument("--uniform", action="store_true", help="--batch > 1: every utterance S = --text-len instead of the ragged mix")
    ap.add_argument("--text-len", type=int, default=S_TEXT, help="phonemes per utterance (47 -> 753 frames = 10 s; 94 -> 1505 = 20 s)")
    ap.add_argument("--no-graph", action="store_true", help="launch the AR step kernel by kernel (rocprofv3 --pmc cannot follow hipGraph replays)")
    ap.add_argument("--dry-run-cpu", action="store_true", help="tests only: gloo on CPU with a stand-in decoder, to exercise the "
                    "launcher / partition / scatter / gather path without a GPU; the line says so in `data`")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------ launcher
def launch_ranks(args, argv) -> int:
    """Parent of an N-rank run.  Starts N fresh interpreters BEFORE anything here has touched HIP (this process never does),
    relays rank 0's JSON line, fails if any rank fails."""
    n = args.gpus
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=True if r == 0 else None))
    out0, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    line = None
    for ln in (out0 or "").splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if any(rcs) or line is None:
        print(f"bench.py: ranks exited with {rcs}; no result line" if line is None else f"bench.py: ranks exited with {rcs}", file=sys.stderr)
        return 1
    rec = json.loads(line)
    if rec.get("n_gpus") != n:
        print(f"bench.py: result line says n_gpus={rec.get('n_gpus')}, asked for {n}", file=sys.stderr)
        return 1
    print(line, flush=True)
    return 0


# ------------------------------------------------------------------------------------------ one rank
class _DryRunModel:
    """--dry-run-cpu: stands in for the engine so that the launcher / sharding path runs on CPU (tests).  Output length follows
    the reference's natural length 16 S + 1 (valle.py:1047); the codes are a checksum of the inputs."""

    def inference(self, x, x_lens, y, enroll, top_k=-100, temperature=1.0):
        import torch

        T = 16 * int(x_lens[0]) + 1
        base = (y[0].sum(0) + x.sum()) % 1024
        return ((base.reshape(1, 1, -1) + torch.arange(T).reshape(1, T, 1)) % 1024).to(torch.int64)

    def inference_batch(self, utts, top_k=-100, temperature=1.0, seeds=None):
        return [self.inference(u[0], u[1], u[2], None) for u in utts]

    def timings(self):
        return dict(prefill_ms=0.0, decode_ms=1.0, nar_ms=0.0, n_pass=1, launches=1, batch_decode_ms=1.0, batch_launches=1)


def run_rank(args) -> int:
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"bench.py: WORLD_SIZE={world} but --gpus {args.gpus}", file=sys.stderr)
        return 2
    dry = args.dry_run_cpu
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if dry:
        dev = torch.device("cpu")
        if world > 1:
            dist.init_process_group("gloo")
    else:
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
        if world > 1:
            dist.init_process_group("nccl", device_id=dev)  # RCCL

    from valle_amd.config import ModelConfig
    from valle_amd.sharding import gather_lists, plan_partition, scatter_lists
    from valle_amd.weights import synthetic_inputs, synthetic_state_dict

    cfg = ModelConfig(decoder_dim=D_MODEL, nhead=N_HEAD, num_decoder_layers=N_LAYER, prefix_mode=1)
    Bt = args.batch
    ragged = Bt > 1 and not args.uniform
    s_of = (lambda i: args.text_len - 7 + (i % 15)) if ragged else (lambda i: args.text_len)  # 40..54 around 47
    s_max = args.text_len + 7 if ragged else args.text_len
    sd = None
    if dry:
        model, eng = _DryRunModel(), None
    else:
        import __graft_entry__ as ge

        if rank == 0:
            ge.build()
        if world > 1:
            dist.barrier()
        from valle_amd.models import VALLE

        sd = synthetic_state_dict(cfg, seed=0)
        max_audio = ((P_PROMPT + 16 * s_max + 2 + 63) // 64) * 64
        model = VALLE(D_MODEL, N_HEAD, N_LAYER, prefix_mode=1, precision=args.precision, max_text=max(64, s_max), max_audio=max_audio,
                      print_eos=False, no_graph=args.no_graph, max_batch=Bt if Bt > 1 else 0)
        model.load_state_dict(sd)
        model.to(dev).eval()
        eng = model.engine()

    n_total = args.warmup + args.steps
    per_step = world * Bt
    utts = None
    if rank == 0:  # rank 0 owns every utterance of the job
        utts = [synthetic_inputs(s_of(i), P_PROMPT, 8, seed=1 + i) for i in range(n_total * per_step)]

    def run_phase(first_step, n_steps):
        """partition -> scatter -> decode -> gather for the utterances of these steps; returns (#frames made here, codes in
        utterance order on rank 0, per-group engine timings, per-group (S, P, T))."""
        lists = plan = None
        if rank == 0:
            mine_all = utts[first_step * per_step : (first_step + n_steps) * per_step]
            plan = plan_partition([u[0].shape[1] for u in mine_all], world, Bt)
            lists = [[mine_all[i] for i in idx] for idx in plan]
        mine = scatter_lists(lists, n_steps * Bt, dev, world, rank)
        outs, frames, tms, shapes = [], 0, [], []
        for i in range(n_steps):
            t_step = time.perf_counter()
            group = mine[i * Bt : (i + 1) * Bt]
            seed0 = 1234 + ((first_step + i) * world + rank) * Bt
            if Bt > 1:
                res = model.inference_batch(group, top_k=TOP_K, temperature=TEMP, seeds=[seed0 + j for j in range(Bt)])
            else:
                torch.manual_seed(seed0)  # seeds the on-device sampler
                res = [model.inference(group[0][0], group[0][1], group[0][2], None, top_k=TOP_K, temperature=TEMP)]
            outs += res
            frames += sum(c.shape[1] for c in res)
            tms.append((eng or model).timings())
            shapes.append([(u[0].shape[1], u[2].shape[1], c.shape[1]) for u, c in zip(group, res)])
            if os.environ.get("VX_BENCH_VERBOSE") and rank == 0:  # host wall time per step (includes what the device timers do not)
                if not dry:
                    torch.cuda.synchronize(dev)
                print(f"bench.py: step {first_step + i}: {(time.perf_counter() - t_step) * 1e3:.2f} ms wall", file=sys.stderr)
        got = gather_lists(outs, dev, world, rank)
        ordered = None
        if got is not None:
            ordered = [None] * (n_steps * per_step)
            for r, idx in enumerate(plan):
                for j, i in enumerate(idx):
                    ordered[i] = got[r][j]
        return frames, ordered, tms, shapes

    def fence():
        if not dry:
            torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        if not dry:
            torch.cuda.synchronize(dev)

    if args.warmup:
        run_phase(0, args.warmup)
    fence()
    t0 = time.perf_counter()
    frames, outs, tms, shapes = run_phase(args.warmup, args.steps)
    fence()
    dt = time.perf_counter() - t0
    tms_gemm = []
    if Bt > 1 and not dry:
        # GEMM roofline of the batched NAR stages: ONE extra, untimed pass over the last step's utterances with a HIP-event pair
        # around every GEMM launch (VX_TIME_GEMMS, ~500 event records per pass) - never inside the timed region
        os.environ["VX_TIME_GEMMS"] = "1"
        _, _, tms_gemm, _ = run_phase(args.warmup + args.steps - 1, 1)
        os.environ.pop("VX_TIME_GEMMS")
        fence()
    if rank == 0 and os.environ.get("VX_BENCH_DUMP"):  # tests: what rank 0 holds after the gather, in utterance order
        with open(os.environ["VX_BENCH_DUMP"], "w") as fh:
            json.dump([[int(c.shape[1]), int(c.sum())] for c in outs], fh)
    t = torch.tensor([dt, float(frames)], dtype=torch.float64, device=dev)
    if world > 1:
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt, frames = float(tmax[0]), float(tsum[1])

    if rank == 0:
        bpe = 4 if args.precision == "fp32" else 2
        # HIP-event device times (engine stream) of every timed group on this rank: totals for the AR step, p50 for NAR / prefill
        dkey, lkey = ("batch_decode_ms", "batch_launches") if Bt > 1 else ("decode_ms", "launches")
        decode_ms = sum(t_[dkey] for t_ in tms)
        launches = sum(t_[lkey] for t_ in tms)
        step_s = decode_ms * 1e-3 / max(1, launches)
        total_bytes = sum(decode_bytes(t_[lkey], sh, bpe) for t_, sh in zip(tms, shapes))
        step_bytes = total_bytes / max(1, launches)
        achieved = step_bytes / step_s / 1e9
        ar_tokens = sum(T for sh in shapes for (_, _, T) in sh)
        T_show = shapes[-1][0][2]
        if Bt == 1:
            workload = (f"BASELINE configs[1]: d=1024 nhead=16 L=12, batch=1 AR top-k(10) + 7 NAR stages, S={args.text_len} P={P_PROMPT} -> "
                        f"T={T_show} frames x 8 codebooks, one utterance per GPU per step")
        else:
            tag = "configs[3]" if (Bt == 32 and world == 8) else "configs[2]" if Bt == 32 else "configs[4]" if (Bt == 64 and args.precision == "fp8nar") else "configs[2] geometry at another batch size"
            mix = (f"S in [{args.text_len - 7}, {args.text_len + 7}] (T = 16 S + 1 = {16 * (args.text_len - 7) + 1}..{16 * (args.text_len + 7) + 1}, mean {16 * args.text_len + 1})"
                   if ragged else f"S={args.text_len} -> T={T_show}")
            workload = (f"BASELINE {tag}: d=1024 nhead=16 L=12, batch={Bt} concurrent utterances per GPU (padded KV, hipGraph step, one batched "
                        f"prefill and one batched NAR pass per step), {mix}, P={P_PROMPT}, x 8 codebooks; {world * Bt} utterances per step")
        out = {
            "metric": METRIC,
            "value": round(frames / dt, 2), "unit": "codec-tokens/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt * 1e3 / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": {"fp32": "f32", "fp8nar": "bf16 (AR) + fp8 e4m3 GEMM operands (NAR)"}.get(args.precision, args.precision),
            "data": "synthetic" if not dry else "dry-run: no engine (launcher / sharding test on CPU)",
            "config": {"workload": workload, "parallelism": f"replica x{world} (utterances sorted by length, sharded, RCCL scatter/gather)"},
            "ar_tokens_per_s": round(ar_tokens / (decode_ms * 1e-3), 1),
            "ar_step_us": round(step_s * 1e6, 2),
            "prefill_ms": round(statistics.median(t_["prefill_ms"] for t_ in tms), 3),
            "nar_7stage_ms": round(statistics.median(t_["nar_ms"] for t_ in tms), 3),  # p50 over the timed groups
            "roofline": {"bound": "hbm", "kernel": "AR decode step (hipGraph of one token's kernels)" if Bt == 1 else
                         f"batched AR decode step (hipGraph, {Bt} slots)",
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None, "bytes_per_launch": int(step_bytes)},
        }
        g_ms = sum(t_.get("nar_gemm_ms", 0.0) for t_ in tms_gemm)
        g_fl = sum(t_.get("nar_gemm_flops", 0.0) for t_ in tms_gemm)
        if Bt > 1 and g_ms > 0:
            fp8 = args.precision == "fp8nar"
            peak = 5000.0 if fp8 else 2500.0  # MI355X_MICROARCH.md: dense fp8 / bf16 MFMA peak, TFLOP/s
            tf = g_fl / (g_ms * 1e-3) / 1e12
            gemm = {"bound": "mfma", "kernel": ("NAR stage GEMMs on MXFP8 (QKV, FFN1, FFN2: mx256p_kernel)" if fp8 else
                                                  "NAR stage GEMMs in bf16 (QKV, out-projection, FFN1, FFN2: mfma256p_kernel)"),
                    "achieved": round(tf, 1), "peak": peak, "unit": "TFLOP/s", "frac": round(tf / peak, 4), "traffic": None,
                    "flops_per_launch": int(g_fl / max(1, len(tms_gemm))), "ms_per_launch": round(g_ms / max(1, len(tms_gemm)), 3),
                    "note": "algorithmic 2 M N K of every GEMM of one NAR pass / HIP-event time around its launches on the engine stream, "
                            "collected in ONE extra pass after the timed region (the timed steps carry no extra events)"}
            if fp8:  # configs[4]: the fp8 GEMMs are the kernel this configuration is about; the AR step's line moves aside
                out["ar_step_roofline"], out["roofline"] = out["roofline"], gemm
            else:
                out["nar_gemm_roofline"] = gemm
        if Bt == 1 and args.precision == "bf16" and not dry:
            prof, src = committed_traffic()
            if prof is not None:
                ctx_mean = args.text_len + P_PROMPT + (T_show - 1) / 2.0
                if "gemv_bytes_per_step_corrected" in prof:  # round 1-2 summaries: GEMV and attention launches apart
                    out["roofline"]["traffic"] = int(prof["gemv_bytes_per_step_corrected"] + prof["attn_bytes_per_ctx_row_corrected"] * ctx_mean)
                else:  # the sharded step reads its K / V rows inside the attention launch: measured bytes at the collection's
                    # mean context, moved to this run's by the algorithmic 49 152 B per cached row
                    out["roofline"]["traffic"] = int(prof["hbm_bytes_per_step_corrected"] + 49152 * (ctx_mean - prof["ctx_mean"]))
                out["roofline"]["traffic_source"] = (f"{src}: FETCH_SIZE x2 from a separate rocprofv3 --pmc pass of an eager (no-graph) run, "
                                                     "quoted at this run's mean context; NOT measured in this run")
                if prof.get("write_bytes_per_step"):
                    out["roofline"]["traffic_writes"] = int(prof["write_bytes_per_step"])  # WRITE_SIZE of the same collection
        if world == 1 and not args.no_cpu_baseline and not dry:
            x, x_lens, y = synthetic_inputs(S_TEXT, P_PROMPT, 8, seed=1)
            out["cpu_baseline"] = cpu_baseline(sd, cfg, x, x_lens, y, args.cpu_budget)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()
    return 0


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args, argv))
    sys.exit(run_rank(args))


if __name__ == "__main__":
    main()
