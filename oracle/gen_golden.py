"""Generate the golden vectors under tests/golden/ by running the UNMODIFIED reference.

Build-container only (needs /root/reference; see oracle/ref_harness.py for the import stubs).
Usage:  PYTHONDONTWRITEBYTECODE=1 python -m oracle.gen_golden [case ...]     (default: all small)
        PYTHONDONTWRITEBYTECODE=1 python -m oracle.gen_golden cfg1            (≈10+ min of CPU)

Each fixture is pure data: the model hyper-parameters and seeds that regenerate the synthetic
weights/inputs (valle_amd.weights), the inputs themselves, the reference's output codes, a few
logits rows captured by forward hooks on the reference's own predict layers, and — for sampled
runs — the Exp(1) noise the reference's torch.multinomial consumed (re-drawn from the same
seed; that oracle(noise) == reference(multinomial) is asserted here before anything is written).
"""
from __future__ import annotations

import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

import valle_amd  # noqa: E402,F401
from valle_amd.config import ModelConfig  # noqa: E402
from valle_amd.weights import synthetic_inputs, synthetic_state_dict  # noqa: E402
from oracle import valle_oracle as vo  # noqa: E402
from oracle.ref_harness import build_reference_model  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")

# name -> (model kwargs, S, P, top_k, temperature, sample_seed, enroll_len, ar probe steps)
CASES = {
    # BASELINE.json configs[0]: d=256 nhead=4 L=4, greedy, 3 s prompt, S=10 -> 161 tokens
    "cfg0_greedy": (dict(decoder_dim=256, nhead=4, num_decoder_layers=4, prefix_mode=1), 10, 225, 1, 1.0, None, None, (0, 1, 80, 160)),
    "cfg0_topk10": (dict(decoder_dim=256, nhead=4, num_decoder_layers=4, prefix_mode=1), 10, 225, 10, 1.0, 1234, None, (0, 1, 80, 160)),
    "cfg0_temp_nofilter": (dict(decoder_dim=256, nhead=4, num_decoder_layers=4, prefix_mode=1), 6, 40, -100, 0.7, 77, None, (0, 50)),
    # options the reference's own smoke test walks through (valle_test.py:106-135)
    "tiny_mode0": (dict(decoder_dim=128, nhead=2, num_decoder_layers=2, prefix_mode=0), 8, 16, 5, 1.0, 5, None, (0, 7)),
    "tiny_mode1_bos": (dict(decoder_dim=128, nhead=2, num_decoder_layers=2, prefix_mode=1, prepend_bos=True), 8, 16, 5, 1.0, 6, None, (0, 7)),
    "tiny_mode2_q6": (dict(decoder_dim=128, nhead=2, num_decoder_layers=2, prefix_mode=2, num_quantizers=6), 8, 16, 5, 1.0, 7, 4, (0, 7)),
    "tiny_mode4": (dict(decoder_dim=128, nhead=2, num_decoder_layers=2, prefix_mode=4), 9, 16, 3, 1.3, 8, 3, (0, 7)),
    "tiny_q1": (dict(decoder_dim=128, nhead=2, num_decoder_layers=2, prefix_mode=0, num_quantizers=1), 5, 12, 1, 1.0, None, None, (0, 3)),
    "tiny_q2_unshared": (dict(decoder_dim=128, nhead=2, num_decoder_layers=2, prefix_mode=1, num_quantizers=2, share_embedding=False), 5, 12, 1, 1.0, None, None, (0, 3)),
    # nar_scale_factor = 0.5 (valle_test.py:131: "params.scale_factor = 0.5"): NAR stack of half width / heads / depth
    "tiny_scale05": (dict(decoder_dim=256, nhead=4, num_decoder_layers=4, prefix_mode=1, scale_factor=0.5), 6, 12, 4, 1.0, 11, None, (0, 9)),
    # empty prompt: only possible with prepend_bos (the BOS row is then the whole audio prefix, valle.py:1006-1007)
    "tiny_bos_empty_prompt": (dict(decoder_dim=128, nhead=2, num_decoder_layers=2, prefix_mode=1, prepend_bos=True), 5, 0, 3, 1.0, 41, None, (0, 5)),
    "tiny_bos_empty_prompt_mode0": (dict(decoder_dim=128, nhead=2, num_decoder_layers=2, prefix_mode=0, prepend_bos=True), 4, 0, 1, 1.0, None, None, (0, 5)),
    # norm_first=False, the ordering the reference's own smoke test builds (valle_test.py:105: "params.norm_first = False")
    "tiny_postnorm": (dict(decoder_dim=128, nhead=2, num_decoder_layers=2, prefix_mode=1, norm_first=False), 7, 14, 4, 1.0, 21, None, (0, 5)),
    "tiny_postnorm_mode0": (dict(decoder_dim=128, nhead=2, num_decoder_layers=3, prefix_mode=0, norm_first=False), 6, 10, 1, 1.0, None, None, (0, 5)),
    "cfg0_postnorm": (dict(decoder_dim=256, nhead=4, num_decoder_layers=4, prefix_mode=1, norm_first=False), 10, 60, 10, 1.0, 4321, None, (0, 1, 80, 160)),
    # add_prenet=True (valle.py:96-123; the reference's smoke test builds its models with it, valle_test.py:106)
    "tiny_prenet": (dict(decoder_dim=128, nhead=2, num_decoder_layers=2, prefix_mode=1, add_prenet=True), 7, 14, 4, 1.0, 31, None, (0, 5)),
    "tiny_prenet_postnorm_mode0": (dict(decoder_dim=128, nhead=2, num_decoder_layers=2, prefix_mode=0, add_prenet=True, norm_first=False), 6, 11, 3, 1.0, 32, None, (0, 5)),
    "tiny_prenet_mode2_bos": (dict(decoder_dim=128, nhead=2, num_decoder_layers=2, prefix_mode=2, add_prenet=True, prepend_bos=True, num_quantizers=7), 9, 12, 1, 1.0, None, 4, (0, 5)),
    # BASELINE.json configs[1]: d=1024 nhead=16 L=12, top-k 10, S=47 -> 753 tokens x 8 codebooks
    # the geometry and option walk of the reference's own test (valle_test.py:90-135): decoder_dim 64 / nhead 16 (head_dim 4),
    # 4 layers, post-norm + prenets, text (1,8), prompt (1,16,8), default sampling (top_k=-100); scale 0.5 from the second
    # iteration on, prepend_bos toggling, one quantizer fewer each time
    "reftest_mode0": (dict(decoder_dim=64, nhead=16, num_decoder_layers=4, prefix_mode=0, norm_first=False, add_prenet=True), 8, 16, -100, 1.0, 51, 2, (0, 5)),
    "reftest_mode1_scale05_bos_q7": (dict(decoder_dim=64, nhead=16, num_decoder_layers=4, prefix_mode=1, norm_first=False, add_prenet=True,
                                          scale_factor=0.5, prepend_bos=True, num_quantizers=7), 8, 16, -100, 1.0, 52, 2, (0, 5)),
    "reftest_mode2_scale05_q6": (dict(decoder_dim=64, nhead=16, num_decoder_layers=4, prefix_mode=2, norm_first=False, add_prenet=True,
                                      scale_factor=0.5, num_quantizers=6), 8, 16, -100, 1.0, 53, 2, (0, 5)),
    "cfg1_topk10": (dict(decoder_dim=1024, nhead=16, num_decoder_layers=12, prefix_mode=1), 47, 225, 10, 1.0, 1234, None, (0, 1, 376, 752)),
    # BASELINE.json configs[4]'s utterance: the cfg1 model, S=94 -> 1505 tokens (20 s) x 8 codebooks.  The no-cache reference is
    # O(T^2): ~1 h of CPU on 8 threads; run with GOLDEN_NARSTATS=1 so that the per-stage NAR statistics come from the same pass
    "cfg4_s94_topk10": (dict(decoder_dim=1024, nhead=16, num_decoder_layers=12, prefix_mode=1), 94, 225, 10, 1.0, 4321, None, (0, 1, 752, 1504)),
}
SMALL = [k for k in CASES if not k.startswith(("cfg1", "cfg4"))]
V = 1025


def run_case(name: str):
    kw, S, P, top_k, temp, sseed, enroll, probes = CASES[name]
    cfg = ModelConfig(**kw)
    sd = synthetic_state_dict(cfg, seed=0)
    x, x_lens, y = synthetic_inputs(S, P, 8, seed=1)
    enroll_x_lens = None if enroll is None else torch.tensor([enroll], dtype=torch.int32)
    ref = build_reference_model(cfg, sd)

    ar_log, nar_log, nar_full = [], {}, {}
    want_stats = os.environ.get("GOLDEN_NARSTATS", "0") == "1"
    ref.ar_predict_layer.register_forward_hook(lambda m, i, o: ar_log.append(o.detach()[0].clone()))
    if cfg.num_quantizers > 1:
        # tied predict layers are distinct modules sharing a Parameter (valle.py:268-271)
        def hook(m, i, o, si):
            nar_log[si] = o.detach()[0, :8].clone()
            if want_stats:
                nar_full[si] = o.detach()[0].clone()
        for si, layer in enumerate(ref.nar_predict_layers):
            layer.register_forward_hook(lambda m, i, o, si=si: hook(m, i, o, si))

    if sseed is not None:
        torch.manual_seed(sseed)
    t0 = time.time()
    with torch.no_grad():
        codes = ref.inference(x, x_lens, y, enroll_x_lens=enroll_x_lens, top_k=top_k, temperature=temp)
    dt = time.time() - t0
    n_pass = len(ar_log)
    print(f"[{name}] reference: codes {tuple(codes.shape)} passes {n_pass} in {dt:.1f}s", flush=True)

    noise = None
    if top_k != 1:
        # the reference drew one (1,V) exponential per pass from the global CPU generator
        torch.manual_seed(sseed)
        noise = torch.stack([torch.empty(1, V).exponential_(1)[0] for _ in range(n_pass)])

    # pin the oracle before writing anything
    m = vo.OracleModel(sd, cfg.decoder_dim, cfg.nhead, cfg.num_decoder_layers, cfg.prefix_mode,
                       cfg.prepend_bos, cfg.num_quantizers, cfg.scale_factor, cfg.norm_first, cfg.add_prenet)
    tr = {}
    oc = vo.inference_cached(m, x, x_lens, y, enroll_x_lens, top_k, temp, noise, trace=tr)
    assert torch.equal(oc, codes), f"{name}: cached oracle differs from the reference"
    err = max(float((tr["ar_logits"][i] - ar_log[i]).abs().max()) for i in range(len(tr["ar_logits"])))
    print(f"[{name}] cached oracle == reference codes; max |logit diff| {err:.2e}", flush=True)
    assert err < 1e-3
    if cfg.decoder_dim <= 256:
        of = vo.inference_faithful(m, x, x_lens, y, enroll_x_lens, top_k, temp, noise)
        assert torch.equal(of, codes), f"{name}: faithful oracle differs from the reference"

    probes = [p for p in probes if p < n_pass]
    out = dict(
        cfg=np.array([cfg.decoder_dim, cfg.nhead, cfg.num_decoder_layers, cfg.prefix_mode,
                      int(cfg.prepend_bos), cfg.num_quantizers, int(cfg.share_embedding)], dtype=np.int32),
        weight_seed=np.int32(0), input_seed=np.int32(1), scale_factor=np.float32(cfg.scale_factor),
        norm_first=np.int32(int(cfg.norm_first)), add_prenet=np.int32(int(cfg.add_prenet)),
        x=x.numpy().astype(np.int16), x_lens=x_lens.numpy(), y=y.numpy().astype(np.int16),
        enroll=np.int32(-1 if enroll is None else enroll),
        top_k=np.int32(top_k), temperature=np.float32(temp),
        codes=codes.numpy().astype(np.int16), n_pass=np.int32(n_pass),
        ar_probe_steps=np.array(probes, dtype=np.int32),
        ar_probe_logits=torch.stack([ar_log[p] for p in probes]).numpy(),
    )
    if nar_log:
        out["nar_probe_logits"] = torch.stack([nar_log[i] for i in sorted(nar_log)]).numpy()
    if noise is not None:
        out["exp_noise"] = noise.numpy()
        out["sample_seed"] = np.int32(sseed)
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, f"{name}.npz"), **out)
    print(f"[{name}] wrote {os.path.getsize(os.path.join(OUT, name + '.npz')) / 1024:.0f} KiB", flush=True)
    if want_stats and nar_full:
        # the same record oracle/gen_nar_stats.py writes, from this pass's hooks (saves a second O(T^2) reference run)
        from oracle.gen_nar_stats import write_stats
        write_stats(name, codes, torch.stack([nar_full[i] for i in sorted(nar_full)]))


def run_continual(name: str, prefix_mode: int, S: int, T: int, add_prenet: bool = False):
    """valle.py:1139-1238 via bin/infer.py:224-230 (--continual)."""
    cfg = ModelConfig(decoder_dim=128, nhead=2, num_decoder_layers=2, prefix_mode=prefix_mode, add_prenet=add_prenet)
    sd = synthetic_state_dict(cfg, seed=0)
    x, x_lens, y = synthetic_inputs(S, T, 8, seed=3)
    ref = build_reference_model(cfg, sd)
    with torch.no_grad():
        codes = ref.continual(x, x_lens, y)
    m = vo.OracleModel(sd, cfg.decoder_dim, cfg.nhead, cfg.num_decoder_layers, cfg.prefix_mode, False, 8, add_prenet=add_prenet)
    assert torch.equal(vo.continual(m, x, x_lens, y), codes), f"{name}: oracle continual differs from the reference"
    print(f"[{name}] reference continual {tuple(codes.shape)}; oracle == reference", flush=True)
    np.savez_compressed(os.path.join(OUT, f"{name}.npz"),
                        cfg=np.array([128, 2, 2, prefix_mode, 0, 8, 1], dtype=np.int32), weight_seed=np.int32(0),
                        add_prenet=np.int32(int(add_prenet)),
                        x=x.numpy().astype(np.int16), x_lens=x_lens.numpy(), y=y.numpy().astype(np.int16),
                        codes=codes.numpy().astype(np.int16))


# ---- VALL-F (valle.py:566-710).  The reference's layers run under the torch-1.13.1 TransformerDecoder loop restated in
# oracle/ref_harness.py (the installed torch 2.10 container rejects the reference's tuple inputs): see oracle/valle_oracle.py.
# name -> (model kwargs, S, P, top_k, temperature, sample_seed, enroll_len, ar probe steps)
CASES_F = {
    "vallf_mode1": (dict(decoder_dim=128, nhead=2, num_decoder_layers=2, prefix_mode=1), 7, 14, 5, 1.0, 3, None, (0, 5)),
    "vallf_mode0_bos": (dict(decoder_dim=128, nhead=2, num_decoder_layers=2, prefix_mode=0, prepend_bos=True), 6, 10, 1, 1.0, None, None, (0, 5)),
    "vallf_postnorm_prenet_mode2_q6": (dict(decoder_dim=128, nhead=2, num_decoder_layers=2, prefix_mode=2, num_quantizers=6, norm_first=False,
                                            add_prenet=True), 8, 12, 4, 1.2, 9, 4, (0, 5)),
    "vallf_postnorm_mode4_bos": (dict(decoder_dim=128, nhead=2, num_decoder_layers=3, prefix_mode=4, norm_first=False, prepend_bos=True), 9, 11, 3, 1.0, 13, 3, (0, 5)),
    "vallf_prenet_scale05": (dict(decoder_dim=256, nhead=4, num_decoder_layers=4, prefix_mode=1, add_prenet=True, scale_factor=0.5), 6, 12, 4, 1.0, 11, None, (0, 9)),
    # the geometry and option walk of the reference's own VALL-F test (valle_test.py:37-89): decoder_dim 64 / nhead 16 (head_dim 4),
    # 4 layers, pre-norm, text (1,8), prompt (1,16,8), default sampling (top_k=-100), prepend_bos toggling, 1 / 2 / 3 quantizers
    "vallf_reftest_mode0_bos_q1": (dict(decoder_dim=64, nhead=16, num_decoder_layers=4, prefix_mode=0, prepend_bos=True, num_quantizers=1), 8, 16, -100, 1.0, 61, 2, (0, 5)),
    "vallf_reftest_mode1_q2": (dict(decoder_dim=64, nhead=16, num_decoder_layers=4, prefix_mode=1, num_quantizers=2), 8, 16, -100, 1.0, 62, 3, (0, 5)),
    "vallf_reftest_mode2_bos_q3": (dict(decoder_dim=64, nhead=16, num_decoder_layers=4, prefix_mode=2, prepend_bos=True, num_quantizers=3), 8, 16, -100, 1.0, 63, 2, (0, 5)),
    # head_dim 64 at BASELINE configs[0]'s width: the geometry the MFMA row kernels and the vectorised decode attention serve
    "vallf_cfg0_topk10": (dict(decoder_dim=256, nhead=4, num_decoder_layers=4, prefix_mode=1), 10, 60, 10, 1.0, 1234, None, (0, 1, 80, 160)),
}


def run_case_f(name: str):
    kw, S, P, top_k, temp, sseed, enroll, probes = CASES_F[name]
    cfg = ModelConfig(model_name="VALL-F", **kw)
    sd = synthetic_state_dict(cfg, seed=0)
    x, x_lens, y = synthetic_inputs(S, P, 8, seed=1)
    enroll_x_lens = None if enroll is None else torch.tensor([enroll], dtype=torch.int32)
    ref = build_reference_model(cfg, sd)
    ar_log, nar_log = [], {}
    ref.ar_predict_layer.register_forward_hook(lambda m, i, o: ar_log.append(o.detach()[0].clone()))
    if cfg.num_quantizers > 1:
        for si, layer in enumerate(ref.nar_predict_layers):
            layer.register_forward_hook(lambda m, i, o, si=si: nar_log.__setitem__(si, o.detach()[0, :8].clone()))
    if sseed is not None:
        torch.manual_seed(sseed)
    with torch.no_grad():
        codes = ref.inference(x, x_lens, y, enroll_x_lens=enroll_x_lens, top_k=top_k, temperature=temp)
    n_pass = len(ar_log)
    noise = None
    if top_k != 1:
        torch.manual_seed(sseed)
        noise = torch.stack([torch.empty(1, V).exponential_(1)[0] for _ in range(n_pass)])
    m = vo.OracleModelF(sd, cfg.decoder_dim, cfg.nhead, cfg.num_decoder_layers, prefix_mode=cfg.prefix_mode, prepend_bos=cfg.prepend_bos,
                        num_quantizers=cfg.num_quantizers, nar_scale_factor=cfg.scale_factor, norm_first=cfg.norm_first, add_prenet=cfg.add_prenet)
    tr = {}
    oc = vo.inference_f(m, x, x_lens, y, enroll_x_lens, top_k, temp, noise, trace=tr)
    assert torch.equal(oc, codes), f"{name}: VALL-F oracle differs from the reference"
    err = max(float((tr["ar_logits"][i] - ar_log[i]).abs().max()) for i in range(n_pass))
    assert err < 1e-3
    print(f"[{name}] reference VALL-F codes {tuple(codes.shape)}, {n_pass} passes; oracle == reference, max |logit diff| {err:.2e}", flush=True)
    probes = [p for p in probes if p < n_pass]
    out = dict(
        cfg=np.array([cfg.decoder_dim, cfg.nhead, cfg.num_decoder_layers, cfg.prefix_mode,
                      int(cfg.prepend_bos), cfg.num_quantizers, int(cfg.share_embedding)], dtype=np.int32),
        vallf=np.int32(1), weight_seed=np.int32(0), input_seed=np.int32(1), scale_factor=np.float32(cfg.scale_factor),
        norm_first=np.int32(int(cfg.norm_first)), add_prenet=np.int32(int(cfg.add_prenet)),
        x=x.numpy().astype(np.int16), x_lens=x_lens.numpy(), y=y.numpy().astype(np.int16),
        enroll=np.int32(-1 if enroll is None else enroll), top_k=np.int32(top_k), temperature=np.float32(temp),
        codes=codes.numpy().astype(np.int16), n_pass=np.int32(n_pass), ar_probe_steps=np.array(probes, dtype=np.int32),
        ar_probe_logits=torch.stack([ar_log[p] for p in probes]).numpy(),
    )
    if nar_log:
        out["nar_probe_logits"] = torch.stack([nar_log[i] for i in sorted(nar_log)]).numpy()
    if noise is not None:
        out["exp_noise"] = noise.numpy()
        out["sample_seed"] = np.int32(sseed)
    np.savez_compressed(os.path.join(OUT, f"{name}.npz"), **out)


CONTINUAL = {"continual_mode0": (0, 7, 41), "continual_mode1": (1, 9, 64),
             "continual_prenet_mode0": (0, 6, 37, True), "continual_prenet_mode1": (1, 8, 50, True)}

if __name__ == "__main__":
    torch.set_num_threads(int(os.environ.get("GOLDEN_THREADS", "8")))
    for c in (sys.argv[1:] or SMALL + list(CONTINUAL) + list(CASES_F)):
        if c in CONTINUAL:
            run_continual(c, *CONTINUAL[c])
        elif c in CASES_F:
            run_case_f(c)
        else:
            run_case(c)
