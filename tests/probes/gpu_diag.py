"""Non-asserting GPU diagnostics: prints error metrics of every kernel / engine mode against the
oracle, as JSON lines, so tolerances in the tests are set from measurements.  Run on the box:
    python tests/probes/gpu_diag.py > gpurun_out/diag.jsonl"""
import json
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import __graft_entry__ as ge  # noqa: E402

ge.build()
from conftest import Golden  # noqa: E402
from oracle import valle_oracle as vo  # noqa: E402
from valle_amd import engine as E  # noqa: E402
from valle_amd.models import VALLE  # noqa: E402


def emit(**kw):
    print(json.dumps(kw), flush=True)


def guard(fn):
    def w(*a, **k):
        try:
            return fn(*a, **k)
        except Exception as ex:  # noqa: BLE001
            emit(test=fn.__name__, args=str(a), error=repr(ex), tb=traceback.format_exc()[-800:])
    return w


def model(g, precision, **kw):
    c = g.cfg
    m = VALLE(c.decoder_dim, c.nhead, c.num_decoder_layers, prefix_mode=c.prefix_mode, share_embedding=c.share_embedding,
              prepend_bos=c.prepend_bos, num_quantizers=c.num_quantizers, precision=precision, max_text=128, max_audio=1280,
              print_eos=False, **kw)
    m.load_state_dict(g.state_dict())
    return m.to("cuda:0").eval()


@guard
def engine_case(name, precision, **kw):
    g = Golden(name)
    t0 = time.time()
    m = model(g, precision, trace_logits=True, **kw)
    e = m.engine()
    t_load = time.time() - t0
    t0 = time.time()
    codes = m.inference(g.x.cuda(), g.x_lens.cuda(), g.y.cuda(), g.enroll_x_lens, top_k=g.top_k, temperature=g.temperature,
                        exp_noise=g.exp_noise).cpu()
    dt = time.time() - t0
    out = dict(test="engine_free_running", case=name, precision=precision, opts=kw, load_s=round(t_load, 2), infer_s=round(dt, 3),
               shape=list(codes.shape), timings=e.timings())
    if codes.shape == g.codes.shape:
        out["ar_equal"] = bool(torch.equal(codes[..., 0], g.codes[..., 0]))
        out["ar_first_diff"] = int((codes[0, :, 0] != g.codes[0, :, 0]).float().argmax()) if not out["ar_equal"] else -1
        out["all_agree"] = float((codes == g.codes).float().mean())
        out["per_stage_agree"] = [float((codes[0, :, j] == g.codes[0, :, j]).float().mean()) for j in range(codes.shape[-1])]
    errs = {}
    for step, ref in zip(g.ar_probe_steps, g.ar_probe_logits):
        got = e.read("ar_logits", (1025,), offset_bytes=step * 1025 * 4)
        errs[step] = [float((got - ref).abs().max()), float(ref.abs().max())]
    out["ar_logit_err"] = errs
    if g.nar_probe_logits is not None:
        got = e.read("nar_logits", (8, 1024))
        out["nar_last_stage_err"] = [float((got - g.nar_probe_logits[-1]).abs().max()), float(g.nar_probe_logits[-1].abs().max())]
    emit(**out)
    m._drop_engine()


@guard
def forced_case(name, **kw):
    g = Golden(name)
    m = model(g, "bf16", trace_logits=True, **kw)
    e = m.engine()
    text, prompts = g.x[0], g.y[0, :, : g.cfg.num_quantizers].contiguous()
    forced = g.codes[0, :, 0].contiguous()
    e.ar_prefill(text, prompts[:, 0].contiguous())
    e.ar_decode(top_k=g.top_k, temperature=g.temperature, exp_noise=g.exp_noise, forced=forced)
    toks, reason, n_pass = e.ar_result()
    got = e.read("ar_logits", (n_pass, 1025))
    samp = e.read("ar_sampled", (n_pass,), dtype=torch.int32)
    out = dict(test="bf16_teacher_forced", case=name, opts=kw, n_pass=n_pass, timings=e.timings())
    if g.cfg.decoder_dim < 1024:
        tr = {}
        vo.inference_cached(g.oracle(), g.x, g.x_lens, g.y, g.enroll_x_lens, g.top_k, g.temperature, g.exp_noise, trace=tr,
                            forced=forced, skip_nar=True)
        ref = torch.stack(tr["ar_logits"])
        rel = ((got - ref).abs().amax(1) / ref.abs().amax(1))
        out["rel_err_max"] = float(rel.max())
        out["rel_err_mean"] = float(rel.mean())
        out["argmax_agree"] = float((got.argmax(1) == ref.argmax(1)).float().mean())
        out["sample_agree"] = float((samp[:-1].long() == torch.tensor(tr["ar_samples"][: n_pass - 1])).float().mean())
    else:
        out["probe_err"] = {s: [float((got[s] - r).abs().max()), float(r.abs().max())] for s, r in zip(g.ar_probe_steps, g.ar_probe_logits)}
        out["sample_agree_vs_forced"] = float((samp[:-1].long() == forced).float().mean())
    tn = text if g.cfg.prefix_mode not in (2, 4) else torch.cat([text[:1], text[int(g.enroll_x_lens.max()) - 1:]])
    if g.cfg.num_quantizers > 1:
        codes = e.nar(tn, prompts, forced).cpu()
        out["nar_per_stage_agree"] = [float((codes[:, j] == g.codes[0, :, j]).float().mean()) for j in range(1, codes.shape[1])]
        out["timings_nar"] = e.timings()
    emit(**out)
    m._drop_engine()


if __name__ == "__main__":
    emit(device=torch.cuda.get_device_name(0), torch=torch.__version__)
    for n in ["tiny_mode0", "cfg0_greedy", "cfg0_topk10"]:
        engine_case(n, "fp32")
    engine_case("cfg0_topk10", "fp32", no_graph=True)
    engine_case("cfg0_topk10", "bf16", simple_rows=True)
    engine_case("cfg0_topk10", "bf16")
    forced_case("cfg0_topk10", simple_rows=True)
    forced_case("cfg0_topk10")
    engine_case("cfg1_topk10", "fp32")
    forced_case("cfg1_topk10")
    engine_case("cfg1_topk10", "bf16")
