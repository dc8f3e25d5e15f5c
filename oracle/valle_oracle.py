"""CPU ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.

A plain PyTorch-CPU fp32 restatement of the reference's ``VALLE.inference`` hot path
(/root/reference/valle/models/valle.py:961-1137) written functionally over a ``state_dict``.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module; nothing under ``vall-e_amd/`` does.

Parity status: **pinned** — ``oracle/gen_golden.py`` imports the unmodified reference model
code in the build container, loads the same name-seeded synthetic weights into it and writes
the vectors under ``tests/golden/``; ``tests/test_oracle_golden.py`` checks this file against
them (tokens exact, logits <= 1e-5).  The reference's own tests hold no numerical vectors for
this path (valle/tests/valle_test.py:91-135 are shape smoke tests).

Two variants:
  * ``inference_faithful`` — no KV cache, whole sequence recomputed at every AR step, same
    op order as valle.py:1012-1057 / 1115-1134.  This is what is timed as "the reference's
    CPU path" (bench.py cpu_baseline, kind "port").
  * ``ArCache`` / ``inference_cached`` — prefill + single-token steps over a KV cache; the
    executable spec of the HIP kernels.  Identical results.

The arithmetic bottoms out in the same torch ops the reference reaches: F.linear,
F.layer_norm, F.scaled_dot_product_attention, F.softmax, torch.topk, torch.multinomial.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

NUM_AUDIO_TOKENS = 1024  # valle/models/macros.py:5
LN_EPS = 1e-5


# ----------------------------------------------------------------------------- building blocks
def sine_pe(length: int, dim: int) -> torch.Tensor:
    """valle/modules/embedding.py:75-88 (fp32 table, sin on even / cos on odd channels)."""
    position = torch.arange(0, length, dtype=torch.float32).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, dim, 2, dtype=torch.float32) * -(math.log(10000.0) / dim))
    pe = torch.zeros(length, dim)
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe


def add_position(x: torch.Tensor, alpha: torch.Tensor, start: int = 0) -> torch.Tensor:
    """SinePositionalEmbedding.forward, embedding.py:93-97: x*1.0 + alpha*pe[start:start+T]."""
    T, d = x.shape[-2], x.shape[-1]
    pe = sine_pe(max(4000, start + T), d)[start : start + T]
    return x * 1.0 + alpha * pe


def layer_norm(x, w, b):
    """valle/modules/transformer.py:57-74 -> F.layer_norm(eps=1e-5)."""
    return F.layer_norm(x, (x.shape[-1],), w, b, LN_EPS)


def ada_layer_norm(x, stage_emb, pw, pb, w, b):
    """AdaptiveLayerNorm.forward, transformer.py:93-108."""
    d = x.shape[-1]
    weight, bias = torch.split(F.linear(stage_emb, pw, pb), d, dim=-1)
    return weight * layer_norm(x, w, b) + bias


def self_attention(x, in_w, in_b, out_w, out_b, nhead: int, mask: Optional[torch.Tensor]):
    """MultiheadAttention.forward -> F.multi_head_attention_forward (activation.py:407-427;
    torch/nn/functional.py:6206-6640): packed in-proj, heads = contiguous channel blocks,
    SDPA with additive -inf mask, out-proj.  x: (N, d) single sequence."""
    N, d = x.shape
    hd = d // nhead
    qkv = F.linear(x, in_w, in_b)
    q, k, v = qkv.chunk(3, dim=-1)
    q = q.reshape(N, nhead, hd).transpose(0, 1).unsqueeze(0)  # (1,H,N,hd)
    k = k.reshape(N, nhead, hd).transpose(0, 1).unsqueeze(0)
    v = v.reshape(N, nhead, hd).transpose(0, 1).unsqueeze(0)
    am = None
    if mask is not None:  # bool, True = masked (functional.py:6154-6180)
        am = torch.zeros(mask.shape, dtype=x.dtype).masked_fill_(mask, float("-inf"))
    o = F.scaled_dot_product_attention(q, k, v, am, 0.0, False)
    o = o.squeeze(0).transpose(0, 1).reshape(N, d)
    return F.linear(o, out_w, out_b), (k.squeeze(0), v.squeeze(0))


class _Layer:
    """Weights of one TransformerEncoderLayer (transformer.py:181-258)."""

    def __init__(self, sd, prefix, adaptive):
        g = lambda n: sd[f"{prefix}.{n}"]
        self.in_w, self.in_b = g("self_attn.in_proj_weight"), g("self_attn.in_proj_bias")
        self.out_w, self.out_b = g("self_attn.out_proj.weight"), g("self_attn.out_proj.bias")
        self.w1, self.b1 = g("linear1.weight"), g("linear1.bias")
        self.w2, self.b2 = g("linear2.weight"), g("linear2.bias")
        self.adaptive = adaptive
        if adaptive:
            self.n = [
                (g(f"{n}.project_layer.weight"), g(f"{n}.project_layer.bias"), g(f"{n}.norm.weight"), g(f"{n}.norm.bias"))
                for n in ("norm1", "norm2")
            ]
        else:
            self.n = [(g(f"{n}.weight"), g(f"{n}.bias")) for n in ("norm1", "norm2")]

    def norm(self, i, x, stage_emb):
        if self.adaptive:
            pw, pb, w, b = self.n[i]
            return ada_layer_norm(x, stage_emb, pw, pb, w, b)
        w, b = self.n[i]
        return layer_norm(x, w, b)


def encoder_layer(L: _Layer, x, nhead, mask, stage_emb=None, norm_first: bool = True):
    """TransformerEncoderLayer.forward: pre-norm branch transformer.py:296-302, post-norm branch 303-308; blocks 315-334."""
    if norm_first:
        a, kv = self_attention(L.norm(0, x, stage_emb), L.in_w, L.in_b, L.out_w, L.out_b, nhead, mask)
        x = x + a
        x = x + F.linear(F.relu(F.linear(L.norm(1, x, stage_emb), L.w1, L.b1)), L.w2, L.b2)
        return x, kv
    a, kv = self_attention(x, L.in_w, L.in_b, L.out_w, L.out_b, nhead, mask)
    x = L.norm(0, x + a, stage_emb)
    x = L.norm(1, x + F.linear(F.relu(F.linear(x, L.w1, L.b1)), L.w2, L.b2), stage_emb)
    return x, kv


def text_prenet(sd, prefix: str, x: torch.Tensor) -> torch.Tensor:
    """{ar,nar}_text_prenet in eval mode (valle.py:97-113): Transpose, 3 x [Conv1d(k=5, same), BatchNorm1d (running
    statistics), ReLU, Dropout (inactive)], Transpose, Linear.  x (S,d) -> (S,d)."""
    h = x.t().unsqueeze(0)  # (1,d,S)
    for conv, bn in ((1, 2), (5, 6), (9, 10)):
        h = F.conv1d(h, sd[f"{prefix}.{conv}.weight"], sd[f"{prefix}.{conv}.bias"], padding="same")
        h = F.batch_norm(h, sd[f"{prefix}.{bn}.running_mean"], sd[f"{prefix}.{bn}.running_var"], sd[f"{prefix}.{bn}.weight"],
                         sd[f"{prefix}.{bn}.bias"], training=False, eps=1e-5)
        h = F.relu(h)
    return F.linear(h[0].t(), sd[f"{prefix}.14.weight"], sd[f"{prefix}.14.bias"])


def audio_prenet(sd, prefix: str, y: torch.Tensor) -> torch.Tensor:
    """{ar,nar}_audio_prenet in eval mode (valle.py:115-123): Linear(d,256), ReLU, Linear(256,256), ReLU, Linear(256,d)."""
    h = F.relu(F.linear(y, sd[f"{prefix}.0.weight"], sd[f"{prefix}.0.bias"]))
    h = F.relu(F.linear(h, sd[f"{prefix}.3.weight"], sd[f"{prefix}.3.bias"]))
    return F.linear(h, sd[f"{prefix}.6.weight"], sd[f"{prefix}.6.bias"])


def ar_mask(S: int, A: int) -> torch.Tensor:
    """valle.py:1010, 1018-1033: text rows see text only; audio rows see all text + causal audio."""
    m = torch.zeros(S + A, S + A, dtype=torch.bool)
    m[:S, S:] = True
    m[S:, S:] = torch.triu(torch.ones(A, A, dtype=torch.bool), diagonal=1)
    return m


# ----------------------------------------------------------------------------- sampling
def top_k_filter_(logits: torch.Tensor, top_k: int) -> torch.Tensor:
    """top_k_top_p_filtering with top_p=1.0, valle.py:1254-1260 — IN PLACE, ties at the k-th
    value are all kept (strict '<')."""
    if top_k > 0:
        top_k = min(max(top_k, 1), logits.size(-1))
        thr = torch.topk(logits, top_k)[0][..., -1, None]
        logits[logits < thr] = -float("inf")
    return logits


def topk_sampling(logits: torch.Tensor, top_k: int, temperature: float, exp_noise: Optional[torch.Tensor] = None):
    """valle.py:1287-1302.  With ``exp_noise`` (a (1,V) tensor of Exp(1) draws) the multinomial
    is evaluated as argmax(p / q), which is what torch.multinomial(p, 1) does on CPU with q
    drawn from the generator (aten/src/ATen/native/Sampling / SURVEY.md §9 v2)."""
    if temperature != 1.0:
        logits = logits / temperature
    logits = top_k_filter_(logits, top_k)
    p = F.softmax(logits, dim=-1)
    if exp_noise is None:
        return torch.multinomial(p, num_samples=1)
    return torch.argmax(p / exp_noise, dim=-1, keepdim=True)


# ----------------------------------------------------------------------------- the model view
class OracleModel:
    def __init__(self, sd: Dict[str, torch.Tensor], d_model: int, nhead: int, num_layers: int,
                 prefix_mode: int = 0, prepend_bos: bool = False, num_quantizers: int = 8,
                 nar_scale_factor: float = 1.0, norm_first: bool = True, add_prenet: bool = False):
        self.sd = sd
        self.norm_first = norm_first
        self.add_prenet = add_prenet
        self.d, self.nhead, self.L = d_model, nhead, num_layers
        self.dn = int(d_model * nar_scale_factor)
        self.nar_nhead = int(nhead * nar_scale_factor)
        self.nar_L = int(num_layers * nar_scale_factor)
        self.prefix_mode, self.prepend_bos, self.Q = prefix_mode, prepend_bos, num_quantizers
        self.ar_layers = [_Layer(sd, f"ar_decoder.layers.{i}", False) for i in range(self.L)]
        if self.Q > 1:
            self.nar_layers = [_Layer(sd, f"nar_decoder.layers.{i}", True) for i in range(self.nar_L)]

    # -- AR pieces ---------------------------------------------------------------------------
    def ar_text(self, text: torch.Tensor) -> torch.Tensor:  # (S,) -> (S,d); valle.py:995-997
        e = F.embedding(text, self.sd["ar_text_embedding.word_embeddings.weight"])
        if self.add_prenet:  # valle.py:996
            e = text_prenet(self.sd, "ar_text_prenet", e)
        return add_position(e, self.sd["ar_text_position.alpha"])

    def ar_audio(self, y: torch.Tensor, start: int = 0) -> torch.Tensor:  # valle.py:1013-1015
        e = F.embedding(y, self.sd["ar_audio_embedding.word_embeddings.weight"])
        if self.add_prenet:  # valle.py:1014
            e = audio_prenet(self.sd, "ar_audio_prenet", e)
        return add_position(e, self.sd["ar_audio_position.alpha"], start)

    def ar_stack(self, xy: torch.Tensor, mask) -> torch.Tensor:  # valle.py:1035-1038
        x = xy
        for L in self.ar_layers:
            x, _ = encoder_layer(L, x, self.nhead, mask, None, self.norm_first)
        return self.ar_final_norm(x)

    def ar_final_norm(self, x: torch.Tensor) -> torch.Tensor:  # norm=LayerNorm(d) if norm_first else None (valle.py:151)
        if not self.norm_first:
            return x
        return layer_norm(x, self.sd["ar_decoder.norm.weight"], self.sd["ar_decoder.norm.bias"])

    def ar_logits(self, h_last: torch.Tensor) -> torch.Tensor:  # valle.py:1039
        return F.linear(h_last, self.sd["ar_predict_layer.weight"])

    # -- NAR pieces --------------------------------------------------------------------------
    def nar_stack(self, xy: torch.Tensor, stage: int) -> torch.Tensor:  # valle.py:1125-1127
        e = self.sd[f"nar_stage_embeddings.{stage}.word_embeddings.weight"]
        x = xy
        for L in self.nar_layers:
            x, _ = encoder_layer(L, x, self.nar_nhead, None, e, self.norm_first)
        if not self.norm_first:  # valle.py:242-246
            return x
        g = lambda n: self.sd[f"nar_decoder.norm.{n}"]
        return ada_layer_norm(x, e, g("project_layer.weight"), g("project_layer.bias"), g("norm.weight"), g("norm.bias"))

    def nar(self, text: torch.Tensor, text_len: int, prompts: torch.Tensor, y: torch.Tensor,
            enroll_x_lens, trace: Optional[dict] = None, position_before_prenet: bool = False) -> List[torch.Tensor]:
        """valle.py:1059-1134.  text (S,), prompts (P,Q), y (P+T,) = prompt cb0 + AR tokens.
        Returns the Q-1 NAR code rows, each (T,).  position_before_prenet: the order VALLE.continual uses in
        prefix mode 0 (valle.py:1193-1194), the reverse of every other call site."""
        sd, P = self.sd, prompts.shape[0]
        y_emb = F.embedding(y, sd["nar_audio_embeddings.0.word_embeddings.weight"]).clone()
        if self.prefix_mode in (2, 4):  # valle.py:1068-1079
            enrolled_len = int(enroll_x_lens.max().item())
            text = torch.cat([text[:1], text[enrolled_len - 1:]])
            text_len = text_len - (enrolled_len - 2)
        x = F.embedding(text, sd["nar_text_embedding.word_embeddings.weight"])
        if self.add_prenet:  # valle.py:1082
            x = text_prenet(sd, "nar_text_prenet", x)
        x = add_position(x, sd["nar_text_position.alpha"])
        codes = []
        if self.prefix_mode != 0:  # valle.py:1110-1113
            for j in range(1, self.Q):
                y_emb[:P] += F.embedding(prompts[:, j], sd[f"nar_audio_embeddings.{j}.word_embeddings.weight"])
        for i in range(self.Q - 1):
            if not self.add_prenet:
                y_pos = add_position(y_emb, sd["nar_audio_position.alpha"])
            elif position_before_prenet:
                y_pos = audio_prenet(sd, "nar_audio_prenet", add_position(y_emb, sd["nar_audio_position.alpha"]))
            else:  # valle.py:1092-1093, 1121-1122
                y_pos = add_position(audio_prenet(sd, "nar_audio_prenet", y_emb), sd["nar_audio_position.alpha"])
            xy = torch.cat([x, y_pos], dim=0)
            h = self.nar_stack(xy, i)
            logits = F.linear(h[text_len + P:], sd[f"nar_predict_layers.{i}.weight"])
            if trace is not None:
                trace.setdefault("nar_logits", []).append(logits.clone())
            samples = torch.argmax(logits, dim=-1)
            codes.append(samples)
            if i < self.Q - 2:
                emb = sd[f"nar_audio_embeddings.{i + 1}.word_embeddings.weight"]
                if self.prefix_mode == 0:  # valle.py:1104-1108
                    y_emb[:P] += F.embedding(prompts[:, i + 1], emb)
                y_emb[P:] += F.embedding(samples, emb)
        return codes


def _stop(logits, samples, n_generated: int, S: int) -> bool:
    """valle.py:1044-1048."""
    return bool(
        torch.argmax(logits, dim=-1)[0] == NUM_AUDIO_TOKENS
        or samples[0, 0] == NUM_AUDIO_TOKENS
        or n_generated > S * 16
    )


@torch.no_grad()
def inference_faithful(m: OracleModel, x, x_lens, y, enroll_x_lens=None, top_k: int = -100,
                       temperature: float = 1.0, exp_noise: Optional[torch.Tensor] = None,
                       max_new_tokens: Optional[int] = None, trace: Optional[dict] = None,
                       skip_nar: bool = False) -> torch.Tensor:
    """No-cache restatement of VALLE.inference (valle.py:986-1137).  ``exp_noise`` (steps,V):
    row i feeds the multinomial of forward pass i; None -> torch.multinomial on the global RNG,
    exactly like the reference.  ``max_new_tokens`` truncates the AR loop (bench sampling only).
    """
    assert x.ndim == 2 and x_lens.ndim == 1 and y.ndim == 3 and y.shape[0] == 1
    assert torch.all(x_lens > 0)
    text = x[0]
    S = int(x_lens.max())
    X = m.ar_text(text)
    prompts = y[0]
    P = prompts.shape[0]
    yy = prompts[:, 0]
    if m.prepend_bos:
        yy = F.pad(yy, (1, 0), value=NUM_AUDIO_TOKENS + 1)
    bos = int(m.prepend_bos)
    step = 0
    while True:
        xy = torch.cat([X, m.ar_audio(yy)], dim=0)
        h = m.ar_stack(xy, ar_mask(S, yy.shape[0]))
        logits = m.ar_logits(h[-1:])
        if trace is not None:
            trace.setdefault("ar_logits", []).append(logits[0].clone())
        noise = None if exp_noise is None else exp_noise[step : step + 1]
        samples = topk_sampling(logits, top_k, temperature, noise)
        n_gen = yy.shape[0] - P - bos
        if _stop(logits, samples, yy.shape[0] - P, S) or (max_new_tokens is not None and n_gen >= max_new_tokens):
            if yy.shape[0] == P and max_new_tokens is None:  # valle.py:1049-1052
                raise SyntaxError("well trained model shouldn't reach here.")
            break
        yy = torch.cat([yy, samples[0]])
        step += 1
    codes = [yy[P + bos:]]
    if m.Q == 1 or skip_nar:
        return torch.stack(codes, dim=-1).unsqueeze(0)
    codes += m.nar(text, S, prompts, yy[bos:], enroll_x_lens, trace)
    return torch.stack(codes, dim=-1).unsqueeze(0)


# ----------------------------------------------------------------------------- cached variant
class ArCache:
    """KV-cached AR decode: prefill over [text | prompt] with the reference mask, then one row
    per step.  Hidden states of earlier rows never change, so this computes the
    same function as the no-cache loop."""

    def __init__(self, m: OracleModel):
        self.m = m
        self.k: List[torch.Tensor] = []
        self.v: List[torch.Tensor] = []

    def prefill(self, text: torch.Tensor, yy: torch.Tensor) -> torch.Tensor:
        m = self.m
        S = text.shape[0]
        self.S, self.n_audio = S, yy.shape[0]
        x = torch.cat([m.ar_text(text), m.ar_audio(yy)], dim=0)
        mask = ar_mask(S, yy.shape[0])
        self.k, self.v = [], []
        for L in m.ar_layers:
            x, (k, v) = encoder_layer(L, x, m.nhead, mask, None, m.norm_first)
            self.k.append(k)
            self.v.append(v)
        self.last_h = m.ar_final_norm(x[-1:])  # input of the predict layer (tests craft EOS rows from it)
        return m.ar_logits(self.last_h)

    def step(self, token: torch.Tensor) -> torch.Tensor:
        """token: (1,) int64 — the audio token appended at audio position ``n_audio``."""
        m = self.m
        x = m.ar_audio(token, start=self.n_audio)  # (1,d)
        self.n_audio += 1
        d, H = m.d, m.nhead
        hd = d // H
        for li, L in enumerate(m.ar_layers):
            hN = L.norm(0, x, None) if m.norm_first else x
            qkv = F.linear(hN, L.in_w, L.in_b)
            q, k, v = qkv.chunk(3, dim=-1)
            q = q.reshape(1, H, hd).transpose(0, 1)
            self.k[li] = torch.cat([self.k[li], k.reshape(1, H, hd).transpose(0, 1)], dim=1)
            self.v[li] = torch.cat([self.v[li], v.reshape(1, H, hd).transpose(0, 1)], dim=1)
            s = torch.matmul(q, self.k[li].transpose(1, 2)) / math.sqrt(hd)
            a = torch.matmul(F.softmax(s, dim=-1), self.v[li])  # (H,1,hd)
            a = a.transpose(0, 1).reshape(1, d)
            x = x + F.linear(a, L.out_w, L.out_b)
            if m.norm_first:
                x = x + F.linear(F.relu(F.linear(L.norm(1, x, None), L.w1, L.b1)), L.w2, L.b2)
            else:
                x = L.norm(0, x, None)
                x = L.norm(1, x + F.linear(F.relu(F.linear(x, L.w1, L.b1)), L.w2, L.b2), None)
        self.last_h = m.ar_final_norm(x)
        return m.ar_logits(self.last_h)


@torch.no_grad()
def inference_cached(m: OracleModel, x, x_lens, y, enroll_x_lens=None, top_k: int = -100,
                     temperature: float = 1.0, exp_noise: Optional[torch.Tensor] = None,
                     trace: Optional[dict] = None, forced: Optional[torch.Tensor] = None,
                     skip_nar: bool = False) -> torch.Tensor:
    """Same contract as ``inference_faithful`` with a KV cache.  ``forced`` (T,) teacher-forces
    the AR tokens (the sample is still drawn and recorded in ``trace['ar_samples']``)."""
    text = x[0]
    S = int(x_lens.max())
    prompts = y[0]
    P = prompts.shape[0]
    yy = prompts[:, 0]
    if m.prepend_bos:
        yy = F.pad(yy, (1, 0), value=NUM_AUDIO_TOKENS + 1)
    bos = int(m.prepend_bos)
    cache = ArCache(m)
    logits = cache.prefill(text, yy)
    step = 0
    while True:
        if trace is not None:
            trace.setdefault("ar_logits", []).append(logits[0].clone())
            trace.setdefault("ar_hidden", []).append(cache.last_h[0].clone())
        noise = None if exp_noise is None else exp_noise[step : step + 1]
        samples = topk_sampling(logits, top_k, temperature, noise)
        if trace is not None:
            trace.setdefault("ar_samples", []).append(int(samples[0, 0]))
        if forced is not None:
            if step >= forced.shape[0]:
                break
            samples = forced[step].reshape(1, 1)
        elif _stop(logits, samples, yy.shape[0] - P, S):
            if yy.shape[0] == P:  # valle.py:1049-1052
                raise SyntaxError("well trained model shouldn't reach here.")
            break
        yy = torch.cat([yy, samples[0]])
        step += 1
        if forced is None and (yy.shape[0] - P) > S * 16:
            break  # the next pass could only stop (valle.py:1047); skip computing it
        logits = cache.step(samples[0])
    codes = [yy[P + bos:]]
    if m.Q == 1 or skip_nar:
        return torch.stack(codes, dim=-1).unsqueeze(0)
    codes += m.nar(text, S, prompts, yy[bos:], enroll_x_lens, trace)
    return torch.stack(codes, dim=-1).unsqueeze(0)


@torch.no_grad()
def continual(m: OracleModel, x, x_lens, y) -> torch.Tensor:
    """VALLE.continual (valle.py:1139-1238): no AR pass — the first half of y (at most 225 frames) is the
    prompt, codebook 0 of the rest is kept, codebooks 1..7 of the rest are predicted by the NAR stages.
    The prefix_mode 2/4 text trim is NOT applied here (the reference does not)."""
    assert x.ndim == 2 and x_lens.ndim == 1 and y.ndim == 3 and y.shape[0] == 1
    assert torch.all(x_lens > 0)
    assert m.Q == 8
    text = x[0]
    S = int(x_lens.max())
    prefix_len = min(int(y.shape[1] * 0.5), 3 * 75)
    prompts = y[0, :prefix_len]
    codes = [y[0, prefix_len:, 0]]
    saved = m.prefix_mode
    try:
        m.prefix_mode = 0 if saved == 0 else 1  # the NAR body only distinguishes mode 0 from the rest here
        codes += m.nar(text, S, prompts, y[0, :, 0], None, position_before_prenet=(saved == 0))
    finally:
        m.prefix_mode = saved
    return torch.stack(codes, dim=-1).unsqueeze(0)


# ----------------------------------------------------------------------------- VALL-F (cross-attention variant)
# /root/reference/valle/models/valle.py:566-710 (VALLF.inference) over the reference's TransformerDecoderLayer
# (valle/modules/transformer.py:409-601) inside torch's nn.TransformerDecoder (valle.py:61-66, 141-151).
#
# Parity status of THIS section: pinned to the reference's own layer code under a RESTATED container.  The reference
# pins torch==1.13.1 (README.md:31), whose nn.TransformerDecoder.forward is a plain loop over the layers followed by the
# optional final norm.  The image's torch 2.10 nn.TransformerDecoder.forward inspects `tgt` (seq-len / causal-mask
# detection) and raises AttributeError on the (tensor, stage_embedding) tuples the reference passes, so the unmodified
# VALLF.inference does not run here.  oracle/ref_harness.py therefore replaces that one container method with the
# torch-1.13.1 loop when it builds a VALL-F reference model; every layer, embedding, mask and sampling line is the
# reference's.  tests/golden/vallf_*.npz are generated that way.


def cross_attention(x, mem, in_w, in_b, out_w, out_b, nhead: int, key_padding_mask: Optional[torch.Tensor] = None):
    """MultiheadAttention.forward(query=x, key=mem, value=mem) -> F.multi_head_attention_forward with k is v, q is not k
    (torch/nn/functional.py `_in_projection_packed`: q from rows [0,d) of the packed weight, one linear for [k|v] from
    rows [d,3d)).  x (N,d), mem (S,d); key_padding_mask (S,) bool, True = padded (valle.py:603, 631)."""
    N, d = x.shape
    S = mem.shape[0]
    hd = d // nhead
    q = F.linear(x, in_w[:d], in_b[:d])
    kv = F.linear(mem, in_w[d:], in_b[d:])
    k, v = kv[:, :d], kv[:, d:]
    q = q.reshape(N, nhead, hd).transpose(0, 1).unsqueeze(0)
    k = k.reshape(S, nhead, hd).transpose(0, 1).unsqueeze(0)
    v = v.reshape(S, nhead, hd).transpose(0, 1).unsqueeze(0)
    am = None
    if key_padding_mask is not None:  # merged into a float mask (1,H,1,S), functional.py key_padding_mask handling
        am = torch.zeros(S, dtype=x.dtype).masked_fill_(key_padding_mask, float("-inf")).view(1, 1, 1, S).expand(1, nhead, 1, S)
    o = F.scaled_dot_product_attention(q, k, v, am, 0.0, False)
    o = o.squeeze(0).transpose(0, 1).reshape(N, d)
    return F.linear(o, out_w, out_b)


class _DecLayer:
    """Weights of one TransformerDecoderLayer (transformer.py:412-500): self_attn, multihead_attn, FFN, norm1-3."""

    def __init__(self, sd, prefix, adaptive):
        g = lambda n: sd[f"{prefix}.{n}"]
        self.in_w, self.in_b = g("self_attn.in_proj_weight"), g("self_attn.in_proj_bias")
        self.out_w, self.out_b = g("self_attn.out_proj.weight"), g("self_attn.out_proj.bias")
        self.cin_w, self.cin_b = g("multihead_attn.in_proj_weight"), g("multihead_attn.in_proj_bias")
        self.cout_w, self.cout_b = g("multihead_attn.out_proj.weight"), g("multihead_attn.out_proj.bias")
        self.w1, self.b1 = g("linear1.weight"), g("linear1.bias")
        self.w2, self.b2 = g("linear2.weight"), g("linear2.bias")
        self.adaptive = adaptive
        if adaptive:
            self.n = [(g(f"{n}.project_layer.weight"), g(f"{n}.project_layer.bias"), g(f"{n}.norm.weight"), g(f"{n}.norm.bias"))
                      for n in ("norm1", "norm2", "norm3")]
        else:
            self.n = [(g(f"{n}.weight"), g(f"{n}.bias")) for n in ("norm1", "norm2", "norm3")]

    norm = _Layer.norm


def decoder_layer(L: _DecLayer, x, mem, nhead, tgt_mask, mem_pad, stage_emb=None, norm_first: bool = True):
    """TransformerDecoderLayer.forward: pre-norm branch transformer.py:536-546, post-norm branch 547-560."""
    ff = lambda h: F.linear(F.relu(F.linear(h, L.w1, L.b1)), L.w2, L.b2)
    if norm_first:
        x = x + self_attention(L.norm(0, x, stage_emb), L.in_w, L.in_b, L.out_w, L.out_b, nhead, tgt_mask)[0]
        x = x + cross_attention(L.norm(1, x, stage_emb), mem, L.cin_w, L.cin_b, L.cout_w, L.cout_b, nhead, mem_pad)
        return x + ff(L.norm(2, x, stage_emb))
    x = L.norm(0, x + self_attention(x, L.in_w, L.in_b, L.out_w, L.out_b, nhead, tgt_mask)[0], stage_emb)
    x = L.norm(1, x + cross_attention(x, mem, L.cin_w, L.cin_b, L.cout_w, L.cout_b, nhead, mem_pad), stage_emb)
    return L.norm(2, x + ff(x), stage_emb)


class OracleModelF(OracleModel):
    """VALLF: same embeddings / prenets / heads as VALLE (valle.py:54-279), decoder stacks with cross-attention."""

    def __init__(self, sd, d_model, nhead, num_layers, **kw):
        OracleModel.__init__(self, sd, d_model, nhead, num_layers, **kw)  # (the encoder-layer keys are a subset of the decoder layer's)
        self.ar_layers = [_DecLayer(sd, f"ar_decoder.layers.{i}", False) for i in range(self.L)]
        if self.Q > 1:
            self.nar_layers = [_DecLayer(sd, f"nar_decoder.layers.{i}", True) for i in range(self.nar_L)]

    def ar_stack(self, y_pos, mem, mem_pad):  # valle.py:626-632
        A = y_pos.shape[0]
        tgt_mask = torch.triu(torch.ones(A, A, dtype=torch.bool), diagonal=1)  # valle.py:619-624
        x = y_pos
        for L in self.ar_layers:
            x = decoder_layer(L, x, mem, self.nhead, tgt_mask, mem_pad, None, self.norm_first)
        return self.ar_final_norm(x)

    def nar_stack(self, y_pos, mem, stage: int):  # valle.py:682-688 (no masks)
        e = self.sd[f"nar_stage_embeddings.{stage}.word_embeddings.weight"]
        x = y_pos
        for L in self.nar_layers:
            x = decoder_layer(L, x, mem, self.nar_nhead, None, None, e, self.norm_first)
        if not self.norm_first:
            return x
        g = lambda n: self.sd[f"nar_decoder.norm.{n}"]
        return ada_layer_norm(x, e, g("project_layer.weight"), g("project_layer.bias"), g("norm.weight"), g("norm.bias"))

    def nar(self, text, prompts, y, enroll_x_lens, trace: Optional[dict] = None):  # valle.py:650-708
        sd, P = self.sd, prompts.shape[0]
        y_emb = F.embedding(y, sd["nar_audio_embeddings.0.word_embeddings.weight"]).clone()
        if self.prefix_mode in (2, 4):  # valle.py:653-662
            enrolled_len = int(enroll_x_lens.max().item())
            text = torch.cat([text[:1], text[enrolled_len - 1:]])
        x = F.embedding(text, sd["nar_text_embedding.word_embeddings.weight"])
        if self.add_prenet:
            x = text_prenet(sd, "nar_text_prenet", x)
        x = add_position(x, sd["nar_text_position.alpha"])
        if self.prefix_mode != 0:  # valle.py:668-672
            for j in range(1, self.Q):
                y_emb[:P] += F.embedding(prompts[:, j], sd[f"nar_audio_embeddings.{j}.word_embeddings.weight"])
        codes = []
        for i in range(self.Q - 1):
            y_pos = audio_prenet(sd, "nar_audio_prenet", y_emb) if self.add_prenet else y_emb
            y_pos = add_position(y_pos, sd["nar_audio_position.alpha"])
            h = self.nar_stack(y_pos, x, i)
            logits = F.linear(h[P:], sd[f"nar_predict_layers.{i}.weight"])
            if trace is not None:
                trace.setdefault("nar_logits", []).append(logits.clone())
            samples = torch.argmax(logits, dim=-1)
            codes.append(samples)
            if i < 6:  # valle.py:698-704 (literal 6; the zip over Q-1 layers ends the loop first when Q < 8)
                emb = sd[f"nar_audio_embeddings.{i + 1}.word_embeddings.weight"]
                if self.prefix_mode == 0:
                    y_emb[:P] += F.embedding(prompts[:, i + 1], emb)
                y_emb[P:] += F.embedding(samples, emb)
        return codes


@torch.no_grad()
def inference_faithful(m: OracleModel, x, x_lens, y, enroll_x_lens=None, top_k: int = -100,
                       temperature: float = 1.0, exp_noise: Optional[torch.Tensor] = None,
                       max_new_tokens: Optional[int] = None, trace: Optional[dict] = None,
                       skip_nar: bool = False) -> torch.Tensor:
    """No-cache restatement of VALLE.inference (valle.py:986-1137).  ``exp_noise`` (steps,V):
    row i feeds the multinomial of forward pass i; None -> torch.multinomial on the global RNG,
    exactly like the reference.  ``max_new_tokens`` truncates the AR loop (bench sampling only).
    """
    assert x.ndim == 2 and x_lens.ndim == 1 and y.ndim == 3 and y.shape[0] == 1
    assert torch.all(x_lens > 0)
    text = x[0]
    S = int(x_lens.max())
    X = m.ar_text(text)
    prompts = y[0]
    P = prompts.shape[0]
    yy = prompts[:, 0]
    if m.prepend_bos:
        yy = F.pad(yy, (1, 0), value=NUM_AUDIO_TOKENS + 1)
    bos = int(m.prepend_bos)
    step = 0
    while True:
        xy = torch.cat([X, m.ar_audio(yy)], dim=0)
        h = m.ar_stack(xy, ar_mask(S, yy.shape[0]))
        logits = m.ar_logits(h[-1:])
        if trace is not None:
            trace.setdefault("ar_logits", []).append(logits[0].clone())
        noise = None if exp_noise is None else exp_noise[step : step + 1]
        samples = topk_sampling(logits, top_k, temperature, noise)
        n_gen = yy.shape[0] - P - bos
        if _stop(logits, samples, yy.shape[0] - P, S) or (max_new_tokens is not None and n_gen >= max_new_tokens):
            if yy.shape[0] == P and max_new_tokens is None:  # valle.py:1049-1052
                raise SyntaxError("well trained model shouldn't reach here.")
            break
        yy = torch.cat([yy, samples[0]])
        step += 1
    codes = [yy[P + bos:]]
    if m.Q == 1 or skip_nar:
        return torch.stack(codes, dim=-1).unsqueeze(0)
    codes += m.nar(text, S, prompts, yy[bos:], enroll_x_lens, trace)
    return torch.stack(codes, dim=-1).unsqueeze(0)


# ----------------------------------------------------------------------------- cached variant
class ArCache:
    """KV-cached AR decode: prefill over [text | prompt] with the reference mask, then one row
    per step.  Hidden states of earlier rows never change, so this computes the
    same function as the no-cache loop."""

    def __init__(self, m: OracleModel):
        self.m = m
        self.k: List[torch.Tensor] = []
        self.v: List[torch.Tensor] = []

    def prefill(self, text: torch.Tensor, yy: torch.Tensor) -> torch.Tensor:
        m = self.m
        S = text.shape[0]
        self.S, self.n_audio = S, yy.shape[0]
        x = torch.cat([m.ar_text(text), m.ar_audio(yy)], dim=0)
        mask = ar_mask(S, yy.shape[0])
        self.k, self.v = [], []
        for L in m.ar_layers:
            x, (k, v) = encoder_layer(L, x, m.nhead, mask, None, m.norm_first)
            self.k.append(k)
            self.v.append(v)
        self.last_h = m.ar_final_norm(x[-1:])  # input of the predict layer (tests craft EOS rows from it)
        return m.ar_logits(self.last_h)

    def step(self, token: torch.Tensor) -> torch.Tensor:
        """token: (1,) int64 — the audio token appended at audio position ``n_audio``."""
        m = self.m
        x = m.ar_audio(token, start=self.n_audio)  # (1,d)
        self.n_audio += 1
        d, H = m.d, m.nhead
        hd = d // H
        for li, L in enumerate(m.ar_layers):
            hN = L.norm(0, x, None) if m.norm_first else x
            qkv = F.linear(hN, L.in_w, L.in_b)
            q, k, v = qkv.chunk(3, dim=-1)
            q = q.reshape(1, H, hd).transpose(0, 1)
            self.k[li] = torch.cat([self.k[li], k.reshape(1, H, hd).transpose(0, 1)], dim=1)
            self.v[li] = torch.cat([self.v[li], v.reshape(1, H, hd).transpose(0, 1)], dim=1)
            s = torch.matmul(q, self.k[li].transpose(1, 2)) / math.sqrt(hd)
            a = torch.matmul(F.softmax(s, dim=-1), self.v[li])  # (H,1,hd)
            a = a.transpose(0, 1).reshape(1, d)
            x = x + F.linear(a, L.out_w, L.out_b)
            if m.norm_first:
                x = x + F.linear(F.relu(F.linear(L.norm(1, x, None), L.w1, L.b1)), L.w2, L.b2)
            else:
                x = L.norm(0, x, None)
                x = L.norm(1, x + F.linear(F.relu(F.linear(x, L.w1, L.b1)), L.w2, L.b2), None)
        self.last_h = m.ar_final_norm(x)
        return m.ar_logits(self.last_h)


@torch.no_grad()
def inference_cached(m: OracleModel, x, x_lens, y, enroll_x_lens=None, top_k: int = -100,
                     temperature: float = 1.0, exp_noise: Optional[torch.Tensor] = None,
                     trace: Optional[dict] = None, forced: Optional[torch.Tensor] = None,
                     skip_nar: bool = False) -> torch.Tensor:
    """Same contract as ``inference_faithful`` with a KV cache.  ``forced`` (T,) teacher-forces
    the AR tokens (the sample is still drawn and recorded in ``trace['ar_samples']``)."""
    text = x[0]
    S = int(x_lens.max())
    prompts = y[0]
    P = prompts.shape[0]
    yy = prompts[:, 0]
    if m.prepend_bos:
        yy = F.pad(yy, (1, 0), value=NUM_AUDIO_TOKENS + 1)
    bos = int(m.prepend_bos)
    cache = ArCache(m)
    logits = cache.prefill(text, yy)
    step = 0
    while True:
        if trace is not None:
            trace.setdefault("ar_logits", []).append(logits[0].clone())
            trace.setdefault("ar_hidden", []).append(cache.last_h[0].clone())
        noise = None if exp_noise is None else exp_noise[step : step + 1]
        samples = topk_sampling(logits, top_k, temperature, noise)
        if trace is not None:
            trace.setdefault("ar_samples", []).append(int(samples[0, 0]))
        if forced is not None:
            if step >= forced.shape[0]:
                break
            samples = forced[step].reshape(1, 1)
        elif _stop(logits, samples, yy.shape[0] - P, S):
            if yy.shape[0] == P:  # valle.py:1049-1052
                raise SyntaxError("well trained model shouldn't reach here.")
            break
        yy = torch.cat([yy, samples[0]])
        step += 1
        if forced is None and (yy.shape[0] - P) > S * 16:
            break  # the next pass could only stop (valle.py:1047); skip computing it
        logits = cache.step(samples[0])
    codes = [yy[P + bos:]]
    if m.Q == 1 or skip_nar:
        return torch.stack(codes, dim=-1).unsqueeze(0)
    codes += m.nar(text, S, prompts, yy[bos:], enroll_x_lens, trace)
    return torch.stack(codes, dim=-1).unsqueeze(0)


@torch.no_grad()
def continual(m: OracleModel, x, x_lens, y) -> torch.Tensor:
    """VALLE.continual (valle.py:1139-1238): no AR pass — the first half of y (at most 225 frames) is the
    prompt, codebook 0 of the rest is kept, codebooks 1..7 of the rest are predicted by the NAR stages.
    The prefix_mode 2/4 text trim is NOT applied here (the reference does not)."""
    assert x.ndim == 2 and x_lens.ndim == 1 and y.ndim == 3 and y.shape[0] == 1
    assert torch.all(x_lens > 0)
    assert m.Q == 8
    text = x[0]
    S = int(x_lens.max())
    prefix_len = min(int(y.shape[1] * 0.5), 3 * 75)
    prompts = y[0, :prefix_len]
    codes = [y[0, prefix_len:, 0]]
    saved = m.prefix_mode
    try:
        m.prefix_mode = 0 if saved == 0 else 1  # the NAR body only distinguishes mode 0 from the rest here
        codes += m.nar(text, S, prompts, y[0, :, 0], None, position_before_prenet=(saved == 0))
    finally:
        m.prefix_mode = saved
    return torch.stack(codes, dim=-1).unsqueeze(0)


# ----------------------------------------------------------------------------- VALL-F (cross-attention variant)
# /root/reference/valle/models/valle.py:566-710 (VALLF.inference) over the reference's TransformerDecoderLayer
# (valle/modules/transformer.py:409-601) inside torch's nn.TransformerDecoder (valle.py:61-66, 141-151).
#
# Parity status of THIS section: pinned to the reference's own layer code under a RESTATED container.  The reference
# pins torch==1.13.1 (README.md:31), whose nn.TransformerDecoder.forward is a plain loop over the layers followed by the
# optional final norm.  The image's torch 2.10 nn.TransformerDecoder.forward inspects `tgt` (seq-len / causal-mask
# detection) and raises AttributeError on the (tensor, stage_embedding) tuples the reference passes, so the unmodified
# VALLF.inference does not run here.  oracle/ref_harness.py therefore replaces that one container method with the
# torch-1.13.1 loop when it builds a VALL-F reference model; every layer, embedding, mask and sampling line is the
# reference's.  tests/golden/vallf_*.npz are generated that way.


def cross_attention(x, mem, in_w, in_b, out_w, out_b, nhead: int, key_padding_mask: Optional[torch.Tensor] = None):
    """MultiheadAttention.forward(query=x, key=mem, value=mem) -> F.multi_head_attention_forward with k is v, q is not k
    (torch/nn/functional.py `_in_projection_packed`: q from rows [0,d) of the packed weight, one linear for [k|v] from
    rows [d,3d)).  x (N,d), mem (S,d); key_padding_mask (S,) bool, True = padded (valle.py:603, 631)."""
    N, d = x.shape
    S = mem.shape[0]
    hd = d // nhead
    q = F.linear(x, in_w[:d], in_b[:d])
    kv = F.linear(mem, in_w[d:], in_b[d:])
    k, v = kv[:, :d], kv[:, d:]
    q = q.reshape(N, nhead, hd).transpose(0, 1).unsqueeze(0)
    k = k.reshape(S, nhead, hd).transpose(0, 1).unsqueeze(0)
    v = v.reshape(S, nhead, hd).transpose(0, 1).unsqueeze(0)
    am = None
    if key_padding_mask is not None:  # merged into a float mask (1,H,1,S), functional.py key_padding_mask handling
        am = torch.zeros(S, dtype=x.dtype).masked_fill_(key_padding_mask, float("-inf")).view(1, 1, 1, S).expand(1, nhead, 1, S)
    o = F.scaled_dot_product_attention(q, k, v, am, 0.0, False)
    o = o.squeeze(0).transpose(0, 1).reshape(N, d)
    return F.linear(o, out_w, out_b)


class _DecLayer:
    """Weights of one TransformerDecoderLayer (transformer.py:412-500): self_attn, multihead_attn, FFN, norm1-3."""

    def __init__(self, sd, prefix, adaptive):
        g = lambda n: sd[f"{prefix}.{n}"]
        self.in_w, self.in_b = g("self_attn.in_proj_weight"), g("self_attn.in_proj_bias")
        self.out_w, self.out_b = g("self_attn.out_proj.weight"), g("self_attn.out_proj.bias")
        self.cin_w, self.cin_b = g("multihead_attn.in_proj_weight"), g("multihead_attn.in_proj_bias")
        self.cout_w, self.cout_b = g("multihead_attn.out_proj.weight"), g("multihead_attn.out_proj.bias")
        self.w1, self.b1 = g("linear1.weight"), g("linear1.bias")
        self.w2, self.b2 = g("linear2.weight"), g("linear2.bias")
        self.adaptive = adaptive
        if adaptive:
            self.n = [(g(f"{n}.project_layer.weight"), g(f"{n}.project_layer.bias"), g(f"{n}.norm.weight"), g(f"{n}.norm.bias"))
                      for n in ("norm1", "norm2", "norm3")]
        else:
            self.n = [(g(f"{n}.weight"), g(f"{n}.bias")) for n in ("norm1", "norm2", "norm3")]

    norm = _Layer.norm


def decoder_layer(L: _DecLayer, x, mem, nhead, tgt_mask, mem_pad, stage_emb=None, norm_first: bool = True):
    """TransformerDecoderLayer.forward: pre-norm branch transformer.py:536-546, post-norm branch 547-560."""
    ff = lambda h: F.linear(F.relu(F.linear(h, L.w1, L.b1)), L.w2, L.b2)
    if norm_first:
        x = x + self_attention(L.norm(0, x, stage_emb), L.in_w, L.in_b, L.out_w, L.out_b, nhead, tgt_mask)[0]
        x = x + cross_attention(L.norm(1, x, stage_emb), mem, L.cin_w, L.cin_b, L.cout_w, L.cout_b, nhead, mem_pad)
        return x + ff(L.norm(2, x, stage_emb))
    x = L.norm(0, x + self_attention(x, L.in_w, L.in_b, L.out_w, L.out_b, nhead, tgt_mask)[0], stage_emb)
    x = L.norm(1, x + cross_attention(x, mem, L.cin_w, L.cin_b, L.cout_w, L.cout_b, nhead, mem_pad), stage_emb)
    return L.norm(2, x + ff(x), stage_emb)


class OracleModelF(OracleModel):
    """VALLF: same embeddings / prenets / heads as VALLE (valle.py:54-279), decoder stacks with cross-attention."""

    def __init__(self, sd, d_model, nhead, num_layers, **kw):
        OracleModel.__init__(self, sd, d_model, nhead, num_layers, **kw)  # (the encoder-layer keys are a subset of the decoder layer's)
        self.ar_layers = [_DecLayer(sd, f"ar_decoder.layers.{i}", False) for i in range(self.L)]
        if self.Q > 1:
            self.nar_layers = [_DecLayer(sd, f"nar_decoder.layers.{i}", True) for i in range(self.nar_L)]

    def ar_stack(self, y_pos, mem, mem_pad):  # valle.py:626-632
        A = y_pos.shape[0]
        tgt_mask = torch.triu(torch.ones(A, A, dtype=torch.bool), diagonal=1)  # valle.py:619-624
        x = y_pos
        for L in self.ar_layers:
            x = decoder_layer(L, x, mem, self.nhead, tgt_mask, mem_pad, None, self.norm_first)
        return self.ar_final_norm(x)

    def nar_stack(self, y_pos, mem, stage: int):  # valle.py:682-688 (no masks)
        e = self.sd[f"nar_stage_embeddings.{stage}.word_embeddings.weight"]
        x = y_pos
        for L in self.nar_layers:
            x = decoder_layer(L, x, mem, self.nar_nhead, None, None, e, self.norm_first)
        if not self.norm_first:
            return x
        g = lambda n: self.sd[f"nar_decoder.norm.{n}"]
        return ada_layer_norm(x, e, g("project_layer.weight"), g("project_layer.bias"), g("norm.weight"), g("norm.bias"))

    def nar(self, text, prompts, y, enroll_x_lens, trace: Optional[dict] = None):  # valle.py:650-708
        sd, P = self.sd, prompts.shape[0]
        y_emb = F.embedding(y, sd["nar_audio_embeddings.0.word_embeddings.weight"]).clone()
        if self.prefix_mode in (2, 4):  # valle.py:653-662
            enrolled_len = int(enroll_x_lens.max().item())
            text = torch.cat([text[:1], text[enrolled_len - 1:]])
        x = F.embedding(text, sd["nar_text_embedding.word_embeddings.weight"])
        if self.add_prenet:
            x = text_prenet(sd, "nar_text_prenet", x)
        x = add_position(x, sd["nar_text_position.alpha"])
        if self.prefix_mode != 0:  # valle.py:668-672
            for j in range(1, self.Q):
                y_emb[:P] += F.embedding(prompts[:, j], sd[f"nar_audio_embeddings.{j}.word_embeddings.weight"])
        codes = []
        for i in range(self.Q - 1):
            y_pos = audio_prenet(sd, "nar_audio_prenet", y_emb) if self.add_prenet else y_emb
            y_pos = add_position(y_pos, sd["nar_audio_position.alpha"])
            h = self.nar_stack(y_pos, x, i)
            logits = F.linear(h[P:], sd[f"nar_predict_layers.{i}.weight"])
            if trace is not None:
                trace.setdefault("nar_logits", []).append(logits.clone())
            samples = torch.argmax(logits, dim=-1)
            codes.append(samples)
            if i < 6:  # valle.py:698-704 (literal 6; the zip over Q-1 layers ends the loop first when Q < 8)
                emb = sd[f"nar_audio_embeddings.{i + 1}.word_embeddings.weight"]
                if self.prefix_mode == 0:
                    y_emb[:P] += F.embedding(prompts[:, i + 1], emb)
                y_emb[P:] += F.embedding(samples, emb)
        return codes


class _NoLayers(dict):
    """state_dict view for OracleModel.__init__: VALL-F layers have other keys, so the encoder-layer lookups are deferred."""

    def __init__(self, sd):
        super().__init__(sd)

    def __missing__(self, k):
        return None


@torch.no_grad()
def inference_f(m: OracleModelF, x, x_lens, y, enroll_x_lens=None, top_k: int = -100, temperature: float = 1.0,
                exp_noise: Optional[torch.Tensor] = None, trace: Optional[dict] = None,
                forced: Optional[torch.Tensor] = None) -> torch.Tensor:
    """VALLF.inference (valle.py:566-710), no cache: the audio sequence is recomputed every pass against the text memory.
    Text positions >= x_lens are masked in the AR cross-attention (memory_key_padding_mask, valle.py:603, 631) and NOT
    masked in the NAR stages (valle.py:687)."""
    assert x.ndim == 2 and x_lens.ndim == 1 and y.ndim == 3 and y.shape[0] == 1
    assert torch.all(x_lens > 0)
    text = x[0]
    S_stop = int(x_lens.max())
    mem = m.ar_text(text)
    mem_pad = torch.arange(text.shape[0]) >= x_lens[0]
    prompts = y[0]
    P = prompts.shape[0]
    yy = prompts[:, 0]
    if m.prepend_bos:
        yy = F.pad(yy, (1, 0), value=NUM_AUDIO_TOKENS + 1)
    bos = int(m.prepend_bos)
    step = 0
    while True:
        h = m.ar_stack(m.ar_audio(yy), mem, mem_pad)
        logits = m.ar_logits(h[-1:])
        if trace is not None:
            trace.setdefault("ar_logits", []).append(logits[0].clone())
        noise = None if exp_noise is None else exp_noise[step : step + 1]
        samples = topk_sampling(logits, top_k, temperature, noise)
        if forced is not None:
            if step >= forced.shape[0]:
                break
            samples = forced[step].reshape(1, 1)
        elif _stop(logits, samples, yy.shape[0] - P, S_stop):
            if yy.shape[0] == P:  # valle.py:641-644 (compares with prompts.shape[1]: never true with prepend_bos)
                raise SyntaxError("well trained model shouldn't reach here.")
            break
        yy = torch.cat([yy, samples[0]])
        step += 1
    codes = [yy[P + bos:]]
    if m.Q == 1:
        return torch.stack(codes, dim=-1).unsqueeze(0)
    codes += m.nar(text, prompts, yy[bos:], enroll_x_lens, trace)
    return torch.stack(codes, dim=-1).unsqueeze(0)
