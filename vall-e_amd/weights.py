"""Checkpoint key layout of the reference's VALLE module and a synthetic weight generator.

The key names/shapes are what ``VALLF.__init__`` registers for the decoder-only variant
(/root/reference/valle/models/valle.py:54-279, built with TransformerEncoder(-Layer) at
valle.py:748-760; layer internals at /root/reference/valle/modules/transformer.py:181-258 and
/root/reference/valle/modules/activation.py:71-160).  ``bin/infer.py:139-143`` loads
``checkpoint["model"]`` with ``strict=True``, so the loader must accept exactly this set.

No trained checkpoint ships with the reference (README.md:23), so tests and the benchmark use
*name-seeded* synthetic weights: each tensor is drawn from its own ``torch.Generator`` seeded
with ``crc32(key) ^ master_seed``.  The same bytes are therefore reproducible in the build
container (where they are loaded into the imported reference to make golden vectors) and on
the GPU box (where they are loaded into the engine) without shipping a 1.4 GB file.
"""
from __future__ import annotations

import math
import zlib
from collections import OrderedDict
from typing import Dict, Tuple

import torch

from .config import NUM_AUDIO_TOKENS, NUM_TEXT_TOKENS, ModelConfig


def _encoder_keys(prefix: str, d: int, layers: int, adaptive: bool, final_norm: bool = True,
                  cross: bool = False) -> "OrderedDict[str, Tuple[int, ...]]":
    """``cross``: the VALL-F stacks are TransformerDecoderLayers (modules/transformer.py:412-500): a second attention
    module ``multihead_attn`` over the text memory and a third norm, registered in that order."""
    out: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    for i in range(layers):
        p = f"{prefix}.layers.{i}"
        out[f"{p}.self_attn.in_proj_weight"] = (3 * d, d)
        out[f"{p}.self_attn.in_proj_bias"] = (3 * d,)
        out[f"{p}.self_attn.out_proj.weight"] = (d, d)
        out[f"{p}.self_attn.out_proj.bias"] = (d,)
        if cross:
            out[f"{p}.multihead_attn.in_proj_weight"] = (3 * d, d)
            out[f"{p}.multihead_attn.in_proj_bias"] = (3 * d,)
            out[f"{p}.multihead_attn.out_proj.weight"] = (d, d)
            out[f"{p}.multihead_attn.out_proj.bias"] = (d,)
        out[f"{p}.linear1.weight"] = (4 * d, d)
        out[f"{p}.linear1.bias"] = (4 * d,)
        out[f"{p}.linear2.weight"] = (d, 4 * d)
        out[f"{p}.linear2.bias"] = (d,)
        for n in ("norm1", "norm2", "norm3") if cross else ("norm1", "norm2"):
            if adaptive:
                out[f"{p}.{n}.project_layer.weight"] = (2 * d, d)
                out[f"{p}.{n}.project_layer.bias"] = (2 * d,)
                out[f"{p}.{n}.norm.weight"] = (d,)
                out[f"{p}.{n}.norm.bias"] = (d,)
            else:
                out[f"{p}.{n}.weight"] = (d,)
                out[f"{p}.{n}.bias"] = (d,)
    if not final_norm:  # norm_first=False: the encoder has no final norm (valle.py:151, 242-246)
        return out
    if adaptive:
        out[f"{prefix}.norm.project_layer.weight"] = (2 * d, d)
        out[f"{prefix}.norm.project_layer.bias"] = (2 * d,)
        out[f"{prefix}.norm.norm.weight"] = (d,)
        out[f"{prefix}.norm.norm.bias"] = (d,)
    else:
        out[f"{prefix}.norm.weight"] = (d,)
        out[f"{prefix}.norm.bias"] = (d,)
    return out


PRENET_HIDDEN = 256  # valle.py:116-122


def _prenet_keys(prefix: str, d: int) -> "OrderedDict[str, Tuple[int, ...]]":
    """{ar,nar}_text_prenet = Sequential(Transpose, [Conv1d(d,d,5,same), BatchNorm1d, ReLU, Dropout] x 3, Transpose,
    Linear(d,d)); {ar,nar}_audio_prenet = Linear(d,256), ReLU, Dropout, Linear(256,256), ReLU, Dropout, Linear(256,d)
    (valle.py:96-123, 181-213).  Module indices are the Sequential positions."""
    out: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    t = f"{prefix}_text_prenet"
    for conv, bn in ((1, 2), (5, 6), (9, 10)):
        out[f"{t}.{conv}.weight"] = (d, d, 5)
        out[f"{t}.{conv}.bias"] = (d,)
        out[f"{t}.{bn}.weight"] = (d,)
        out[f"{t}.{bn}.bias"] = (d,)
        out[f"{t}.{bn}.running_mean"] = (d,)
        out[f"{t}.{bn}.running_var"] = (d,)
        out[f"{t}.{bn}.num_batches_tracked"] = ()
    out[f"{t}.14.weight"] = (d, d)
    out[f"{t}.14.bias"] = (d,)
    a = f"{prefix}_audio_prenet"
    out[f"{a}.0.weight"] = (PRENET_HIDDEN, d)
    out[f"{a}.0.bias"] = (PRENET_HIDDEN,)
    out[f"{a}.3.weight"] = (PRENET_HIDDEN, PRENET_HIDDEN)
    out[f"{a}.3.bias"] = (PRENET_HIDDEN,)
    out[f"{a}.6.weight"] = (d, PRENET_HIDDEN)
    out[f"{a}.6.bias"] = (d,)
    return out


def expected_keys(cfg: ModelConfig) -> "OrderedDict[str, Tuple[int, ...]]":
    """state_dict keys -> shapes for VALLE, and for VALLF when ``cfg.model_name`` names it (valle.py:85-259).  372 entries at L=12 / 8 quantizers with the defaults;
    post-norm models have no final encoder norms, add_prenet adds the four prenets.  Key ORDER follows the
    reference's module registration order."""
    d, dn = cfg.decoder_dim, cfg.nar_dim
    q = cfg.num_quantizers
    k: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    k["ar_text_embedding.word_embeddings.weight"] = (NUM_TEXT_TOKENS, d)
    k["nar_text_embedding.word_embeddings.weight"] = (NUM_TEXT_TOKENS, dn)
    k["ar_audio_embedding.word_embeddings.weight"] = (NUM_AUDIO_TOKENS + 1 + int(cfg.prepend_bos), d)
    if cfg.add_prenet:
        k.update(_prenet_keys("ar", d))
    k["ar_text_position.alpha"] = (1,)
    k["ar_audio_position.alpha"] = (1,)
    cross = cfg.is_vallf
    k.update(_encoder_keys("ar_decoder", d, cfg.num_decoder_layers, adaptive=False, final_norm=cfg.norm_first, cross=cross))
    k["ar_predict_layer.weight"] = (NUM_AUDIO_TOKENS + 1, d)
    if q > 1:
        k["nar_audio_embeddings.0.word_embeddings.weight"] = (NUM_AUDIO_TOKENS + 1, dn)
        for j in range(1, q):
            k[f"nar_audio_embeddings.{j}.word_embeddings.weight"] = (NUM_AUDIO_TOKENS, dn)
        if cfg.add_prenet:
            k.update(_prenet_keys("nar", dn))
        k["nar_text_position.alpha"] = (1,)
        k["nar_audio_position.alpha"] = (1,)
        k.update(_encoder_keys("nar_decoder", dn, cfg.nar_layers, adaptive=True, final_norm=cfg.norm_first, cross=cross))
        for j in range(q - 1):
            k[f"nar_predict_layers.{j}.weight"] = (NUM_AUDIO_TOKENS, dn)
        for j in range(q - 1):
            k[f"nar_stage_embeddings.{j}.word_embeddings.weight"] = (1, dn)
    return k


def tied_keys(cfg: ModelConfig) -> Dict[str, str]:
    """predict-layer key -> embedding key it aliases (valle.py:261-271)."""
    if not cfg.share_embedding or cfg.num_quantizers <= 2:
        return {}
    return {
        f"nar_predict_layers.{j}.weight": f"nar_audio_embeddings.{j + 2}.word_embeddings.weight"
        for j in range(cfg.num_quantizers - 2)
    }


def _gen(key: str, seed: int) -> torch.Generator:
    g = torch.Generator(device="cpu")
    g.manual_seed((zlib.crc32(key.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFFFFFFFFFF)
    return g


def synthetic_tensor(key: str, shape: Tuple[int, ...], seed: int = 0) -> torch.Tensor:
    """Distribution follows the reference's initialisers so activations stay O(1):
    in_proj xavier-uniform (activation.py:175-177), other Linear U(+-1/sqrt(fan_in)) (torch
    default), embeddings N(0,1) (nn.Embedding default).  LayerNorm affine and the attention
    biases are perturbed away from the reference's 1/0 init (transformer.py:52-55,
    activation.py:183-185) so that those code paths are exercised numerically."""
    g = _gen(key, seed)
    if key.endswith(".alpha"):
        if key.startswith("ar_text"):
            return torch.tensor([0.9375])
        if key.startswith("ar_audio"):
            return torch.tensor([1.0625])
        return torch.ones(1)  # NAR alphas are frozen at 1.0 (valle.py:218-229)
    if "word_embeddings" in key:
        return torch.randn(shape, generator=g)
    if "_prenet." in key:
        if key.endswith("num_batches_tracked"):
            return torch.tensor(0, dtype=torch.int64)
        if key.endswith("running_mean"):
            return 0.1 * torch.randn(shape, generator=g)
        if key.endswith("running_var"):
            return 0.5 + torch.rand(shape, generator=g)
        if len(shape) == 3:  # Conv1d: U(+-1/sqrt(fan_in)), fan_in = in_channels * kernel
            return (torch.rand(shape, generator=g) * 2 - 1) / math.sqrt(shape[1] * shape[2])
        if len(shape) == 1 and ".weight" in key and key.split(".")[-2] in ("2", "6", "10") and "text_prenet" in key:
            return 1.0 + 0.05 * torch.randn(shape, generator=g)  # BatchNorm gamma
    if key.endswith("in_proj_weight"):
        bound = math.sqrt(6.0 / (shape[0] + shape[1]))
        return (torch.rand(shape, generator=g) * 2 - 1) * bound
    if key.endswith("in_proj_bias") or key.endswith("out_proj.bias"):
        return torch.randn(shape, generator=g) * 0.02
    is_ln = (".norm1." in key or ".norm2." in key or ".norm3." in key or "_decoder.norm." in key) and "project_layer" not in key
    if is_ln:
        if key.endswith("weight"):
            return 1.0 + 0.02 * torch.randn(shape, generator=g)
        return 0.02 * torch.randn(shape, generator=g)
    if "project_layer" in key:
        # AdaLN projection: scale part centred on 1 so that w * LN(x) keeps unit scale
        fan_in = shape[-1]
        bound = 1.0 / math.sqrt(fan_in)
        t = (torch.rand(shape, generator=g) * 2 - 1) * bound
        if key.endswith("bias"):
            t[: shape[0] // 2] += 1.0
        return t
    if key.endswith("weight") and len(shape) == 2:  # linear1/2, out_proj, predict layers
        bound = 1.0 / math.sqrt(shape[1])
        return (torch.rand(shape, generator=g) * 2 - 1) * bound
    if key.endswith("bias"):
        # nn.Linear bias: U(+-1/sqrt(fan_in)); fan_in is not recoverable from the bias shape,
        # use the model width as a stand-in
        return (torch.rand(shape, generator=g) * 2 - 1) * 0.03
    raise KeyError(key)


def synthetic_state_dict(cfg: ModelConfig, seed: int = 0, zero_eos: bool = True) -> "OrderedDict[str, torch.Tensor]":
    """Full fp32 state_dict.  ``zero_eos`` zeroes the EOS row of ``ar_predict_layer`` so a
    random-init model never stops early (its logit is then exactly 0, valle.py:153-155)."""
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    ties = tied_keys(cfg)
    for key, shape in expected_keys(cfg).items():
        if key in ties:
            continue
        sd[key] = synthetic_tensor(key, shape, seed).contiguous()
    for pk, ek in ties.items():
        sd[pk] = sd[ek]
    if zero_eos:
        sd["ar_predict_layer.weight"][NUM_AUDIO_TOKENS].zero_()
    # keep the reference's key order
    return OrderedDict((k, sd[k]) for k in expected_keys(cfg))


def synthetic_inputs(S: int, P: int, num_quantizers: int = 8, seed: int = 1):
    """Synthetic phoneme ids / prompt codes: ids uniform in [3,100), x[0]=<bos>=1, x[-1]=<eos>=2
    (collation.py:46-54); prompt codes uniform in [0,1024)."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    x = torch.randint(3, 100, (1, S), generator=g, dtype=torch.int64)
    x[0, 0] = 1
    x[0, -1] = 2
    x_lens = torch.tensor([S], dtype=torch.int32)
    y = torch.randint(0, NUM_AUDIO_TOKENS, (1, P, num_quantizers), generator=g, dtype=torch.int64)
    return x, x_lens, y


def sine_table(length: int, dim: int) -> torch.Tensor:
    """The reference's fp32 table, computed with the same torch CPU ops so that it is
    bit-identical (embedding.py:75-88): pe[p,2i]=sin(p*w_i), pe[p,2i+1]=cos(p*w_i),
    w_i = exp(2i * -(ln 1e4 / dim))."""
    position = torch.arange(0, length, dtype=torch.float32).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, dim, 2, dtype=torch.float32) * -(math.log(10000.0) / dim))
    pe = torch.zeros(length, dim)
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe
