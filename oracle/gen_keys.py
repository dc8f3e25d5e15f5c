"""Dump the reference's state_dict layout (key -> shape, dtype, and which keys share storage) for a set of
constructor options -> tests/golden/state_dict_layout.json.  Build-container only (imports /root/reference through
oracle/ref_harness.py).  Usage: PYTHONDONTWRITEBYTECODE=1 python -m oracle.gen_keys"""
from __future__ import annotations

import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

import valle_amd  # noqa: E402,F401
from valle_amd.config import ModelConfig  # noqa: E402
from oracle.ref_harness import load_reference  # noqa: E402

CASES = {
    "default": dict(),
    "baseline_cfg1": dict(decoder_dim=1024, nhead=16, num_decoder_layers=12, prefix_mode=1),
    "post_norm": dict(norm_first=False),
    "prenet": dict(add_prenet=True),
    "prenet_post_norm": dict(add_prenet=True, norm_first=False),
    "prepend_bos_q6": dict(prepend_bos=True, num_quantizers=6),
    "q1": dict(num_quantizers=1),
    "q2_unshared": dict(num_quantizers=2, share_embedding=False),
    "scale05": dict(decoder_dim=256, nhead=4, num_decoder_layers=4, scale_factor=0.5),
    # VALLF (valle.py:49-279 with the default decoder classes): layers gain multihead_attn.* and norm3.*
    "vallf": dict(model_name="VALL-F"),
    "vallf_post_norm_prenet_q6": dict(model_name="VALL-F", norm_first=False, add_prenet=True, num_quantizers=6, prepend_bos=True),
    "vallf_scale05": dict(model_name="VALL-F", decoder_dim=256, nhead=4, num_decoder_layers=4, scale_factor=0.5),
}

if __name__ == "__main__":
    models = load_reference()
    from icefall.utils import AttributeDict

    out = {}
    for name, kw in CASES.items():
        base = dict(decoder_dim=128, nhead=2, num_decoder_layers=2, prefix_mode=1)
        base.update(kw)
        cfg = ModelConfig(**base)
        params = AttributeDict(model_name=cfg.model_name, decoder_dim=cfg.decoder_dim, nhead=cfg.nhead,
                               num_decoder_layers=cfg.num_decoder_layers, scale_factor=cfg.scale_factor,
                               norm_first=cfg.norm_first, add_prenet=cfg.add_prenet, prefix_mode=cfg.prefix_mode,
                               share_embedding=cfg.share_embedding, prepend_bos=cfg.prepend_bos,
                               num_quantizers=cfg.num_quantizers)
        sd = models.get_model(params).state_dict()
        ptr = {}
        shared = []
        for k, v in sd.items():
            if v.numel() and v.data_ptr() in ptr:
                shared.append([k, ptr[v.data_ptr()]])
            elif v.numel():
                ptr[v.data_ptr()] = k
        out[name] = dict(cfg=base, keys=[[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in sd.items()],
                         shared=shared)
        print(name, len(sd), "keys,", len(shared), "tied")
    path = os.path.join(ROOT, "tests", "golden", "state_dict_layout.json")
    json.dump(out, open(path, "w"), indent=0)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")
