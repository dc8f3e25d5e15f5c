"""CPU: the host-side mirror of the reference's model factory (models/__init__.py:98-136) - no GPU, no library calls."""
import pytest
import torch

from valle_amd.config import ModelConfig
from valle_amd.models import VALLE, VALLF, get_model
from valle_amd.weights import expected_keys, synthetic_state_dict


def _params(name, **kw):
    p = dict(model_name=name, decoder_dim=64, nhead=2, num_decoder_layers=2, scale_factor=1.0, norm_first=True, add_prenet=False,
             prefix_mode=1, share_embedding=True, prepend_bos=False, num_quantizers=8)
    p.update(kw)
    return p


@pytest.mark.parametrize("name,cls", [("VALL-E", VALLE), ("valle", VALLE), ("VALL-F", VALLF), ("vallf", VALLF)])
def test_get_model_dispatches_on_model_name(name, cls):
    m = get_model(_params(name))
    assert type(m) is cls and m.cfg.is_vallf == (cls is VALLF)
    assert list(m.state_dict()) == list(expected_keys(m.cfg))


def test_debug_transformer_is_not_built():
    with pytest.raises(NotImplementedError):
        get_model(_params("Transformer"))


def test_vallf_state_dict_has_cross_attention_and_third_norm():
    cfg = ModelConfig(model_name="VALL-F", decoder_dim=64, nhead=2, num_decoder_layers=2)
    keys = expected_keys(cfg)
    assert keys["ar_decoder.layers.1.multihead_attn.in_proj_weight"] == (192, 64)
    assert keys["nar_decoder.layers.0.norm3.project_layer.weight"] == (128, 64)
    assert "ar_decoder.layers.0.multihead_attn.in_proj_weight" not in expected_keys(ModelConfig(decoder_dim=64, nhead=2, num_decoder_layers=2))


def test_vallf_strict_load_and_surface():
    m = get_model(_params("VALL-F"))
    sd = synthetic_state_dict(m.cfg, 3)
    m.load_state_dict(sd)
    bad = dict(sd)
    del bad["ar_decoder.layers.0.multihead_attn.out_proj.bias"]
    with pytest.raises(RuntimeError, match="VALLF"):
        m.load_state_dict(bad, strict=True)
    with pytest.raises(RuntimeError):  # a VALL-E checkpoint does not fit (bin/infer.py:139-143 loads strictly)
        m.load_state_dict(synthetic_state_dict(ModelConfig(decoder_dim=64, nhead=2, num_decoder_layers=2, prefix_mode=1), 3))
    with pytest.raises(AttributeError):
        m.continual(None, None, None)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m.inference(torch.zeros(1, 3, dtype=torch.int64) + 5, torch.tensor([3], dtype=torch.int32), torch.zeros(1, 4, 8, dtype=torch.int64), None)


def test_batched_path_refuses_the_variants_it_does_not_serve():
    for kw in (dict(norm_first=False), dict(add_prenet=True)):
        with pytest.raises(NotImplementedError):
            VALLE(64, 1, 2, max_batch=4, **kw)
    with pytest.raises(NotImplementedError):
        VALLF(64, 1, 2, max_batch=4)
    with pytest.raises(NotImplementedError):
        VALLF(64, 1, 2, precision="fp8nar")
