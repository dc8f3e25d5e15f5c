"""CPU: the C-ABI library builds, loads and exports every symbol include/vallex.h declares
(no compute call is made without a GPU), and the host-side mirror validates arguments like the
reference does."""
import os
import re

import pytest
import torch

from conftest import ROOT


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as ge

    ge.build()
    from valle_amd import engine

    return engine.load_library()


def test_header_symbols_exported(lib):
    from valle_amd import engine

    hdr = open(os.path.join(ROOT, "include", "vallex.h")).read()
    declared = sorted(set(re.findall(r"\b(vx_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 18
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in vallex.h but not exported"
    assert declared == engine.declared_symbols()  # the ctypes table covers the whole header


def test_struct_sizes_match_header(lib):
    from valle_amd.engine import VxConfig, VxDecodeParams
    import ctypes as C

    assert C.sizeof(VxConfig) == 16 * 4
    assert C.sizeof(VxDecodeParams) == 56


def test_error_reporting_without_gpu(lib):
    import ctypes as C
    from valle_amd.engine import VxConfig

    c = VxConfig()
    c.struct_size = 0  # wrong on purpose: rejected before any HIP call
    h = C.c_void_p()
    assert lib.vx_create(C.byref(c), C.byref(h)) == 1
    assert b"struct_size" in lib.vx_last_error()


def test_geometry_checks_without_gpu(lib):
    """Geometry is validated before any HIP call: head_dim 4..64 (powers of two) is accepted by the Python mirror, anything
    else and batched decode at head_dim != 64 are refused loudly on both sides of the C ABI."""
    import ctypes as C
    from valle_amd.engine import VxConfig
    from valle_amd.models import VALLE

    VALLE(64, 16, 4, norm_first=False, add_prenet=True)  # the reference's own test geometry (valle_test.py:93-95)
    with pytest.raises(NotImplementedError, match="head_dim"):
        VALLE(96, 8, 2)  # head_dim 12
    with pytest.raises(NotImplementedError, match="head_dim 64"):
        VALLE(64, 16, 4, max_batch=4)
    c = VxConfig()
    c.struct_size = C.sizeof(VxConfig)
    c.d_model, c.nhead, c.num_layers = 96, 8, 2
    c.nar_d_model, c.nar_nhead, c.nar_num_layers = 96, 8, 2
    c.num_quantizers, c.prefix_mode, c.precision, c.max_text, c.max_audio = 8, 1, 0, 16, 64
    h = C.c_void_p()
    assert lib.vx_create(C.byref(c), C.byref(h)) != 0
    assert b"head_dim" in lib.vx_last_error()


def test_wrapper_argument_checks_match_reference():
    from valle_amd.models import VALLE

    m = VALLE(128, 2, 2).eval()
    x = torch.randint(3, 50, (1, 6))
    y = torch.randint(0, 1024, (1, 10, 8))
    with pytest.raises(AssertionError):  # valle.py:986
        m.inference(x[0], torch.tensor([6]), y, None)
    with pytest.raises(AssertionError):  # valle.py:989 batch-1 only
        m.inference(x, torch.tensor([6]), y.repeat(2, 1, 1), None)
    with pytest.raises(AssertionError):  # valle.py:991
        m.inference(x, torch.tensor([0]), y, None)
    with pytest.raises(RuntimeError, match="no CPU fallback"):  # product path never falls back to CPU
        m.inference(x, torch.tensor([6], dtype=torch.int32), y, None)


def test_state_dict_strict_loading():
    from valle_amd.config import ModelConfig
    from valle_amd.models import VALLE
    from valle_amd.weights import expected_keys, synthetic_state_dict

    cfg = ModelConfig(decoder_dim=128, nhead=2, num_decoder_layers=2)
    sd = synthetic_state_dict(cfg, 3)
    m = VALLE(128, 2, 2)
    r = m.load_state_dict(sd, strict=True)
    assert not r.missing_keys and not r.unexpected_keys
    assert list(m.state_dict()) == list(expected_keys(cfg))
    bad = dict(sd)
    bad.pop("ar_predict_layer.weight")
    with pytest.raises(RuntimeError):
        m.load_state_dict(bad, strict=True)
    bad = dict(sd)
    bad["bogus"] = torch.zeros(1)
    with pytest.raises(RuntimeError):
        m.load_state_dict(bad, strict=True)


def test_key_table_is_372_entries_at_baseline_config():
    from valle_amd.config import ModelConfig
    from valle_amd.weights import expected_keys

    keys = expected_keys(ModelConfig())
    assert len(keys) == 372
    n = 0
    seen = set()
    for k, s in keys.items():
        if k.startswith("nar_predict_layers.") and int(k.split(".")[1]) < 6:
            continue  # tied to nar_audio_embeddings.{j+2}
        p = 1
        for v in s:
            p *= v
        n += p
    assert n == 367386628


def _layouts():
    import json

    return json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "state_dict_layout.json")))


@pytest.mark.parametrize("case", sorted(_layouts()))
def test_state_dict_layout_matches_reference(case):
    """tests/golden/state_dict_layout.json is the reference's own `get_model(params).state_dict()` (names, order, shapes,
    dtypes, tied storage) for each constructor option the build supports (oracle/gen_keys.py): the loader contract of
    bin/infer.py:139-143 (`load_state_dict(checkpoint["model"], strict=True)`)."""
    from valle_amd.config import ModelConfig
    from valle_amd.weights import expected_keys, synthetic_state_dict, tied_keys

    ref = _layouts()[case]
    cfg = ModelConfig(**ref["cfg"])
    want = expected_keys(cfg)
    assert [k for k, _, _ in ref["keys"]] == list(want)  # same names, same order
    for k, shape, dtype in ref["keys"]:
        assert tuple(shape) == tuple(want[k]), k
    sd = synthetic_state_dict(cfg, 0)
    for k, shape, dtype in ref["keys"]:
        assert str(sd[k].dtype).replace("torch.", "") == dtype, k
    assert sorted((a, b) for a, b in ref["shared"]) == sorted(tied_keys(cfg).items())  # valle.py:261-271
