"""CPU: the oracle restatement against the vectors produced by the unmodified reference
(oracle/gen_golden.py).  This is what pins the oracle; the GPU parity tests then compare the
HIP engine with the oracle and with the same fixtures."""
import os

import pytest
import torch

from conftest import Golden, golden_names
from oracle import valle_oracle as vo

SMALL = [n for n in golden_names() if not n.startswith(("cfg1", "cfg4"))]  # the full-size fixtures have tests of their own


@pytest.mark.parametrize("name", SMALL)
def test_cached_oracle_matches_reference(name):
    g = Golden(name)
    tr = {}
    codes = vo.inference_cached(g.oracle(), g.x, g.x_lens, g.y, g.enroll_x_lens, g.top_k, g.temperature,
                                g.exp_noise, trace=tr)
    assert torch.equal(codes, g.codes)  # integer output: bit-exact
    for step, ref in zip(g.ar_probe_steps, g.ar_probe_logits):
        assert (tr["ar_logits"][step] - ref).abs().max() <= 1e-5
    if g.nar_probe_logits is not None:
        for i, ref in enumerate(g.nar_probe_logits):
            assert (tr["nar_logits"][i][:8] - ref).abs().max() <= 2e-4, i


@pytest.mark.parametrize("name", [n for n in SMALL if n.startswith("tiny") or n == "cfg0_greedy"])
def test_faithful_oracle_matches_reference(name):
    g = Golden(name)
    codes = vo.inference_faithful(g.oracle(), g.x, g.x_lens, g.y, g.enroll_x_lens, g.top_k, g.temperature, g.exp_noise)
    assert torch.equal(codes, g.codes)


def test_multinomial_is_argmax_of_p_over_exponential():
    """torch.multinomial(p, 1) on CPU == argmax(p / q), q ~ Exp(1) from the same generator state
    (the identity the engine's host-supplied-noise sampling relies on)."""
    for seed in range(50):
        torch.manual_seed(seed)
        logits = torch.randn(1, 1025) * 3
        p = torch.softmax(vo.top_k_filter_(logits.clone(), 10), -1)
        torch.manual_seed(1000 + seed)
        a = torch.multinomial(p, 1)
        torch.manual_seed(1000 + seed)
        q = torch.empty(1, 1025).exponential_(1)
        assert int(a) == int(torch.argmax(p / q, -1))


def test_noise_regenerates_from_seed():
    g = Golden("cfg0_topk10")
    torch.manual_seed(g.sample_seed)
    q = torch.stack([torch.empty(1, 1025).exponential_(1)[0] for _ in range(g.n_pass)])
    assert torch.equal(q, g.exp_noise)


def test_topk_ties_kept_and_inplace():
    l = torch.tensor([[1.0, 3.0, 2.0, 2.0, 0.5]])
    out = vo.top_k_filter_(l, 2)
    assert out is l
    assert l.tolist() == [[float("-inf"), 3.0, 2.0, 2.0, float("-inf")]]


def test_natural_length_is_16S_plus_1():
    g = Golden("cfg0_greedy")
    assert g.codes.shape == (1, 16 * int(g.x_lens[0]) + 1, 8)


@pytest.mark.skipif(not os.path.isfile(os.path.join(os.path.dirname(__file__), "golden", "cfg1_topk10.npz")),
                    reason="cfg1 fixture not generated")
def test_cfg1_fixture_shape():
    g = Golden("cfg1_topk10")
    assert g.codes.shape == (1, 753, 8) and g.cfg.decoder_dim == 1024


@pytest.mark.parametrize("name", golden_names("continual"))
def test_continual_oracle_matches_reference(name):
    """VALLE.continual (valle.py:1139-1238), SURVEY.md §8(f) rank 1."""
    g = Golden(name)
    assert torch.equal(vo.continual(g.oracle(), g.x, g.x_lens, g.y), g.codes)


@pytest.mark.parametrize("name", golden_names("vallf"))
def test_vallf_oracle_matches_reference(name):
    """VALLF.inference (valle.py:566-710), SURVEY.md §8(f) rank 4.  The fixtures come from the reference's own layers under the
    torch-1.13.1 TransformerDecoder loop restated in oracle/ref_harness.py (see oracle/valle_oracle.py: the installed torch 2.10
    container rejects the reference's tuple inputs)."""
    g = Golden(name)
    assert g.cfg.is_vallf
    tr = {}
    codes = vo.inference_f(g.oracle(), g.x, g.x_lens, g.y, g.enroll_x_lens, g.top_k, g.temperature, g.exp_noise, trace=tr)
    assert torch.equal(codes, g.codes)
    for step, ref in zip(g.ar_probe_steps, g.ar_probe_logits):
        assert (tr["ar_logits"][step] - ref).abs().max() <= 1e-5
    if g.nar_probe_logits is not None:
        for i, ref in enumerate(g.nar_probe_logits):
            assert (tr["nar_logits"][i][:8] - ref).abs().max() <= 2e-4, i
