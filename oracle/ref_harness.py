"""Import the UNMODIFIED reference model code as a checker — build container only.

Test infrastructure.  ``/root/reference`` is absent on the GPU box, so everything here must be
called behind ``reference_available()``; the GPU-side tests use the committed fixtures under
``tests/golden/`` instead.

The reference cannot be imported plainly (``valle/__init__.py`` pulls in icefall / lhotse /
encodec, which are not installed).  None of those is on the arithmetic path of
``VALLE.inference``; they are replaced by in-process stubs:
  * ``icefall.utils``: ``make_pad_mask`` (semantics per valle.py:804-806 usage), ``AttributeDict``,
    ``str2bool``
  * ``torchmetrics.classification``: ``MulticlassAccuracy`` / ``BinaryAccuracy`` constructors only
    (valle.py:157-163)
  * a stub package ``valle`` / ``valle.data`` exposing ``valle.data.input_strategies.PromptedFeatures``
    so that the real ``valle/data/__init__.py`` is not executed.
Everything that computes — valle/models/valle.py, valle/modules/{transformer,activation,
embedding,scaling}.py — is loaded from the reference tree as is.  No bytecode is written.
"""
from __future__ import annotations

import importlib
import importlib.util
import os
import sys
import types

REF_ROOT = os.environ.get("VALLE_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REF_ROOT, "valle", "models", "valle.py"))


def _stub_module(name: str, **attrs) -> types.ModuleType:
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _install_stubs():
    import torch

    class AttributeDict(dict):
        def __getattr__(self, k):
            if k in self:
                return self[k]
            raise AttributeError(k)

        def __setattr__(self, k, v):
            self[k] = v

    def str2bool(v):
        if isinstance(v, bool):
            return v
        return str(v).lower() in ("yes", "true", "t", "y", "1")

    def make_pad_mask(lengths, max_len: int = 0):
        n = max(max_len, int(lengths.max()))
        return torch.arange(n, device=lengths.device)[None, :] >= lengths[:, None]

    if "icefall" not in sys.modules:
        ice = _stub_module("icefall")
        ice.utils = _stub_module("icefall.utils", AttributeDict=AttributeDict, str2bool=str2bool,
                                 make_pad_mask=make_pad_mask)

    class _Metric(torch.nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

        def forward(self, *a, **k):
            return torch.tensor(0.0)

    if "torchmetrics" not in sys.modules:
        tm = _stub_module("torchmetrics")
        tm.classification = _stub_module("torchmetrics.classification", MulticlassAccuracy=_Metric,
                                         BinaryAccuracy=_Metric)

    class PromptedFeatures:
        def __init__(self, prompts, features):
            self.prompts, self.features = prompts, features

        @property
        def data(self):
            return (self.prompts, self.features)

    # stub *package* shells whose __path__ points into the reference tree, so that submodules
    # (valle.models, valle.modules) load from the reference but valle/__init__.py and
    # valle/data/__init__.py (lhotse/encodec imports) never run.
    pkg = _stub_module("valle")
    pkg.__path__ = [os.path.join(REF_ROOT, "valle")]
    data = _stub_module("valle.data")
    data.__path__ = []
    data.input_strategies = _stub_module("valle.data.input_strategies", PromptedFeatures=PromptedFeatures)
    pkg.data = data
    # valle/models/visualizer.py imports matplotlib (present) — leave it alone.


_loaded = None


def load_reference():
    """Returns the reference's ``valle.models`` module (get_model, VALLE, ...)."""
    global _loaded
    if _loaded is not None:
        return _loaded
    if not reference_available():
        raise RuntimeError(f"reference tree not found at {REF_ROOT}")
    sys.dont_write_bytecode = True  # the reference mount must stay untouched
    os.environ.setdefault("MPLBACKEND", "Agg")
    _install_stubs()
    _loaded = importlib.import_module("valle.models")
    return _loaded


def _torch113_decoder_forward(self, tgt, memory, tgt_mask=None, memory_mask=None, tgt_key_padding_mask=None,
                              memory_key_padding_mask=None):
    """nn.TransformerDecoder.forward as of torch 1.13.1, the version the reference pins (README.md:31): a plain loop over
    the layers and the optional final norm.  Restated here (third-party, 6 lines): the installed torch 2.10 version inspects
    ``tgt`` for sequence length / causality and raises AttributeError on the (tensor, stage_embedding) tuple that
    VALLF.inference passes (valle.py:626-632, 682-688), so the unmodified VALL-F path cannot run in this image."""
    output = tgt
    for mod in self.layers:
        output = mod(output, memory, tgt_mask=tgt_mask, memory_mask=memory_mask,
                     tgt_key_padding_mask=tgt_key_padding_mask, memory_key_padding_mask=memory_key_padding_mask)
    if self.norm is not None:
        output = self.norm(output)
    return output


def _use_torch113_decoder_container(model):
    import types

    for dec in (model.ar_decoder, getattr(model, "nar_decoder", None)):
        if dec is not None:
            dec.forward = types.MethodType(_torch113_decoder_forward, dec)


def build_reference_model(cfg, state_dict):
    """cfg: valle_amd.config.ModelConfig.  Builds the reference VALLE via its own get_model
    (models/__init__.py:98-136), loads ``state_dict`` strictly (bin/infer.py:139-143), eval()."""
    models = load_reference()
    from icefall.utils import AttributeDict

    params = AttributeDict(
        model_name=cfg.model_name, decoder_dim=cfg.decoder_dim, nhead=cfg.nhead,
        num_decoder_layers=cfg.num_decoder_layers, scale_factor=cfg.scale_factor,
        norm_first=cfg.norm_first, add_prenet=cfg.add_prenet, prefix_mode=cfg.prefix_mode,
        share_embedding=cfg.share_embedding, prepend_bos=cfg.prepend_bos,
        num_quantizers=cfg.num_quantizers,
    )
    model = models.get_model(params)
    if cfg.is_vallf:
        _use_torch113_decoder_container(model)
    missing, unexpected = model.load_state_dict(state_dict, strict=True)
    assert not missing and not unexpected
    model.eval()
    return model
