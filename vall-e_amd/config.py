"""Model configuration for the VALL-E inference hot path.

Mirrors the flag names of the reference's ``add_model_arguments``
(/root/reference/valle/models/__init__.py:18-95) and the constants of
/root/reference/valle/models/macros.py:2-5, so that a ``params`` object built for the
reference's ``get_model`` can be handed to ours unchanged.
"""
from __future__ import annotations

import argparse
from dataclasses import dataclass

# valle/models/macros.py:2-5
NUM_TEXT_TOKENS = 512
NUM_AUDIO_TOKENS = 1024  # EnCodec RVQ bins; id 1024 = EOS/PAD, id 1025 = BOS (valle.py:88-93)

# valle/modules/embedding.py:64 — initial sine table length (auto-extends in the reference)
SINE_TABLE_LEN = 4000
LN_EPS = 1e-5  # valle/modules/transformer.py:26,197


def str2bool(v):
    """Same accepted spellings as icefall.utils.str2bool (used by the reference's flags)."""
    if isinstance(v, bool):
        return v
    if v.lower() in ("yes", "true", "t", "y", "1"):
        return True
    if v.lower() in ("no", "false", "f", "n", "0"):
        return False
    raise argparse.ArgumentTypeError("Boolean value expected.")


@dataclass
class ModelConfig:
    """Attribute names are the argparse dests of the reference (models/__init__.py:18-95)."""

    model_name: str = "VALL-E"
    decoder_dim: int = 1024
    nhead: int = 16
    num_decoder_layers: int = 12
    scale_factor: float = 1.0
    norm_first: bool = True
    add_prenet: bool = False
    prefix_mode: int = 0
    share_embedding: bool = True
    prepend_bos: bool = False
    num_quantizers: int = 8

    @property
    def is_vallf(self) -> bool:
        """models/__init__.py:99: the cross-attention variant is selected by name."""
        return self.model_name.lower() in ("vall-f", "vallf")

    # derived (valle.py:83, 231-241)
    @property
    def nar_dim(self) -> int:
        return int(self.decoder_dim * self.scale_factor)

    @property
    def nar_nhead(self) -> int:
        return int(self.nhead * self.scale_factor)

    @property
    def nar_layers(self) -> int:
        return int(self.num_decoder_layers * self.scale_factor)

    @classmethod
    def from_params(cls, params) -> "ModelConfig":
        """Accepts an argparse.Namespace / AttributeDict / dict like the reference's get_model."""
        get = params.get if isinstance(params, dict) else lambda k, d=None: getattr(params, k, d)
        kw = {}
        for f in cls.__dataclass_fields__:
            v = get(f, None)
            if v is not None:
                kw[f] = v
        return cls(**kw)


def add_model_arguments(parser: argparse.ArgumentParser):
    """Same flags, defaults and dests as the reference (models/__init__.py:18-95).
    ``--scaling-xformers`` is accepted for CLI compatibility; it only affects the reference's
    debug mel-Transformer, which is outside the hot path."""
    parser.add_argument("--model-name", type=str, default="VALL-E")
    parser.add_argument("--decoder-dim", type=int, default=1024)
    parser.add_argument("--nhead", type=int, default=16)
    parser.add_argument("--num-decoder-layers", type=int, default=12)
    parser.add_argument("--scale-factor", type=float, default=1.0)
    parser.add_argument("--norm-first", type=str2bool, default=True)
    parser.add_argument("--add-prenet", type=str2bool, default=False)
    parser.add_argument("--prefix-mode", type=int, default=0)
    parser.add_argument("--share-embedding", type=str2bool, default=True)
    parser.add_argument("--prepend-bos", type=str2bool, default=False)
    parser.add_argument("--num-quantizers", type=int, default=8)
    parser.add_argument("--scaling-xformers", type=str2bool, default=False)
