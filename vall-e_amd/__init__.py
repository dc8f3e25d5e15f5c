"""MI355X-native VALL-E inference engine (drop-in for ``valle.models.VALLE.inference``)."""
from .config import ModelConfig, add_model_arguments, NUM_AUDIO_TOKENS, NUM_TEXT_TOKENS  # noqa: F401

__all__ = ["ModelConfig", "add_model_arguments", "NUM_AUDIO_TOKENS", "NUM_TEXT_TOKENS"]
