"""MI355X-native decode of Qwen3-TTS 12 Hz speech-tokenizer codes (RVQ codes -> 24 kHz waveform).

The directory name carries a hyphen, so import it as `qwen3_tts_amd` (the one-file alias module at
the repository root) or with `importlib.import_module("qwen3-tts_amd")`.
"""
from .config import (DecoderConfig, FRONTEND_MEAN_EMBEDDING, FRONTEND_SPLIT_RVQ, flagship_config,
                     sibling_default_config, tiny_config)
from .weights import param_specs, random_codes, synthetic_weights, validate_weights

__all__ = ["DecoderConfig", "FRONTEND_MEAN_EMBEDDING", "FRONTEND_SPLIT_RVQ", "flagship_config",
           "sibling_default_config", "tiny_config", "param_specs", "random_codes", "synthetic_weights",
           "validate_weights"]
