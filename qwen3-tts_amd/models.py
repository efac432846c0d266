"""Model manager: the reference's `TTSModelManager` (reference models.py:15-132) with the MI355X decode installed.

What it keeps from the reference: a cache of loaded models keyed by model type (models.py:19,34-35,59), the dtype map
(models.py:24-30) and the attention-implementation string (models.py:22), the `TTS_DEVICE` / `TTS_DTYPE` /
`TTS_USE_FLASH_ATTENTION` / `TTS_USE_WARMUP` / `TTS_DEFAULT_MODEL` environment names (config.py:12-16,72), the model keys
and their aliases (config.py:57-69), `load_model / get_model / load_default_models / is_loaded / get_loaded_models`
(models.py:32-66,108-128) and the warm-up generation (models.py:68-106).

What differs, on purpose:
  * **local directories only.**  The reference falls back to a hub NAME when the model directory is absent
    (config.py:23-54), which makes `from_pretrained` download it.  Here a model type resolves to `<env var>` or
    `<models_root>/<directory name>`, and anything that is not a directory on this machine raises FileNotFoundError before
    the factory is called.
  * after `from_pretrained` the tokenizer's decode is replaced by this package's (`install.install_decoder`: the one added
    line of the integration), and the warm-up also warms the HIP path (`Decoder.warmup`: code object, workspace, pinned
    staging buffers, and - with stream slots - the push plan and its graph).
  * `torch.compile` of the talker (models.py:51-57, off by default in the reference "causes CUDA errors") is not offered.

The `qwen-tts` package is an injected dependency: `factory` is `Qwen3TTSModel.from_pretrained` by default and is imported
only when a model is actually loaded, so this module imports (and its logic is tested) without it."""
from __future__ import annotations

import os
import time
from typing import Any, Callable, Dict, List, Mapping, Optional

import torch

from .checkpoint import require_local_dir

WARMUP_TEXT = "안녕하세요."                # the reference's warm-up sentence (models.py:12)

# model type -> (environment variable that overrides the path, directory name under the models root) (config.py:23-69)
MODEL_TABLE: Dict[str, tuple] = {
    "custom_voice_1.7b": ("MODEL_1_7B_CUSTOM_VOICE", "Qwen3-TTS-12Hz-1.7B-CustomVoice"),
    "voice_design_1.7b": ("MODEL_1_7B_VOICE_DESIGN", "Qwen3-TTS-12Hz-1.7B-VoiceDesign"),
    "base_1.7b": ("MODEL_1_7B_BASE", "Qwen3-TTS-12Hz-1.7B-Base"),
    "custom_voice_0.6b": ("MODEL_0_6B_CUSTOM_VOICE", "Qwen3-TTS-12Hz-0.6B-CustomVoice"),
    "base_0.6b": ("MODEL_0_6B_BASE", "Qwen3-TTS-12Hz-0.6B-Base"),
    # aliases for backward compatibility (config.py:65-68)
    "custom_voice": ("MODEL_1_7B_CUSTOM_VOICE", "Qwen3-TTS-12Hz-1.7B-CustomVoice"),
    "voice_design": ("MODEL_1_7B_VOICE_DESIGN", "Qwen3-TTS-12Hz-1.7B-VoiceDesign"),
    "base": ("MODEL_1_7B_BASE", "Qwen3-TTS-12Hz-1.7B-Base"),
}
DTYPE_MAP = {"bfloat16": torch.bfloat16, "float16": torch.float16, "float32": torch.float32}
# The decode's arithmetic per model dtype.  float16 has no kernel path of its own: it runs the fp32 decode (at least the
# precision the model asked for) rather than bf16 (fewer mantissa bits than fp16).
DECODE_PRECISION = {torch.bfloat16: "bf16", torch.float16: "f32", torch.float32: "f32"}


def _flag(env: Mapping[str, str], name: str, default: str = "false") -> bool:
    return env.get(name, default).lower() == "true"


class TTSModelManager:
    """factory(model_path, device_map=..., dtype=..., attn_implementation=...) -> model   (Qwen3TTSModel.from_pretrained)
    installer(model, precision=..., device=..., **decoder_kwargs) -> decoder            (install.install_decoder)"""

    def __init__(self, factory: Optional[Callable[..., Any]] = None, installer: Optional[Callable[..., Any]] = None,
                 env: Optional[Mapping[str, str]] = None, models_root: Optional[str] = None,
                 decoder_kwargs: Optional[Dict[str, Any]] = None, warm_batch: int = 1, warm_frames: int = 25):
        env = os.environ if env is None else env
        self._env = env
        self._factory = factory
        self._installer = installer
        self.models: Dict[str, Any] = {}
        self.decoders: Dict[str, Any] = {}
        self.device = env.get("TTS_DEVICE", "cuda:0")
        self.dtype = DTYPE_MAP.get(env.get("TTS_DTYPE", "bfloat16"), torch.bfloat16)      # unknown names: bfloat16, as models.py:30
        self.attn_impl = "flash_attention_2" if _flag(env, "TTS_USE_FLASH_ATTENTION") else "sdpa"
        self.use_warmup = _flag(env, "TTS_USE_WARMUP")
        self.default_model = env.get("TTS_DEFAULT_MODEL", "base_0.6b")
        self.models_root = models_root or env.get("TTS_MODELS_DIR") or os.path.join(os.getcwd(), "models")
        self.decoder_kwargs = dict(decoder_kwargs or {})
        self.warm_batch, self.warm_frames = int(warm_batch), int(warm_frames)
        self.sample_audio = env.get("TTS_WARMUP_AUDIO", os.path.join(os.getcwd(), "sample(1).mp3"))

    # ---- paths ---------------------------------------------------------------------------------------------------------
    def model_path(self, model_type: str) -> str:
        """The LOCAL directory of a model type: its environment variable when set, else `<models_root>/<name>`.
        Raises ValueError for an unknown type and FileNotFoundError when the path is not a directory here (a hub name such
        as "Qwen/Qwen3-TTS-12Hz-0.6B-Base" is such a path: nothing is ever downloaded)."""
        if model_type not in MODEL_TABLE:
            raise ValueError(f"Unknown model type: {model_type}. Available: {list(MODEL_TABLE.keys())}")
        var, name = MODEL_TABLE[model_type]
        return require_local_dir(self._env.get(var) or os.path.join(self.models_root, name), f"model '{model_type}' path")

    # ---- loading -------------------------------------------------------------------------------------------------------
    def load_model(self, model_type: str) -> Any:
        if model_type in self.models:
            return self.models[model_type]
        path = self.model_path(model_type)
        factory = self._factory
        if factory is None:
            try:
                from qwen_tts import Qwen3TTSModel
            except ImportError as e:
                raise ImportError("loading a model needs the `qwen-tts` package (Qwen3TTSModel.from_pretrained), or pass "
                                  "factory= to TTSModelManager") from e
            factory = Qwen3TTSModel.from_pretrained
        installer = self._installer
        if installer is None:
            from .install import install_decoder as installer
        print(f"Loading model: {model_type} from {path}...")
        model = factory(path, device_map=self.device, dtype=self.dtype, attn_implementation=self.attn_impl)
        decoder = installer(model, precision=DECODE_PRECISION[self.dtype], device=self.device, **self.decoder_kwargs)
        self.models[model_type] = model
        self.decoders[model_type] = decoder
        print(f"Model {model_type} loaded successfully!")
        if self.use_warmup:
            self._warmup_model(model, decoder, model_type)
        return model

    def _warmup_model(self, model: Any, decoder: Any, model_type: str) -> None:
        """First-use costs out of the first request: the decode's own (see Decoder.warmup), then one short generation as
        the reference runs it (models.py:68-106).  A failing warm-up never fails the load."""
        print(f"Warming up {model_type}...")
        t0 = time.time()
        try:
            if hasattr(decoder, "warmup"):
                decoder.warmup(batch=self.warm_batch, frames=self.warm_frames)
            if "base" in model_type:
                if not os.path.exists(self.sample_audio):
                    print(f"Skipping generation warmup: sample audio not found at {self.sample_audio}")
                else:
                    model.generate_voice_clone(text=WARMUP_TEXT, language="Korean", ref_audio=self.sample_audio,
                                               ref_text="안녕하세요.", x_vector_only_mode=True, max_new_tokens=256)
            else:
                model.generate_custom_voice(text=WARMUP_TEXT, language="Korean", speaker="Sohee", max_new_tokens=256)
            if torch.cuda.is_available():
                torch.cuda.synchronize()
            print(f"Warmup completed in {time.time() - t0:.2f}s (subsequent inferences will be faster)")
        except Exception as e:                                     # as the reference: "this is okay"
            print(f"Warmup failed (this is okay): {e}")

    def get_model(self, model_type: str) -> Any:
        return self.models[model_type] if model_type in self.models else self.load_model(model_type)

    def get_decoder(self, model_type: str) -> Any:
        """The installed MI355X decoder of a loaded model (loads the model first if needed)."""
        self.get_model(model_type)
        return self.decoders[model_type]

    def load_default_models(self) -> None:
        if self.default_model == "all":
            for model_type in ["custom_voice", "voice_design", "base"]:
                self.load_model(model_type)
        else:
            self.load_model(self.default_model)

    def is_loaded(self, model_type: str) -> bool:
        return model_type in self.models

    def get_loaded_models(self) -> List[str]:
        return list(self.models.keys())
