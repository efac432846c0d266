"""Local-path checkpoint loading for the decoder.

The reference loads a whole model with `Qwen3TTSModel.from_pretrained(model_path, ...)` (reference models.py:43-48), where
`model_path` is a local directory when one exists and otherwise a hub NAME that the library downloads (reference
config.py:23-54).  This package never resolves names: `load_local` reads `config.json` and `*.safetensors` files from a
directory that exists, with `safetensors` (a format that holds tensors only - nothing in the file is executed), and refuses
anything else.  The decoder's tensors are found the same way `install.decoder_weights` finds them in a live state dict: by
the conv decoder's first weight, `decoder.0.conv.weight`, whose key prefix is stripped from every key that carries it; the
architecture comes from the tensor shapes (`install.config_from_weights`), the config file supplying only what no shape
shows.  `save_local` writes the same layout (tests, and exporting a decoder sub-checkpoint once from a loaded model)."""
from __future__ import annotations

import json
import os
from typing import Any, Dict, Optional, Tuple

import torch

from .config import DecoderConfig
from .install import _ANCHOR, config_from_weights, rename_weights
from .weights import param_specs

_CONFIG_KEYS = ("decoder_config", "code2wav_config")        # where a tokenizer's config.json may nest the decoder's sizes


def require_local_dir(path: str, what: str = "model path") -> str:
    """`path` as an absolute directory path, or FileNotFoundError: hub names ("Qwen/...") and anything else that is not a
    directory on this machine are refused - nothing is ever fetched."""
    if not isinstance(path, (str, os.PathLike)) or not os.path.isdir(path):
        raise FileNotFoundError(f"{what} {path!r} is not a local directory; this package loads checkpoints from local "
                                "directories only and never resolves hub names (set the model's path to a directory that exists)")
    return os.path.abspath(path)


def _safetensors_files(root: str, max_depth: int = 2):
    out = []
    for cur, dirs, files in os.walk(root):
        depth = 0 if cur == root else cur[len(root):].count(os.sep)
        if depth >= max_depth:
            dirs[:] = []
        dirs.sort()
        out += [os.path.join(cur, f) for f in sorted(files) if f.endswith(".safetensors")]
    return out


def _read_config(dirpath: str) -> Dict[str, Any]:
    p = os.path.join(dirpath, "config.json")
    if not os.path.isfile(p):
        return {}
    with open(p, "r", encoding="utf-8") as fh:
        cfg = json.load(fh)
    for k in _CONFIG_KEYS:
        if isinstance(cfg.get(k), dict):
            return cfg[k]
    return cfg


def load_local(path: str, rename: Optional[Dict[str, str]] = None) -> Tuple[DecoderConfig, Dict[str, torch.Tensor]]:
    """-> (DecoderConfig, fp32 CPU tensors under this package's names) from a local checkpoint directory (or one
    .safetensors file).  Searches `path` and up to two levels of sub-directories (a model directory usually keeps the
    speech tokenizer in one) for the file(s) holding `...decoder.0.conv.weight`; the config.json beside them supplies the
    sizes no tensor shape shows.  Shapes that do not fit together raise ValueError."""
    from safetensors import safe_open
    if os.path.isfile(path) and str(path).endswith(".safetensors"):
        files, root = [os.path.abspath(path)], os.path.dirname(os.path.abspath(path))
    else:
        root = require_local_dir(path, "checkpoint path")
        files = _safetensors_files(root)
    if not files:
        raise FileNotFoundError(f"no .safetensors file under {root!r}")
    # the anchor tells which file(s) and which key prefix belong to the decoder
    prefix, anchor_dir = None, None
    for f in files:
        with safe_open(f, framework="pt", device="cpu") as fh:
            for k in fh.keys():
                k2 = k
                for old, new in (rename or {}).items():
                    k2 = k2.replace(old, new)
                if k2 == _ANCHOR or k2.endswith("." + _ANCHOR):
                    prefix, anchor_dir = k2[: len(k2) - len(_ANCHOR)], os.path.dirname(f)
                    break
        if prefix is not None:
            break
    if prefix is None:
        raise KeyError(f"no tensor named '...{_ANCHOR}' in any .safetensors file under {root!r}; pass rename= if the checkpoint "
                       "uses other key names")
    raw: Dict[str, torch.Tensor] = {}
    for f in files:
        if os.path.dirname(f) != anchor_dir:
            continue                                   # shards of one checkpoint sit side by side
        with safe_open(f, framework="pt", device="cpu") as fh:
            for k in fh.keys():
                raw[k] = fh.get_tensor(k)
    w = {k[len(prefix):]: v for k, v in rename_weights(raw, rename).items() if k.startswith(prefix)}
    cfg = config_from_weights(_read_config(anchor_dir), w)
    keep = {n for n, *_ in param_specs(cfg)} | {k for k in w if k.startswith("quantizer.") and k.endswith(".input_proj.weight")}
    return cfg, {k: v.to(torch.float32).contiguous() for k, v in w.items() if k in keep}


def save_local(path: str, cfg: DecoderConfig, weights: Dict[str, torch.Tensor], prefix: str = "decoder.",
               dtype: torch.dtype = torch.float32) -> str:
    """Write `config.json` (under "decoder_config") and `model.safetensors` (keys `prefix` + name) into directory `path`."""
    from safetensors.torch import save_file
    os.makedirs(path, exist_ok=True)
    with open(os.path.join(path, "config.json"), "w", encoding="utf-8") as fh:
        json.dump({"decoder_config": cfg.to_dict()}, fh, indent=1)
    save_file({prefix + k: v.detach().to("cpu", dtype).contiguous() for k, v in weights.items()}, os.path.join(path, "model.safetensors"))
    return path
