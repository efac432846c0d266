"""The tokenizer's ENCODE side: 24 kHz audio -> 12.5 Hz code frames (configuration, tensor names, synthetic weights).

The reference would reach it through `tokenizer.encode(ref_audio)` in its in-context voice-clone mode, which its server
switches off (`x_vector_only_mode=True`, reference server.py:221, 283), and the arithmetic lives in the absent `qwen-tts`
package.  The sizes and tensor names here are those of the Mimi encoder in the in-container sibling
(transformers/models/mimi/modeling_mimi.py: MimiEncoder :450-492, MimiTransformerModel :782-928, the stride-2
`downsample` conv :1197-1206, `_encode_frame` :1230-1263), whose quantiser the decode side already follows.
The speaker x-vector extractor of the reference's default mode is a separate network of `qwen-tts` and is not part of this.
"""
from __future__ import annotations

import dataclasses
import math
from dataclasses import dataclass
from typing import Dict, List, Tuple

import torch


@dataclass(frozen=True)
class EncoderConfig:
    num_filters: int = 64                 # channels after the first conv; doubled by every strided conv
    kernel_size: int = 7                  # first conv (1 audio channel -> num_filters)
    residual_kernel_size: int = 3         # first conv of a residual block (dim -> dim / compress), dilation 1
    compress: int = 2
    last_kernel_size: int = 3             # last conv (num_filters * 2^len(ratios) -> hidden_size)
    ratios: Tuple[int, ...] = (4, 5, 6, 8)   # strides in the order the encoder applies them (kernel = 2 * stride)
    hidden_size: int = 512
    num_attention_heads: int = 8
    head_dim: int = 64
    intermediate_size: int = 2048
    num_hidden_layers: int = 8
    sliding_window: int = 250
    norm_eps: float = 1e-5
    rope_theta: float = 10000.0
    downsample_stride: int = 2            # 25 Hz -> 12.5 Hz, kernel = 2 * stride, replicate padding, no bias

    def __post_init__(self):
        if self.num_filters % 8 or (self.num_filters // self.compress) % 8:
            raise ValueError("num_filters and num_filters / compress must be multiples of 8")
        if self.hidden_size % 8 or self.head_dim % 2:
            raise ValueError("hidden_size must be a multiple of 8 and head_dim even")
        if self.hidden_size != self.num_attention_heads * self.head_dim:
            raise ValueError("hidden_size must equal num_attention_heads * head_dim")

    @property
    def hop(self) -> int:
        """Audio samples per code frame."""
        return math.prod(self.ratios) * self.downsample_stride

    def frames(self, samples: int) -> int:
        """Code frames for `samples` audio samples: every strided conv rounds up (mimi :269-279)."""
        n = samples
        for r in self.ratios:
            n = -(-n // r)
        return -(-n // self.downsample_stride)

    def to_dict(self) -> dict:
        d = dataclasses.asdict(self)
        d["ratios"] = list(self.ratios)
        return d

    @classmethod
    def from_dict(cls, d: dict) -> "EncoderConfig":
        d = dict(d)
        d["ratios"] = tuple(d["ratios"])
        return cls(**d)


def tiny_encoder_config(**over) -> EncoderConfig:
    base = dict(num_filters=16, ratios=(2, 3), hidden_size=32, num_attention_heads=2, head_dim=16, intermediate_size=64,
                num_hidden_layers=2, sliding_window=5)
    base.update(over)
    return EncoderConfig(**base)


_W, _B, _ONE, _SCALE = "w", "b", "one", "scale"


def encoder_param_specs(cfg: EncoderConfig) -> List[Tuple[str, Tuple[int, ...], str, float]]:
    """(name, shape, kind, fan_in) of every encoder tensor, names as in the sibling's MimiModel state dict."""
    s: List[Tuple[str, Tuple[int, ...], str, float]] = []
    F0 = cfg.num_filters
    s.append(("encoder.layers.0.conv.weight", (F0, 1, cfg.kernel_size), _W, cfg.kernel_size))
    s.append(("encoder.layers.0.conv.bias", (F0,), _B, 1.0))
    dim, idx = F0, 1
    for r in cfg.ratios:
        hid = dim // cfg.compress
        p = f"encoder.layers.{idx}.block."
        s.append((p + "1.conv.weight", (hid, dim, cfg.residual_kernel_size), _W, dim * cfg.residual_kernel_size))
        s.append((p + "1.conv.bias", (hid,), _B, 1.0))
        s.append((p + "3.conv.weight", (dim, hid, 1), _W, hid * 4.0))       # residual branch at gain 1/2
        s.append((p + "3.conv.bias", (dim,), _B, 1.0))
        s.append((f"encoder.layers.{idx + 2}.conv.weight", (2 * dim, dim, 2 * r), _W, dim * 2 * r))
        s.append((f"encoder.layers.{idx + 2}.conv.bias", (2 * dim,), _B, 1.0))
        dim *= 2
        idx += 3
    s.append((f"encoder.layers.{idx + 1}.conv.weight", (cfg.hidden_size, dim, cfg.last_kernel_size), _W, dim * cfg.last_kernel_size))
    s.append((f"encoder.layers.{idx + 1}.conv.bias", (cfg.hidden_size,), _B, 1.0))
    H, I = cfg.hidden_size, cfg.intermediate_size
    for l in range(cfg.num_hidden_layers):
        p = f"encoder_transformer.layers.{l}."
        for n in ("q_proj", "k_proj", "v_proj", "o_proj"):
            s.append((p + f"self_attn.{n}.weight", (H, H), _W, H))
        s.append((p + "mlp.fc1.weight", (I, H), _W, H))
        s.append((p + "mlp.fc2.weight", (H, I), _W, I))
        for n in ("input_layernorm", "post_attention_layernorm"):
            s.append((p + n + ".weight", (H,), _ONE, 1.0))
            s.append((p + n + ".bias", (H,), _B, 1.0))
        s.append((p + "self_attn_layer_scale.scale", (H,), _SCALE, 1.0))
        s.append((p + "mlp_layer_scale.scale", (H,), _SCALE, 1.0))
    k = 2 * cfg.downsample_stride
    s.append(("downsample.conv.weight", (H, H, k), _W, H * k))
    return s


def synthetic_encoder_weights(cfg: EncoderConfig, codebook_dim: int, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Seeded fp32 CPU tensors for every entry of `encoder_param_specs`, plus the quantiser's two `input_proj`
    ([codebook_dim][hidden_size][1]) that the encode direction needs beside the decode side's code books."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    out: Dict[str, torch.Tensor] = {}
    for name, shape, kind, fan in encoder_param_specs(cfg):
        if kind == _W:
            t = torch.randn(shape, generator=g) * (1.0 / math.sqrt(fan))
        elif kind == _B:
            t = torch.randn(shape, generator=g) * 0.02
        elif kind == _ONE:
            t = 1.0 + torch.randn(shape, generator=g) * 0.1
        else:
            t = torch.full(shape, 0.5)
        out[name] = t.to(torch.float32).contiguous()
    for grp in ("semantic", "acoustic"):
        out[f"quantizer.{grp}.input_proj.weight"] = (torch.randn(codebook_dim, cfg.hidden_size, 1, generator=g)
                                                     * cfg.hidden_size ** -0.5 * 3.0).contiguous()
    return out
