"""CPU oracle for the tokenizer's ENCODE side (24 kHz audio -> 12.5 Hz code frames).

TEST INFRASTRUCTURE ONLY: used by `tests/` as the checker, never by the product path.

PARITY UNPINNED BY THE REFERENCE, as for the decode: the reference would call `tokenizer.encode` of the absent `qwen-tts`
package (and its server keeps that mode off, reference server.py:221, 283).  This is a restatement, in plain fp32 torch
functions, of the encode path of the in-container sibling `MimiModel` (transformers 5.15.0,
transformers/models/mimi/modeling_mimi.py; `mimi` below), pinned to that model's own outputs by the fixtures
tests/golden/encoder_*.npz (tests/golden/make_goldens.py::make_encoder).
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

import decoder_oracle as dec

Tensor = torch.Tensor


def causal_conv1d(x: Tensor, weight: Tensor, bias: Optional[Tensor], stride: int = 1, dilation: int = 1,
                  pad_mode: str = "constant") -> Tensor:
    """x [B, C, L].  Left padding of (effective kernel - stride), extra right padding up to a whole number of strides,
    both in `pad_mode` (mimi :269-279, :327-347)."""
    k_eff = (weight.shape[-1] - 1) * dilation + 1
    pad_total = k_eff - stride
    L = x.shape[-1]
    n_frames = math.ceil((L - k_eff + pad_total) / stride + 1) - 1
    extra = n_frames * stride + k_eff - pad_total - L
    x = F.pad(x, (pad_total, extra), mode=pad_mode)
    return F.conv1d(x, weight, bias, stride=stride, dilation=dilation)


def seanet_encoder(cfg, w: Dict[str, Tensor], audio: Tensor) -> Tensor:
    """audio [B, 1, L] -> [B, hidden, L / prod(ratios)] (mimi :450-492; residual block :408-447: ELU, conv k3 to
    dim / compress, ELU, conv k1 back, identity shortcut)."""
    x = causal_conv1d(audio, w["encoder.layers.0.conv.weight"], w["encoder.layers.0.conv.bias"])
    idx = 1
    for r in cfg.ratios:
        p = f"encoder.layers.{idx}.block."
        h = causal_conv1d(F.elu(x), w[p + "1.conv.weight"], w[p + "1.conv.bias"])
        h = causal_conv1d(F.elu(h), w[p + "3.conv.weight"], w[p + "3.conv.bias"])
        x = x + h
        q = f"encoder.layers.{idx + 2}.conv."
        x = causal_conv1d(F.elu(x), w[q + "weight"], w[q + "bias"], stride=r)
        idx += 3
    q = f"encoder.layers.{idx + 1}.conv."
    return causal_conv1d(F.elu(x), w[q + "weight"], w[q + "bias"])


def encoder_transformer(cfg, w: Dict[str, Tensor], x: Tensor) -> Tensor:
    """x [B, T, H].  Pre-norm layers with LayerNorm (with bias), bias-free projections, rotary embedding, causal
    sliding-window attention, GELU MLP and LayerScale on both branches; no final norm (mimi :729-779, :657-726, :602-615)."""
    B, T, H = x.shape
    nh, d = cfg.num_attention_heads, cfg.head_dim
    cos, sin = dec.rope_tables(d, cfg.rope_theta, torch.arange(T))
    for l in range(cfg.num_hidden_layers):
        p = f"encoder_transformer.layers.{l}."
        h = F.layer_norm(x, (H,), w[p + "input_layernorm.weight"], w[p + "input_layernorm.bias"], cfg.norm_eps)
        q = F.linear(h, w[p + "self_attn.q_proj.weight"]).view(B, T, nh, d).transpose(1, 2)
        k = F.linear(h, w[p + "self_attn.k_proj.weight"]).view(B, T, nh, d).transpose(1, 2)
        v = F.linear(h, w[p + "self_attn.v_proj.weight"]).view(B, T, nh, d).transpose(1, 2)
        q = q * cos + dec.rotate_half(q) * sin
        k = k * cos + dec.rotate_half(k) * sin
        a = dec.sliding_window_attention(q, k, v, cfg.sliding_window).transpose(1, 2).reshape(B, T, nh * d)
        x = x + w[p + "self_attn_layer_scale.scale"] * F.linear(a, w[p + "self_attn.o_proj.weight"])
        h = F.layer_norm(x, (H,), w[p + "post_attention_layernorm.weight"], w[p + "post_attention_layernorm.bias"], cfg.norm_eps)
        m = F.linear(F.gelu(F.linear(h, w[p + "mlp.fc1.weight"])), w[p + "mlp.fc2.weight"])
        x = x + w[p + "mlp_layer_scale.scale"] * m
    return x


def embeddings(cfg, w: Dict[str, Tensor], audio: Tensor) -> Tensor:
    """audio [B, 1, L] -> what the quantiser encodes, [B, hidden, frames]: conv encoder, transformer, then the stride-2 conv with
    REPLICATE padding and no bias (mimi :1245-1259, :1197-1206)."""
    x = seanet_encoder(cfg, w, audio.float())
    x = encoder_transformer(cfg, w, x.transpose(1, 2)).transpose(1, 2)
    return causal_conv1d(x, w["downsample.conv.weight"], None, stride=cfg.downsample_stride, pad_mode="replicate")


def encode(cfg, rvq_cfg, w: Dict[str, Tensor], audio: Tensor) -> Tuple[Tensor, Tensor]:
    """-> (codes [B, Q, frames] int64, embeddings [B, hidden, frames]); the quantiser is the decode side's split RVQ run in
    its encode direction (decoder_oracle.rvq_encode; mimi :1261-1262)."""
    emb = embeddings(cfg, w, audio)
    return dec.rvq_encode(rvq_cfg, w, emb), emb
