"""Parameter inventory of the decoder and a seeded synthetic-weight generator.

Names follow the state-dict keys of the in-container sibling `Qwen3OmniMoeCode2Wav`
(transformers/models/qwen3_omni_moe/modeling_qwen3_omni_moe.py:3636-3670) so that a sibling
state dict loads unchanged; the split-RVQ front end adds Mimi-style buffers
(transformers/models/mimi/modeling_mimi.py:964-1048).

There is no checkpoint of the 12 Hz tokenizer in this build's environment, so tests and the bench
use the generator below.  Library default init (std 0.02, layer scale 0.01, gamma 1e-6, alpha =
beta = 0) makes every residual branch numerically invisible; the generator instead draws
variance-preserving weights so that each branch contributes at O(1).
"""
from __future__ import annotations

import math
from typing import Dict, List, Tuple

import torch

from .config import DecoderConfig, FRONTEND_MEAN_EMBEDDING

# kinds: how the generator draws a tensor
_W = "weight"       # N(0, gain^2 / fan_in)
_B = "bias"         # N(0, 0.02^2)
_ONE = "norm"       # 1 + N(0, 0.1^2)
_SNAKE = "snake"    # N(0, 0.3^2)
_SCALE = "scale"    # constant 0.5
_EMB = "embed"      # N(0, 1)
_USAGE = "usage"    # U(0.5, 50), a few entries below the clamp


def param_specs(cfg: DecoderConfig) -> List[Tuple[str, Tuple[int, ...], str, float]]:
    """(name, shape, kind, fan_in*1/gain^2) for every tensor the decoder reads, in forward order."""
    s: List[Tuple[str, Tuple[int, ...], str, float]] = []
    H, Lt = cfg.hidden_size, cfg.latent_dim
    if cfg.frontend == FRONTEND_MEAN_EMBEDDING:
        s.append(("code_embedding.weight", (cfg.codebook_size * cfg.num_quantizers, H), _EMB, 1.0))
    else:
        groups = [("semantic", cfg.num_semantic_quantizers)]
        if cfg.num_quantizers > cfg.num_semantic_quantizers:
            groups.append(("acoustic", cfg.num_quantizers - cfg.num_semantic_quantizers))
        for g, n in groups:
            for i in range(n):
                s.append((f"quantizer.{g}.layers.{i}.embed_sum", (cfg.codebook_size, cfg.codebook_dim), _EMB, 1.0))
                s.append((f"quantizer.{g}.layers.{i}.cluster_usage", (cfg.codebook_size,), _USAGE, 1.0))
            # each group's projection sees a sum of n unit-variance rows
            s.append((f"quantizer.{g}.output_proj.weight", (cfg.rvq_out_dim, cfg.codebook_dim, 1), _W,
                      cfg.codebook_dim * n * len(groups)))
        k = cfg.pre_conv_kernel
        s.append(("pre_conv.conv.weight", (Lt, cfg.rvq_out_dim, k), _W, cfg.rvq_out_dim * k))
        s.append(("pre_conv.conv.bias", (Lt,), _B, 1.0))
    if cfg.io_proj:
        s.append(("pre_transformer.input_proj.weight", (H, Lt), _W, Lt))
        s.append(("pre_transformer.input_proj.bias", (H,), _B, 1.0))
    qd = cfg.num_attention_heads * cfg.head_dim
    kvd = cfg.num_key_value_heads * cfg.head_dim
    I = cfg.intermediate_size
    for l in range(cfg.num_hidden_layers):
        p = f"pre_transformer.layers.{l}."
        s.append((p + "input_layernorm.weight", (H,), _ONE, 1.0))
        s.append((p + "self_attn.q_proj.weight", (qd, H), _W, H))
        s.append((p + "self_attn.k_proj.weight", (kvd, H), _W, H))
        s.append((p + "self_attn.v_proj.weight", (kvd, H), _W, H))
        s.append((p + "self_attn.o_proj.weight", (H, qd), _W, qd))
        s.append((p + "self_attn_layer_scale.scale", (H,), _SCALE, 1.0))
        s.append((p + "post_attention_layernorm.weight", (H,), _ONE, 1.0))
        s.append((p + "mlp.gate_proj.weight", (I, H), _W, H))
        s.append((p + "mlp.up_proj.weight", (I, H), _W, H))
        s.append((p + "mlp.down_proj.weight", (H, I), _W, I / 4.0))
        s.append((p + "mlp_layer_scale.scale", (H,), _SCALE, 1.0))
    s.append(("pre_transformer.norm.weight", (H,), _ONE, 1.0))
    if cfg.io_proj:
        s.append(("pre_transformer.output_proj.weight", (Lt, H), _W, H))
        s.append(("pre_transformer.output_proj.bias", (Lt,), _B, 1.0))
    for i, f in enumerate(cfg.upsampling_ratios):
        p = f"upsample.{i}."
        s.append((p + "0.conv.weight", (Lt, Lt, f), _W, Lt))          # ConvTranspose1d [Cin, Cout, k], k = s
        s.append((p + "0.conv.bias", (Lt,), _B, 1.0))
        s.append((p + "1.dwconv.conv.weight", (Lt, 1, 7), _W, 7))
        s.append((p + "1.dwconv.conv.bias", (Lt,), _B, 1.0))
        s.append((p + "1.norm.weight", (Lt,), _ONE, 1.0))
        s.append((p + "1.norm.bias", (Lt,), _B, 1.0))
        s.append((p + "1.pwconv1.weight", (4 * Lt, Lt), _W, Lt))
        s.append((p + "1.pwconv1.bias", (4 * Lt,), _B, 1.0))
        s.append((p + "1.pwconv2.weight", (Lt, 4 * Lt), _W, 4 * Lt / 4.0))
        s.append((p + "1.pwconv2.bias", (Lt,), _B, 1.0))
        s.append((p + "1.gamma", (Lt,), _SCALE, 1.0))
    s.append(("decoder.0.conv.weight", (cfg.decoder_dim, Lt, 7), _W, Lt * 7))
    s.append(("decoder.0.conv.bias", (cfg.decoder_dim,), _B, 1.0))
    for i, (cin, cout, r) in enumerate(cfg.block_channels):
        p = f"decoder.{i + 1}.block."
        s.append((p + "0.alpha", (cin,), _SNAKE, 1.0))
        s.append((p + "0.beta", (cin,), _SNAKE, 1.0))
        # 2 taps reach each output; gain 2/3 undoes the ~1.5x growth of the three residual units that follow,
        # so activations stay O(1) through the four blocks instead of growing 5x
        s.append((p + "1.conv.weight", (cin, cout, 2 * r), _W, cin * 2 * 2.25))
        s.append((p + "1.conv.bias", (cout,), _B, 1.0))
        for j, _d in enumerate(cfg.dilations):
            q = p + f"{j + 2}."
            s.append((q + "act1.alpha", (cout,), _SNAKE, 1.0))
            s.append((q + "act1.beta", (cout,), _SNAKE, 1.0))
            s.append((q + "conv1.conv.weight", (cout, cout, 7), _W, cout * 7))
            s.append((q + "conv1.conv.bias", (cout,), _B, 1.0))
            s.append((q + "act2.alpha", (cout,), _SNAKE, 1.0))
            s.append((q + "act2.beta", (cout,), _SNAKE, 1.0))
            s.append((q + "conv2.conv.weight", (cout, cout, 1), _W, cout * 4.0))  # residual branch at gain 1/2
            s.append((q + "conv2.conv.bias", (cout,), _B, 1.0))
    nb = len(cfg.upsample_rates)
    cl = cfg.block_channels[-1][1]
    s.append((f"decoder.{nb + 1}.alpha", (cl,), _SNAKE, 1.0))
    s.append((f"decoder.{nb + 1}.beta", (cl,), _SNAKE, 1.0))
    s.append((f"decoder.{nb + 2}.conv.weight", (1, cl, 7), _W, cl * 7 * 36.0))   # keeps most samples inside the clamp
    s.append((f"decoder.{nb + 2}.conv.bias", (1,), _B, 1.0))
    return s


PROFILES = ("chaotic", "conditioned")


def synthetic_weights(cfg: DecoderConfig, seed: int = 0, profile: str = "chaotic") -> Dict[str, torch.Tensor]:
    """Seeded fp32 CPU tensors for every entry of `param_specs`.

    profile "chaotic" (the default, and the stress case): alpha, beta ~ N(0, 0.3) and residual branches at gain 1/2.  SnakeBeta
    is then x + sin(x)^2 with slope 1 + sin(2x) in [0, 2], and the conv decoder amplifies any perturbation of its input by
    about 5.5 dB per block: a bf16 run ends as a noise realisation (~16 dB at the waveform), whoever runs it.

    profile "conditioned": the same architecture with every branch still visible, but a decoder that does not amplify what it
    is fed: alpha ~ N(-0.7, 0.2), beta ~ N(0.7, 0.2) (SnakeBeta slope within 1 +- 0.25), the residual units' second conv at
    gain 1/4, LayerScale and ConvNeXt gamma at 0.25, the final conv scaled so the waveform keeps an rms near 0.3.  Rounding errors then add up instead of multiplying, which is what
    makes an absolute bf16 tolerance for the whole launch sequence meaningful."""
    if profile not in PROFILES:
        raise ValueError(f"unknown weight profile {profile!r}")
    cond = profile == "conditioned"
    g = torch.Generator(device="cpu").manual_seed(seed)
    out: Dict[str, torch.Tensor] = {}
    for name, shape, kind, fan in param_specs(cfg):
        if kind == _W:
            t = torch.randn(shape, generator=g) * (1.0 / math.sqrt(fan))
            if cond and name.endswith("conv2.conv.weight"):
                t = t * 0.5                                        # residual branch at gain 1/4 instead of 1/2
            if cond and shape[0] == 1 and name.startswith("decoder."):
                t = t * 5.0                                        # final conv: the tamer blocks leave less energy; waveform rms ~0.3
        elif kind == _B:
            t = torch.randn(shape, generator=g) * 0.02
        elif kind == _ONE:
            t = 1.0 + torch.randn(shape, generator=g) * 0.1
        elif kind == _SNAKE:
            if cond:
                t = torch.randn(shape, generator=g) * 0.2 + (-0.7 if name.endswith("alpha") else 0.7)
            else:
                t = torch.randn(shape, generator=g) * 0.3
        elif kind == _SCALE:
            t = torch.full(shape, 0.25 if cond else 0.5)
        elif kind == _EMB:
            t = torch.randn(shape, generator=g)
        elif kind == _USAGE:
            t = torch.rand(shape, generator=g) * 49.5 + 0.5
            n = max(1, shape[0] // 64)
            idx = torch.randperm(shape[0], generator=g)[:n]
            t[idx] = torch.rand(n, generator=g) * 2e-5          # straddles the 1e-5 clamp
            # keep those rows bounded after the divide
        else:  # pragma: no cover
            raise AssertionError(kind)
        out[name] = t.to(torch.float32).contiguous()
    # rows whose usage sits near the clamp would be ~1e5 large: shrink their embed_sum so the dequantised
    # row stays O(1) while the clamp is still what decides its value
    for name in list(out):
        if name.endswith("cluster_usage"):
            es = name[: -len("cluster_usage")] + "embed_sum"
            u = out[name]
            small = u < 1e-3
            out[es][small] = out[es][small] * 1e-5
            big = ~small
            out[es][big] = out[es][big] * u[big][:, None]     # embed ~ N(0,1) after the divide
    return out


def validate_weights(cfg: DecoderConfig, weights: Dict[str, torch.Tensor]) -> None:
    """Refuse a state dict whose names or shapes do not match the configuration."""
    missing, bad = [], []
    for name, shape, _k, _f in param_specs(cfg):
        t = weights.get(name)
        if t is None:
            missing.append(name)
        elif tuple(t.shape) != tuple(shape):
            bad.append(f"{name}: expected {tuple(shape)}, got {tuple(t.shape)}")
    if missing or bad:
        raise ValueError("decoder weights do not match the configuration: "
                         + "; ".join((["missing " + ", ".join(missing[:8]) + ("..." if len(missing) > 8 else "")]
                                      if missing else []) + bad[:8]))


def random_codes(cfg: DecoderConfig, batch: int, frames: int, seed: int) -> torch.Tensor:
    """i.i.d. uniform codes [B, Q, T] int64 (worst case for gather locality)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    return torch.randint(0, cfg.codebook_size, (batch, cfg.num_quantizers, frames), generator=g, dtype=torch.int64)
