"""Decoder configuration for the 12 Hz speech-tokenizer decode (codes -> 24 kHz waveform).

Every size the decode uses is a field here; nothing is hard-coded in the kernels' host side.
The defaults are the sizes of the in-container sibling `Qwen3OmniMoeCode2WavConfig`
(transformers/models/qwen3_omni_moe/configuration_qwen3_omni_moe.py:538-554) plus the
split-RVQ front end of the Mimi family (transformers/models/mimi/modeling_mimi.py:1084-1137).
The reference reaches this decode implicitly through `Qwen3TTSModel.generate_voice_clone`
(reference server.py:245-251) and `generate_custom_voice` (reference models.py:96-101).
"""
from __future__ import annotations

import dataclasses
import math
from dataclasses import dataclass
from typing import Dict, List, Tuple

FRONTEND_SPLIT_RVQ = "split_rvq"          # Mimi-style split residual VQ + 1x1 output_proj + pre_conv
FRONTEND_MEAN_EMBEDDING = "mean_embedding"  # one embedding table, codes + q*bins, mean over codebooks


@dataclass(frozen=True)
class DecoderConfig:
    # --- code book / front end -------------------------------------------------------------
    frontend: str = FRONTEND_SPLIT_RVQ
    codebook_size: int = 2048
    num_quantizers: int = 16
    num_semantic_quantizers: int = 1
    codebook_dim: int = 256           # width of one code-book row (split_rvq only)
    rvq_out_dim: int = 512            # output of the per-group 1x1 output_proj (split_rvq only)
    rvq_eps: float = 1e-5             # clamp on cluster_usage
    pre_conv_kernel: int = 3          # causal conv rvq_out_dim -> latent_dim (split_rvq only)
    # --- transformer -----------------------------------------------------------------------
    latent_dim: int = 1024            # channel count of the conv stack at frame rate
    hidden_size: int = 1024           # transformer width
    io_proj: bool = True              # Linear latent->hidden before, hidden->latent after (with bias)
    num_attention_heads: int = 16
    num_key_value_heads: int = 16
    head_dim: int = 64
    intermediate_size: int = 3072
    num_hidden_layers: int = 8
    sliding_window: int = 72
    rms_norm_eps: float = 1e-5
    rope_theta: float = 10000.0
    # --- conv decoder ----------------------------------------------------------------------
    upsampling_ratios: Tuple[int, ...] = (2, 2)
    upsample_rates: Tuple[int, ...] = (8, 5, 4, 3)
    decoder_dim: int = 1536
    convnext_eps: float = 1e-6
    snake_eps: float = 1e-9
    dilations: Tuple[int, ...] = (1, 3, 9)
    # --- wrapper ---------------------------------------------------------------------------
    sample_rate: int = 24000
    chunk_size: int = 300
    left_context_size: int = 25

    def __post_init__(self):
        if self.frontend not in (FRONTEND_SPLIT_RVQ, FRONTEND_MEAN_EMBEDDING):
            raise ValueError(f"unknown frontend {self.frontend!r}")
        if self.frontend == FRONTEND_MEAN_EMBEDDING and (self.io_proj or self.latent_dim != self.hidden_size):
            raise ValueError("mean_embedding front end feeds the transformer directly: io_proj must be "
                             "False and latent_dim must equal hidden_size")
        if not self.io_proj and self.latent_dim != self.hidden_size:
            raise ValueError("latent_dim != hidden_size needs io_proj=True")
        if self.num_attention_heads % self.num_key_value_heads:
            raise ValueError("num_attention_heads must be a multiple of num_key_value_heads")
        if self.head_dim % 2:
            raise ValueError("head_dim must be even (rotary embedding)")
        if self.decoder_dim % (2 ** len(self.upsample_rates)):
            raise ValueError("decoder_dim must be divisible by 2**len(upsample_rates)")
        if not (0 < self.num_semantic_quantizers <= self.num_quantizers):
            raise ValueError("num_semantic_quantizers out of range")

    # ---- derived sizes ---------------------------------------------------------------------
    @property
    def total_upsample(self) -> int:
        return int(math.prod(self.upsampling_ratios) * math.prod(self.upsample_rates))

    @property
    def trim(self) -> int:
        """Samples one forward loses at its end: every decoder block's transposed conv
        (kernel 2r, stride r) is trimmed on both sides, which drops one input step; that step
        is worth prod(later rates) output samples (8,5,4,3 -> 480+60+12+3 = 555)."""
        t, later = 0, 1
        for r in reversed(self.upsample_rates):
            later *= r
            t += later
        return t

    @property
    def frame_rate(self) -> float:
        return self.sample_rate / self.total_upsample

    def output_length(self, frames: int) -> int:
        """Length of one un-chunked forward."""
        return frames * self.total_upsample - self.trim

    def chunk_plan(self, frames: int, chunk_size: int | None = None,
                   left_context_size: int | None = None) -> List[Tuple[int, int, int]]:
        """(start_with_context, end, context) per chunk; same rule as the sibling's
        chunked_decode (modeling_qwen3_omni_moe.py:3686-3696): context is left_context_size
        when start - left_context_size > 0, else start."""
        cs = self.chunk_size if chunk_size is None else chunk_size
        lc = self.left_context_size if left_context_size is None else left_context_size
        plan, start = [], 0
        while start < frames:
            end = min(start + cs, frames)
            ctx = lc if start - lc > 0 else start
            plan.append((start - ctx, end, ctx))
            start = end
        return plan

    def chunked_output_length(self, frames: int, chunk_size: int | None = None,
                              left_context_size: int | None = None) -> int:
        return sum((e - s) * self.total_upsample - self.trim - c * self.total_upsample
                   for s, e, c in self.chunk_plan(frames, chunk_size, left_context_size))

    @property
    def block_channels(self) -> List[Tuple[int, int, int]]:
        """(in_channels, out_channels, rate) of each decoder block."""
        return [(self.decoder_dim // 2 ** i, self.decoder_dim // 2 ** (i + 1), r)
                for i, r in enumerate(self.upsample_rates)]

    # ---- work counting ---------------------------------------------------------------------
    def macs_per_frame(self) -> Dict[str, float]:
        """Multiply-accumulates per code frame in steady state (the constant end trim ignored).
        Conv1d: Lout*Cout*(Cin/groups)*k; ConvTranspose1d: Lin*Cin*Cout*k; Linear: rows*in*out.
        Attention score/PV products and pointwise work are not counted."""
        m: Dict[str, float] = {}
        H, Lt, I = self.hidden_size, self.latent_dim, self.intermediate_size
        qd, kvd = self.num_attention_heads * self.head_dim, self.num_key_value_heads * self.head_dim
        fe = 0.0
        if self.frontend == FRONTEND_SPLIT_RVQ:
            groups = 1 + (1 if self.num_quantizers > self.num_semantic_quantizers else 0)
            fe += groups * self.codebook_dim * self.rvq_out_dim
            fe += self.rvq_out_dim * Lt * self.pre_conv_kernel
        if self.io_proj:
            fe += 2 * Lt * H
        m["frontend"] = fe
        m["transformer"] = self.num_hidden_layers * (H * (qd + 2 * kvd) + qd * H + 3 * H * I)
        rows = 1
        for i, f in enumerate(self.upsampling_ratios):
            tc = rows * Lt * Lt * f                        # transposed conv, k = s = f
            rows *= f
            cn = rows * (Lt * 7 + 2 * Lt * 4 * Lt)         # depthwise k7 + two pointwise
            m[f"upsample{i}"] = tc + cn
        m["conv_in"] = rows * Lt * self.decoder_dim * 7
        for i, (cin, cout, r) in enumerate(self.block_channels):
            tc = rows * cin * cout * 2 * r
            rows *= r
            ru = len(self.dilations) * rows * (cout * cout * 7 + cout * cout)
            m[f"block{i}"] = tc + ru
        m["conv_out"] = rows * self.block_channels[-1][1] * 7
        m["total"] = sum(m.values())
        return m

    def flops_per_frame(self) -> float:
        return 2.0 * self.macs_per_frame()["total"]

    def to_dict(self) -> dict:
        d = dataclasses.asdict(self)
        for k, v in d.items():
            if isinstance(v, tuple):
                d[k] = list(v)
        return d

    @classmethod
    def from_dict(cls, d: dict) -> "DecoderConfig":
        known = {f.name for f in dataclasses.fields(cls)}
        kw = {}
        for k, v in d.items():
            if k in known:
                kw[k] = tuple(v) if isinstance(v, list) else v
        return cls(**kw)


def sibling_default_config() -> DecoderConfig:
    """Exactly the sibling Code2Wav architecture (mean-embedding front end), used to check the
    oracle end to end against `Qwen3OmniMoeCode2Wav`."""
    return DecoderConfig(frontend=FRONTEND_MEAN_EMBEDDING, io_proj=False)


def flagship_config() -> DecoderConfig:
    """Bench configuration: sibling trunk sizes with the split-RVQ front end the 12 Hz tokenizer uses."""
    return DecoderConfig()


def tiny_config(frontend: str = FRONTEND_SPLIT_RVQ, **over) -> DecoderConfig:
    """Small architecture with every structural feature of the full one (all kernel/dilation/stride
    combinations, grouped KV heads, window edge reachable with few frames)."""
    base = dict(frontend=frontend, codebook_size=32, num_quantizers=4, num_semantic_quantizers=1,
                codebook_dim=16, rvq_out_dim=32, latent_dim=64, hidden_size=64,
                io_proj=(frontend == FRONTEND_SPLIT_RVQ), num_attention_heads=4, num_key_value_heads=4,
                head_dim=16, intermediate_size=128, num_hidden_layers=2, sliding_window=72,
                decoder_dim=128)
    base.update(over)
    return DecoderConfig(**base)
