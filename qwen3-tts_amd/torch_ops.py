"""PyTorch custom ops (`torch.ops.q3tts.*`) over the C-ABI library.

The ops are thin: each looks up a `Decoder` by integer handle (contexts own device memory and cannot travel through
an op schema) and calls the same ctypes entry point the `Decoder` methods use, on torch's current stream.  Shape
functions are registered so the ops can be traced with fake tensors; there is no CPU kernel - calling an op on CPU
tensors raises NotImplementedError from the dispatcher.

    h = torch_ops.register_decoder(dec)
    wav = torch.ops.q3tts.decode_forward(codes, h)          # [B,1,T*1920-555]
    sums = torch.ops.q3tts.rvq_dequant(codes, h)            # [B,T,G,codebook_dim], bit-exact fp32
    y = torch.ops.q3tts.causal_conv1d(x, w, b, 1, "f32")    # operator level, channels-last
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

_decoders: Dict[int, object] = {}
_next = [1]


def register_decoder(decoder) -> int:
    h = _next[0]
    _next[0] += 1
    _decoders[h] = decoder
    return h


def unregister_decoder(handle: int) -> None:
    _decoders.pop(handle, None)


def _dec(handle: int):
    try:
        return _decoders[handle]
    except KeyError:
        raise KeyError(f"no decoder registered under handle {handle}") from None


@torch.library.custom_op("q3tts::decode_forward", mutates_args=(), device_types="cuda")
def decode_forward(codes: torch.Tensor, handle: int) -> torch.Tensor:
    return _dec(handle).forward(codes)


@decode_forward.register_fake
def _(codes, handle):
    cfg = _dec(handle).cfg
    return codes.new_empty((codes.shape[0], 1, cfg.output_length(codes.shape[2])), dtype=torch.float32)


@torch.library.custom_op("q3tts::chunked_decode", mutates_args=(), device_types="cuda")
def chunked_decode(codes: torch.Tensor, handle: int, chunk_size: int, left_context_size: int) -> torch.Tensor:
    return _dec(handle).chunked_decode(codes, chunk_size, left_context_size)


@chunked_decode.register_fake
def _(codes, handle, chunk_size, left_context_size):
    cfg = _dec(handle).cfg
    n = cfg.chunked_output_length(codes.shape[2], chunk_size, left_context_size)
    return codes.new_empty((codes.shape[0], 1, n), dtype=torch.float32)


@torch.library.custom_op("q3tts::rvq_dequant", mutates_args=(), device_types="cuda")
def rvq_dequant(codes: torch.Tensor, handle: int) -> torch.Tensor:
    return _dec(handle).rvq_dequant(codes)


@rvq_dequant.register_fake
def _(codes, handle):
    cfg = _dec(handle).cfg
    groups = 2 if cfg.num_quantizers > cfg.num_semantic_quantizers else 1
    return codes.new_empty((codes.shape[0], codes.shape[2], groups, cfg.codebook_dim), dtype=torch.float32)


@torch.library.custom_op("q3tts::causal_conv1d", mutates_args=(), device_types="cuda")
def causal_conv1d(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], dilation: int,
                  precision: str) -> torch.Tensor:
    from . import decoder
    return decoder.causal_conv1d(x, weight, bias, dilation, precision)


@causal_conv1d.register_fake
def _(x, weight, bias, dilation, precision):
    return x.new_empty((x.shape[0], x.shape[1], weight.shape[0]), dtype=torch.float32)


@torch.library.custom_op("q3tts::causal_trans_conv1d", mutates_args=(), device_types="cuda")
def causal_trans_conv1d(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], stride: int,
                        precision: str) -> torch.Tensor:
    from . import decoder
    return decoder.causal_trans_conv1d(x, weight, bias, stride, precision)


@causal_trans_conv1d.register_fake
def _(x, weight, bias, stride, precision):
    lout = (x.shape[1] - 1) * stride - weight.shape[2] + 2 * stride
    return x.new_empty((x.shape[0], max(lout, 0), weight.shape[1]), dtype=torch.float32)


@torch.library.custom_op("q3tts::snake_beta", mutates_args=(), device_types="cuda")
def snake_beta(x: torch.Tensor, alpha: torch.Tensor, beta: torch.Tensor, eps: float) -> torch.Tensor:
    from . import decoder
    return decoder.snake_beta(x, alpha, beta, eps)


@snake_beta.register_fake
def _(x, alpha, beta, eps):
    return torch.empty_like(x, dtype=torch.float32)
