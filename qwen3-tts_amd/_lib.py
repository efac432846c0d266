"""ctypes binding of libq3c.so (declared in include/q3c.h).  There is no fallback: if the library is
missing or a call fails, an exception is raised."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

from .config import DecoderConfig, FRONTEND_MEAN_EMBEDDING

# Q3C_LIB names another build of the library (A/B measurements of two builds on one device)
LIB_PATH = os.environ.get("Q3C_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libq3c.so")
Q3C_MAX_STAGES = 8

PRECISION = {"f32": 0, "fp32": 0, "float32": 0, "bf16": 1, "bfloat16": 1}

# stage ids of q3c_set_tap
STATIC_SCRATCH_BYTES = 64 << 20        # Q3C_STATIC_SCRATCH_BYTES of include/q3c.h
STAGE_RVQ_OUT, STAGE_TRANSFORMER_IN, STAGE_TRANSFORMER_OUT, STAGE_UPSAMPLE0, STAGE_CONV_IN, STAGE_BLOCK0 = 0, 1, 2, 3, 16, 17

EXPORTS = ["q3c_create", "q3c_destroy", "q3c_last_error", "q3c_version", "q3c_load_weight", "q3c_finalize",
           "q3c_workspace_bytes", "q3c_output_length", "q3c_forward", "q3c_forward_skip", "q3c_forward_scatter", "q3c_forward_static", "q3c_rvq_dequant_f32", "q3c_rvq_encode", "q3c_check_errors",
           "q3c_set_tap", "q3c_causal_conv1d", "q3c_causal_tconv1d", "q3c_snake_beta", "q3c_profile_enable",
           "q3c_profile_read", "q3c_stream_pool_create", "q3c_stream_pool_destroy", "q3c_stream_state_bytes",
           "q3c_stream_reset", "q3c_stream_workspace_bytes", "q3c_stream_push", "q3c_stream_push_graph", "q3c_residual_unit", "q3c_convnext_block",
           "q3c_transformer", "q3c_wav_to_pcm16", "q3c_encoder_configure", "q3c_encode_frames", "q3c_encode_workspace_bytes", "q3c_encode",
           "q3c_debug_set_tuning", "q3c_debug_set_timeline", "q3c_debug_set_operator_context"]


class Q3cProfileEntry(C.Structure):
    _fields_ = [("kernel", C.c_char * 64), ("launches", C.c_int64), ("ms", C.c_double), ("flops", C.c_double),
                ("bytes", C.c_double)]


class Q3cConfig(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int32), ("frontend", C.c_int32), ("precision", C.c_int32),
        ("codebook_size", C.c_int32), ("num_quantizers", C.c_int32), ("num_semantic_quantizers", C.c_int32),
        ("codebook_dim", C.c_int32), ("rvq_out_dim", C.c_int32), ("pre_conv_kernel", C.c_int32),
        ("latent_dim", C.c_int32), ("hidden_size", C.c_int32), ("io_proj", C.c_int32),
        ("num_attention_heads", C.c_int32), ("num_key_value_heads", C.c_int32), ("head_dim", C.c_int32),
        ("intermediate_size", C.c_int32), ("num_hidden_layers", C.c_int32), ("sliding_window", C.c_int32),
        ("num_upsampling_ratios", C.c_int32), ("upsampling_ratios", C.c_int32 * Q3C_MAX_STAGES),
        ("num_upsample_rates", C.c_int32), ("upsample_rates", C.c_int32 * Q3C_MAX_STAGES),
        ("decoder_dim", C.c_int32),
        ("num_dilations", C.c_int32), ("dilations", C.c_int32 * Q3C_MAX_STAGES),
        ("max_positions", C.c_int32),
        ("rvq_eps", C.c_float), ("rms_norm_eps", C.c_float), ("convnext_eps", C.c_float), ("snake_eps", C.c_float),
    ]


class Q3cEncoderConfig(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int32),
        ("num_filters", C.c_int32), ("kernel_size", C.c_int32), ("residual_kernel_size", C.c_int32), ("compress", C.c_int32),
        ("last_kernel_size", C.c_int32),
        ("num_ratios", C.c_int32), ("ratios", C.c_int32 * Q3C_MAX_STAGES),
        ("hidden_size", C.c_int32), ("num_attention_heads", C.c_int32), ("head_dim", C.c_int32), ("intermediate_size", C.c_int32),
        ("num_hidden_layers", C.c_int32), ("sliding_window", C.c_int32),
        ("downsample_stride", C.c_int32), ("max_positions", C.c_int32),
        ("norm_eps", C.c_float),
    ]


def make_encoder_config(ecfg, max_positions: int) -> Q3cEncoderConfig:
    """`ecfg`: qwen3_tts_amd.encoder.EncoderConfig; max_positions in rows at the encoder transformer's rate."""
    if len(ecfg.ratios) > Q3C_MAX_STAGES:
        raise ValueError(f"ratios has more than {Q3C_MAX_STAGES} entries")
    c = Q3cEncoderConfig()
    c.struct_size = C.sizeof(Q3cEncoderConfig)
    for f in ("num_filters", "kernel_size", "residual_kernel_size", "compress", "last_kernel_size", "hidden_size", "num_attention_heads",
              "head_dim", "intermediate_size", "num_hidden_layers", "sliding_window", "downsample_stride"):
        setattr(c, f, int(getattr(ecfg, f)))
    c.num_ratios = len(ecfg.ratios)
    for i, v in enumerate(ecfg.ratios):
        c.ratios[i] = int(v)
    c.max_positions = int(max_positions)
    c.norm_eps = float(ecfg.norm_eps)
    return c


class Q3cError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"q3c error {code}: {message}")
        self.code = code


_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """Load libq3c.so and declare every entry point's signature."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          f"(hipcc --offload-arch=gfx950); there is no CPU or eager fallback for the decode")
    lib = C.CDLL(LIB_PATH)
    vp, i32, i64p, fp = C.c_void_p, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_float)
    lib.q3c_create.argtypes = [C.POINTER(Q3cConfig), C.POINTER(vp)]
    lib.q3c_destroy.argtypes = [vp]
    lib.q3c_destroy.restype = None
    lib.q3c_last_error.argtypes = [vp]
    lib.q3c_last_error.restype = C.c_char_p
    lib.q3c_version.argtypes = []
    lib.q3c_version.restype = C.c_char_p
    lib.q3c_load_weight.argtypes = [vp, C.c_char_p, vp, i64p, i32]
    lib.q3c_finalize.argtypes = [vp]
    lib.q3c_workspace_bytes.argtypes = [vp, i32, i32, C.POINTER(C.c_size_t)]
    lib.q3c_output_length.argtypes = [vp, i32, i64p]
    lib.q3c_forward.argtypes = [vp, vp, i32, i32, vp, vp, C.c_size_t, vp]
    lib.q3c_forward_skip.argtypes = [vp, vp, i32, i32, i32, vp, vp, C.c_size_t, vp]
    lib.q3c_forward_scatter.argtypes = [vp, vp, i32, i32, i32, vp, vp, vp, C.c_size_t, vp]
    lib.q3c_forward_static.argtypes = [vp, vp, i32, i32, vp, vp]
    lib.q3c_rvq_dequant_f32.argtypes = [vp, vp, i32, i32, vp, vp]
    lib.q3c_rvq_encode.argtypes = [vp, vp, i32, i32, i32, vp, vp]
    lib.q3c_check_errors.argtypes = [vp, vp]
    lib.q3c_set_tap.argtypes = [vp, i32, vp, C.c_size_t]
    lib.q3c_causal_conv1d.argtypes = [vp, i32, i32, i32, i32, i32, i32, vp, vp, vp, i32, vp]
    lib.q3c_causal_tconv1d.argtypes = [vp, i32, i32, i32, i32, i32, i32, vp, vp, vp, i32, vp]
    lib.q3c_snake_beta.argtypes = [vp, C.c_int64, i32, vp, vp, C.c_float, vp, i32, vp]
    lib.q3c_residual_unit.argtypes = [vp, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, C.c_float, vp, i32, vp]
    lib.q3c_convnext_block.argtypes = [vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, C.c_float, vp, i32, vp]
    lib.q3c_transformer.argtypes = [vp, vp, i32, i32, vp, vp, C.c_size_t, vp]
    lib.q3c_wav_to_pcm16.argtypes = [vp, C.c_int64, vp, vp]
    lib.q3c_profile_enable.argtypes = [vp, i32]
    lib.q3c_debug_set_tuning.argtypes = [vp, i32]
    lib.q3c_debug_set_timeline.argtypes = [vp, vp, i32]
    lib.q3c_debug_set_operator_context.argtypes = [i32, vp, i32]
    lib.q3c_profile_read.argtypes = [vp, C.POINTER(Q3cProfileEntry), i32, C.POINTER(i32)]
    lib.q3c_stream_pool_create.argtypes = [vp, i32, C.POINTER(vp)]
    lib.q3c_stream_pool_destroy.argtypes = [vp]
    lib.q3c_stream_pool_destroy.restype = None
    lib.q3c_stream_state_bytes.argtypes = [vp, C.POINTER(C.c_size_t)]
    lib.q3c_stream_reset.argtypes = [vp, vp, i32, vp]
    lib.q3c_stream_workspace_bytes.argtypes = [vp, i32, i32, C.POINTER(C.c_size_t)]
    lib.q3c_stream_push.argtypes = [vp, vp, i32, vp, i32, vp, vp, C.c_size_t, vp]
    lib.q3c_stream_push_graph.argtypes = [vp, vp, i32, vp, i32, vp, vp, C.c_size_t, vp]
    lib.q3c_encoder_configure.argtypes = [vp, C.POINTER(Q3cEncoderConfig)]
    lib.q3c_encode_frames.argtypes = [vp, C.c_int64, i64p]
    lib.q3c_encode_workspace_bytes.argtypes = [vp, i32, i32, C.POINTER(C.c_size_t)]
    lib.q3c_encode.argtypes = [vp, vp, i32, i32, vp, vp, vp, C.c_size_t, vp]
    for name in EXPORTS:
        fn = getattr(lib, name)
        if name not in ("q3c_destroy", "q3c_last_error", "q3c_version", "q3c_stream_pool_destroy"):
            fn.restype = C.c_int
    _lib = lib
    return lib


def make_config(cfg: DecoderConfig, precision: str, max_positions: int) -> Q3cConfig:
    if precision not in PRECISION:
        raise ValueError(f"precision must be one of {sorted(PRECISION)}, got {precision!r}")
    c = Q3cConfig()
    c.struct_size = C.sizeof(Q3cConfig)
    c.frontend = 1 if cfg.frontend == FRONTEND_MEAN_EMBEDDING else 0
    c.precision = PRECISION[precision]
    for f in ("codebook_size", "num_quantizers", "num_semantic_quantizers", "codebook_dim", "rvq_out_dim", "pre_conv_kernel",
              "latent_dim", "hidden_size", "num_attention_heads", "num_key_value_heads", "head_dim", "intermediate_size",
              "num_hidden_layers", "sliding_window", "decoder_dim"):
        setattr(c, f, int(getattr(cfg, f)))
    c.io_proj = int(cfg.io_proj)
    for name, seq in (("upsampling_ratios", cfg.upsampling_ratios), ("upsample_rates", cfg.upsample_rates),
                      ("dilations", cfg.dilations)):
        if len(seq) > Q3C_MAX_STAGES:
            raise ValueError(f"{name} has more than {Q3C_MAX_STAGES} entries")
        arr = getattr(c, name)
        for i, v in enumerate(seq):
            arr[i] = int(v)
    c.num_upsampling_ratios = len(cfg.upsampling_ratios)
    c.num_upsample_rates = len(cfg.upsample_rates)
    c.num_dilations = len(cfg.dilations)
    c.max_positions = int(max_positions)
    c.rvq_eps, c.rms_norm_eps, c.convnext_eps, c.snake_eps = cfg.rvq_eps, cfg.rms_norm_eps, cfg.convnext_eps, cfg.snake_eps
    return c
