"""Helpers shared by the GPU parity tests: error metrics and a per-stage comparison against the oracle."""
import numpy as np
import torch

import qwen3_tts_amd as q3
from qwen3_tts_amd import _lib
import decoder_oracle as oracle


def _host(x):
    if isinstance(x, torch.Tensor):
        x = x.detach().cpu()
    return np.asarray(x, dtype=np.float64)


def snr_db(ref, got):
    ref, got = _host(ref), _host(got)
    noise = ((ref - got) ** 2).sum()
    return float("inf") if noise == 0 else 10.0 * np.log10((ref ** 2).sum() / noise)


def max_abs(ref, got):
    return float(np.abs(_host(ref) - _host(got)).max())


def stage_report(dec, cfg, w, codes, cached=None):
    """Run one forward with taps on every stage; return [(stage name, max-abs, snr dB)] against the oracle.
    `cached` = (taps, ref) of an earlier oracle run on the same codes and weights."""
    if cached is None:
        taps = {}
        ref = oracle.forward(cfg, w, codes.cpu(), taps)
    else:
        taps, ref = cached
    B, _, T = codes.shape
    stages = []
    if cfg.frontend == q3.FRONTEND_SPLIT_RVQ:
        stages.append((_lib.STAGE_RVQ_OUT, "rvq_out", taps["rvq_out"]))
    stages.append((_lib.STAGE_TRANSFORMER_IN, "transformer_in", taps["transformer_in"]))
    stages.append((_lib.STAGE_TRANSFORMER_OUT, "transformer_out", taps["transformer_out"]))
    for i in range(len(cfg.upsampling_ratios)):
        stages.append((_lib.STAGE_UPSAMPLE0 + i, f"upsample{i}", taps[f"upsample{i}"]))

    def snaked(t, prefix):      # channels-last tensor through the snake the library fuses into the producer
        return oracle.snake_beta(t.permute(0, 2, 1), w[prefix + "alpha"], w[prefix + "beta"], cfg.snake_eps).permute(0, 2, 1)

    nb = len(cfg.upsample_rates)
    stages.append((_lib.STAGE_CONV_IN, "conv_in+snake", snaked(taps["conv_in"], "decoder.1.block.0.")))
    for i in range(nb):
        prefix = f"decoder.{i + 2}.block.0." if i + 1 < nb else f"decoder.{nb + 1}."
        stages.append((_lib.STAGE_BLOCK0 + i, f"block{i}+snake", snaked(taps[f"block{i}"], prefix)))
    bufs = [(name, want, dec.set_tap(stage, want.numel())) for stage, name, want in stages]
    wav = dec.forward(codes)
    torch.cuda.synchronize()
    dec.clear_taps()
    rows = []
    for name, want, buf in bufs:
        got = buf.cpu().numpy().reshape(want.shape)
        rows.append((name, max_abs(want.numpy(), got), snr_db(want.numpy(), got), float(want.abs().max())))
    rows.append(("wav", max_abs(ref.numpy(), wav.cpu().numpy()), snr_db(ref.numpy(), wav.cpu().numpy()), 1.0))
    return rows, ref, wav


def format_report(rows):
    return "\n".join(f"  {n:18s} max|err| {e:.3e}  snr {s:7.2f} dB  (|ref|max {m:.2f})" for n, e, s, m in rows)
