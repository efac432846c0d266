"""The bulk attention kernels against each other and the CPU oracle: the flagship transformer stack (16 heads x 64, window
72) on random hidden states, bf16, with the matrix-core kernel (default) and the vector-ALU kernel (tuning bit 256)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import qwen3_tts_amd as q3
from qwen3_tts_amd import _lib
from qwen3_tts_amd.decoder import Decoder
import decoder_oracle as oracle

cfg = q3.flagship_config()
w = q3.synthetic_weights(cfg, 0)
dec = Decoder(cfg, w, precision="bf16")
lib = _lib.load()
for T in (1, 31, 100, 129, 300):
    x = torch.randn(2, T, cfg.latent_dim, generator=torch.Generator().manual_seed(T))
    want = oracle.transformer(cfg, x, w)
    dec.set_tuning(0)
    a = dec.transformer(x).float().cpu()
    dec.set_tuning(256)
    b = dec.transformer(x).float().cpu()
    dec.set_tuning(0)
    snr = lambda ref, y: float(10 * torch.log10((ref ** 2).sum() / ((ref - y) ** 2).sum()))
    print(f"T={T}: matrix-core kernel vs oracle {snr(want, a):.1f} dB, vector kernel vs oracle {snr(want, b):.1f} dB, "
          f"each other {snr(b, a):.1f} dB; max|y| {float(want.abs().max()):.2f}; finite {bool(torch.isfinite(a).all())}")
