"""Per-kernel time of one tokenizer encode (64 x 30 s at Mimi sizes, bf16)."""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qwen3_tts_amd as q3
from qwen3_tts_amd.decoder import Decoder

E = importlib.import_module("qwen3-tts_amd.encoder")
ecfg, cfg = E.EncoderConfig(), q3.flagship_config()
w = {**q3.synthetic_weights(cfg, 0), **E.synthetic_encoder_weights(ecfg, cfg.codebook_dim, 3)}
dec = Decoder(cfg, w, precision="bf16", encoder=ecfg)
audio = (torch.randn(64, 720000, generator=torch.Generator().manual_seed(1)) * 0.3).cuda()
dec.encode(audio)
dec._check(dec._lib.q3c_profile_enable(dec._ctx, 2))
dec.encode(audio)
torch.cuda.synchronize()
rows = sorted(dec.profile_read(), key=lambda p: -p["ms"])
tot = sum(p["ms"] for p in rows)
print(f"encode 64 x 30 s: {tot:.2f} ms of kernel time")
for p in rows[:30]:
    rate = f"{p['tflops']:.1f} TF" if p.get("tflops") else (f"{p['gbs']:.0f} GB/s" if p.get("gbs") else "")
    print(f"  {p['ms']:8.3f} ms  {100 * p['ms'] / tot:5.1f}%  x{p['launches']:<3d} {rate:>12s}  {p['kernel']}")
