"""A/B parity of a tuning switch at the flagship widths: same codes through the decoder with tuning 0 and (bits); prints max |diff|.
Usage: python tools/ab_flagship.py BITS [batch] [frames]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qwen3_tts_amd as q3
from qwen3_tts_amd.decoder import Decoder

bits = int(sys.argv[1])
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 4
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 60
cfg = q3.flagship_config()
dec = Decoder(cfg, q3.synthetic_weights(cfg, 0, profile="conditioned"), precision="bf16")
codes = q3.random_codes(cfg, batch, frames, seed=3).cuda()
dec.set_tuning(0)
a = dec.forward(codes)
torch.cuda.synchronize()
print("tuning 0 done", flush=True)
dec.set_tuning(bits)
b = dec.forward(codes)
torch.cuda.synchronize()
dec.set_tuning(0)
print("max |diff|", float((a - b).abs().max()), "max |a|", float(a.abs().max()), "finite", bool(torch.isfinite(b).all()))
sys.exit(0 if bool(torch.isfinite(b).all()) and float((a - b).abs().max()) <= 1e-2 * float(a.abs().max()) else 1)
