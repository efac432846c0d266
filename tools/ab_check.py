"""A/B parity of a tuning switch: same codes through the decoder with q3c_debug_set_tuning(0) and (bits); prints max |diff|."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qwen3_tts_amd as q3
from qwen3_tts_amd import _lib
from qwen3_tts_amd.decoder import Decoder

bits = int(sys.argv[1])
cfg = q3.tiny_config("split_rvq", decoder_dim=1536, latent_dim=256, hidden_size=256, intermediate_size=512)
dec = Decoder(cfg, q3.synthetic_weights(cfg, 0), precision="bf16")
codes = q3.random_codes(cfg, 3, 40, seed=3).cuda()
lib = _lib.load()
dec.set_tuning(0)
a = dec.forward(codes)
dec.set_tuning(bits)
b = dec.forward(codes)
dec.set_tuning(0)
torch.cuda.synchronize()
print("max |diff|", float((a - b).abs().max()), "max |a|", float(a.abs().max()))
