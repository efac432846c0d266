"""Are the frame-rate GEMMs of a push waiting for HBM?  The transformer stack alone on 256 rows, back to back: its 218 MB of
weights stay in the 256 MB Infinity Cache between repetitions, which a real push (GBs of conv-decoder traffic in between)
does not allow.  Prints the per-layer-shape times to set beside tools/stream_layers.py's."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qwen3_tts_amd as q3
from qwen3_tts_amd.decoder import Decoder

cfg = q3.flagship_config()
dec = Decoder(cfg, q3.synthetic_weights(cfg, 0), precision="bf16")
x = torch.randn(256, 1, cfg.latent_dim, device="cuda")
for _ in range(3):
    dec.transformer(x)
torch.cuda.synchronize()
dec.profile(True, detail=True)
reps = 10
for _ in range(reps):
    dec.transformer(x)
torch.cuda.synchronize()
rows = sorted(dec.profile_read(), key=lambda p: -p["ms"])
dec.profile(False)
print(f"transformer stack alone, 256 rows, {reps} repetitions back to back: {sum(p['ms'] for p in rows) / reps * 1e3:.0f} us of kernel time each")
for p in rows[:14]:
    print(f"{p['ms'] / reps * 1e3:8.1f} us  x{p['launches'] // reps:<3d} {p['ms'] / p['launches'] * 1e3:6.1f} us each  {p['kernel']}")
