"""Mixed-length decode on the device (no copy to the host) with fixed 25-frame buckets against the adaptive granularity, same
process and device.  Usage: python tools/mixed_ab.py [--utterances 2000]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qwen3_tts_amd as q3
from qwen3_tts_amd.decoder import Decoder

ap = argparse.ArgumentParser()
ap.add_argument("--utterances", type=int, default=2000)
args = ap.parse_args()
cfg = q3.flagship_config()
dec = Decoder(cfg, q3.synthetic_weights(cfg, 0), precision="bf16", max_frames_per_launch=64 * 325)
rng = np.random.Generator(np.random.PCG64(1234))
frames = [int(round(s * cfg.frame_rate)) for s in rng.uniform(2.0, 30.0, size=args.utterances)]
g = torch.Generator().manual_seed(1234)
items = [torch.randint(0, cfg.codebook_size, (cfg.num_quantizers, t), generator=g, dtype=torch.int64) for t in frames]
audio = sum(frames) / cfg.frame_rate
dec.decode_device(items[:64])
torch.cuda.synchronize()
for rep in range(2):
    for pad in (25, None, 10, 5):
        t0 = time.perf_counter()
        dec.decode_device(items, pad_to=pad)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"pad_to {pad}: {dt * 1e3:8.1f} ms  {audio / dt:9.1f} audio-s/s", flush=True)
