"""Per-layer-shape timing of one streaming push (HIP events around every launch).  Usage: python tools/stream_layers.py [--streams 256] [--frames 1]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qwen3_tts_amd as q3
from qwen3_tts_amd.decoder import Decoder

ap = argparse.ArgumentParser()
ap.add_argument("--streams", type=int, default=256)
ap.add_argument("--frames", type=int, default=1)
ap.add_argument("--tuning", type=int, default=0)
args = ap.parse_args()
cfg = q3.flagship_config()
dec = Decoder(cfg, q3.synthetic_weights(cfg, 0), precision="bf16")
dec.set_tuning(args.tuning)
pool = dec.open_streams(args.streams)
ids = torch.tensor(pool.open_many(args.streams), dtype=torch.int32, device="cuda")
codes = q3.random_codes(cfg, args.streams, args.frames, 1).cuda()
for _ in range(80):
    pool.push_raw(ids, codes)
torch.cuda.synchronize()
dec.profile(True, detail=True)
reps = 5
for _ in range(reps):
    pool.push_raw(ids, codes)
torch.cuda.synchronize()
rows = sorted(dec.profile_read(), key=lambda p: -p["ms"])
dec.profile(False)
tot = sum(p["ms"] for p in rows)
print(f"push of {args.streams} streams x {args.frames} frame(s): {tot / reps:.3f} ms of kernel time")
for p in rows[:45]:
    rate = f"{p['flops'] / p['ms'] / 1e9:8.1f} TF" if p["flops"] else f"{p['bytes'] / p['ms'] / 1e6:8.1f} GB/s"
    print(f"{p['ms'] / reps * 1e3:9.1f} us {100 * p['ms'] / tot:5.1f}%  x{p['launches'] // reps:<3d} {rate}  {p['kernel']}")
