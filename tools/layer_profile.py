"""Per-layer-shape timing of one bench step (HIP events around every launch).  Usage: python tools/layer_profile.py [--batch 64]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qwen3_tts_amd as q3
from qwen3_tts_amd.decoder import Decoder

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--frames", type=int, default=300)
ap.add_argument("--precision", default="bf16")
ap.add_argument("--tuning", type=int, nargs="*", default=[0], help="q3c_debug_set_tuning values to compare in one process")
args = ap.parse_args()
cfg = q3.flagship_config()
dec = Decoder(cfg, q3.synthetic_weights(cfg, 0), precision=args.precision, max_frames_per_launch=args.batch * 325)
codes = q3.random_codes(cfg, args.batch, args.frames, 1).cuda()
from qwen3_tts_amd import _lib
lib = _lib.load()
dec.forward(codes)
torch.cuda.synchronize()
for tuning in args.tuning:
    dec.set_tuning(int(tuning))
    dec.forward(codes)
    torch.cuda.synchronize()
    dec.profile(True, detail=True)
    for _ in range(2):
        dec.forward(codes)
    torch.cuda.synchronize()
    rows = sorted(dec.profile_read(), key=lambda p: -p["ms"])
    dec.profile(False)
    tot = sum(p["ms"] for p in rows)
    print(f"tuning {tuning}: total {tot / 2:.2f} ms per forward (B={args.batch}, T={args.frames})")
    for p in rows[: (40 if len(args.tuning) == 1 else 12)]:
        rate = f"{p['flops'] / p['ms'] / 1e9:8.1f} TF" if p["flops"] else f"{p['bytes'] / p['ms'] / 1e6:8.1f} GB/s"
        print(f"{p['ms'] / 2:9.3f} ms {100 * p['ms'] / tot:5.1f}%  x{p['launches'] // 2:<3d} {rate}  {p['kernel']}")
dec.set_tuning(0)
