"""One forward of the same number of frames cut into pieces of different length: what short pieces cost.
Usage: python tools/shape_sweep.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qwen3_tts_amd as q3
from qwen3_tts_amd.decoder import Decoder

cfg = q3.flagship_config()
dec = Decoder(cfg, q3.synthetic_weights(cfg, 0), precision="bf16", max_frames_per_launch=64 * 325)
for B, T in ((64, 300), (128, 150), (256, 75), (768, 25), (69, 277), (160, 120), (32, 300), (16, 300)):
    codes = q3.random_codes(cfg, B, T, seed=1).cuda()
    dec.forward(codes)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        dec.forward(codes)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print(f"B {B:4d} x T {T:4d} = {B * T:6d} frames: {dt * 1e3:8.2f} ms  {B * T / cfg.frame_rate / dt:9.1f} audio-s/s", flush=True)
