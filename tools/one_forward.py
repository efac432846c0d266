"""One warm-up and one measured forward of the flagship decoder (target of rocprofv3 runs)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qwen3_tts_amd as q3
from qwen3_tts_amd.decoder import Decoder

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--frames", type=int, default=300)
ap.add_argument("--precision", default="bf16")
ap.add_argument("--reps", type=int, default=1)
ap.add_argument("--tuning", type=int, default=0)
args = ap.parse_args()
cfg = q3.flagship_config()
dec = Decoder(cfg, q3.synthetic_weights(cfg, 0), precision=args.precision, max_frames_per_launch=args.batch * 325)
codes = q3.random_codes(cfg, args.batch, args.frames, 1).cuda()
dec.set_tuning(args.tuning)
for _ in range(1 + args.reps):
    wav = dec.forward(codes)
torch.cuda.synchronize()
print("ok", tuple(wav.shape))
