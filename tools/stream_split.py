"""Does a frame-rate push gain from being issued as G groups of streams on G HIP streams (the latency-bound transformer
launches of one group beside the conv decoder of another)?  256 streams x 1 frame, flagship architecture, bf16."""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qwen3_tts_amd as q3
from qwen3_tts_amd.decoder import Decoder

ap = argparse.ArgumentParser()
ap.add_argument("--streams", type=int, default=256)
ap.add_argument("--groups", type=int, nargs="*", default=[1, 2, 4])
ap.add_argument("--steps", type=int, default=200)
args = ap.parse_args()
cfg = q3.flagship_config()
dev = torch.device("cuda:0")
dec = Decoder(cfg, q3.synthetic_weights(cfg, 0), device=dev, precision="bf16", max_positions=8192)
lib = dec._lib
S = args.streams
pool = dec.open_streams(S)
ids = pool.open_many(S)
codes = [q3.random_codes(cfg, S, 1, seed=2 + j).to(dev) for j in range(8)]
out = torch.empty((S, cfg.total_upsample), dtype=torch.float32, device=dev)
for G in args.groups:
    per = S // G
    need = C.c_size_t()
    dec._check(lib.q3c_stream_workspace_bytes(pool._pool, per, 1, C.byref(need)))
    ws = [torch.empty(need.value + 256, dtype=torch.uint8, device=dev) for _ in range(G)]
    streams = [torch.cuda.Stream(dev) for _ in range(G)]
    idg = [torch.tensor(ids[g * per:(g + 1) * per], dtype=torch.int32, device=dev) for g in range(G)]
    cg = [[c[g * per:(g + 1) * per].contiguous() for g in range(G)] for c in codes]
    og = [out[g * per:(g + 1) * per] for g in range(G)]

    def step(j):
        for g in range(G):
            base = (ws[g].data_ptr() + 255) // 256 * 256
            dec._check(lib.q3c_stream_push(pool._pool, C.c_void_p(idg[g].data_ptr()), per, C.c_void_p(cg[j % 8][g].data_ptr()), 1,
                                           C.c_void_p(og[g].data_ptr()), C.c_void_p(base), need.value, C.c_void_p(streams[g].cuda_stream)))

    torch.cuda.synchronize(dev)
    for j in range(10):
        step(j)
    torch.cuda.synchronize(dev)
    lat = []
    for j in range(args.steps):
        t0 = time.perf_counter()
        step(j)
        torch.cuda.synchronize(dev)
        lat.append((time.perf_counter() - t0) * 1e3)
    lat = np.asarray(lat)
    print(f"{G} group(s) of {per} streams: p50 {np.percentile(lat, 50):.3f} ms, p99 {np.percentile(lat, 99):.3f} ms; finite {bool(torch.isfinite(out).all())}")
