"""Mixed-length decode, same device and process: the launches alone (samples left on the GPU) against the whole `decode`
(samples on the host), whose copies run under the later launches.  Usage: python tools/mixed_overlap.py [utterances]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qwen3_tts_amd as q3
from qwen3_tts_amd.decoder import Decoder, to_host

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
cfg = q3.flagship_config()
dec = Decoder(cfg, q3.synthetic_weights(cfg, 0), precision="bf16", max_frames_per_launch=64 * 325)
rng = np.random.Generator(np.random.PCG64(1234))
frames = [int(round(s * cfg.frame_rate)) for s in rng.uniform(2.0, 30.0, size=n)]
g = torch.Generator().manual_seed(1234)
items = [torch.randint(0, cfg.codebook_size, (cfg.num_quantizers, t), generator=g, dtype=torch.int64) for t in frames]
audio = sum(frames) / cfg.frame_rate
dec.decode(items[:64])
torch.cuda.synchronize()
for rep in range(2):
    t0 = time.perf_counter()
    out, starts, lengths = dec.decode_device(items)
    torch.cuda.synchronize()
    t_dev = time.perf_counter() - t0
    t0 = time.perf_counter()
    host = to_host(out)
    t_copy = time.perf_counter() - t0
    del out
    t0 = time.perf_counter()
    wavs, _ = dec.decode(items)
    t_all = time.perf_counter() - t0
    assert all(np.array_equal(w, host[s:s + m]) for w, s, m in zip(wavs, starts, lengths)), "overlapped copy differs from the plain one"
    print(f"{n} utterances, {audio:.0f} s of audio: launches alone {t_dev * 1e3:.0f} ms ({audio / t_dev:.0f} audio-s/s), copy afterwards "
          f"{t_copy * 1e3:.0f} ms, decode() with the copies under the launches {t_all * 1e3:.0f} ms ({audio / t_all:.0f} audio-s/s)", flush=True)
