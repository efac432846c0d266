"""Does splitting a batch over two HIP streams help?  Memory-bound layers of one half could run under the MFMA-bound
layers of the other.  Times one 64 x 300-frame forward on one stream against two 32 x 300-frame forwards on two."""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qwen3_tts_amd as q3
from qwen3_tts_amd.decoder import Decoder

cfg = q3.flagship_config()
B, T = 64, 300
dec = Decoder(cfg, q3.synthetic_weights(cfg, 0), precision="bf16", max_frames_per_launch=B * 325)
lib = dec._lib
codes = q3.random_codes(cfg, B, T, 1).cuda()
n = cfg.output_length(T)
wav = torch.empty((B, 1, n), dtype=torch.float32, device="cuda")


def ws_for(b):
    t = torch.empty(dec.workspace_bytes(b, T) + 512, dtype=torch.uint8, device="cuda")
    return t, (t.data_ptr() + 255) // 256 * 256


def launch(c, w, b, ws, stream):
    rc = lib.q3c_forward(dec._ctx, C.c_void_p(c.data_ptr()), b, T, C.c_void_p(w.data_ptr()), C.c_void_p(ws[1]),
                         ws[0].numel() - 256, C.c_void_p(stream.cuda_stream))
    assert rc == 0, rc


def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


s0 = torch.cuda.current_stream()
full = ws_for(B)
print(f"one stream, {B} items: {timed(lambda: launch(codes, wav, B, full, s0)):.2f} ms")
del full
torch.cuda.empty_cache()
for parts in (2, 4):
    streams = [torch.cuda.Stream() for _ in range(parts)]
    h = B // parts
    wss = [ws_for(h) for _ in range(parts)]
    cs = [codes[i * h:(i + 1) * h].contiguous() for i in range(parts)]
    ws_ = [wav[i * h:(i + 1) * h] for i in range(parts)]

    def go():
        for i in range(parts):
            launch(cs[i], ws_[i], h, wss[i], streams[i])

    print(f"{parts} streams, {h} items each: {timed(go):.2f} ms")
    del wss
    torch.cuda.empty_cache()

# the same two streams half a forward out of phase: stream 1 starts with a 150-frame forward, so that afterwards its
# memory-bound layers fall on the other stream's MFMA-bound ones (and vice versa)
streams = [torch.cuda.Stream() for _ in range(2)]
h = B // 2
wss = [ws_for(h) for _ in range(2)]
cs = [codes[i * h:(i + 1) * h].contiguous() for i in range(2)]
ws_ = [wav[i * h:(i + 1) * h] for i in range(2)]
half_codes = cs[1][:, :, :150].contiguous()
half_wav = torch.empty((h, 1, cfg.output_length(150)), dtype=torch.float32, device="cuda")
REPS = 4


def launch_T(c, w, b, t, ws, stream):
    rc = lib.q3c_forward(dec._ctx, C.c_void_p(c.data_ptr()), b, t, C.c_void_p(w.data_ptr()), C.c_void_p(ws[1]),
                         ws[0].numel() - 256, C.c_void_p(stream.cuda_stream))
    assert rc == 0, rc


def phased(offset):
    if offset:
        launch_T(half_codes, half_wav, h, 150, wss[1], streams[1])
    for _ in range(REPS):
        for i in range(2):
            launch(cs[i], ws_[i], h, wss[i], streams[i])


for off in (False, True):
    ms = timed(lambda: phased(off), reps=2)
    frames = REPS * B * T + (h * 150 if off else 0)
    print(f"two streams, {'half a forward out of phase' if off else 'in phase'}: {ms:.1f} ms for {frames} frames = {ms / frames * B * T:.2f} ms per {B} x {T}")
