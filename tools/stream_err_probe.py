import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, "oracle"); sys.path.insert(0, ".")
import torch
import qwen3_tts_amd as q3
from qwen3_tts_amd.decoder import Decoder
for frontend in ("split_rvq", "mean_embedding"):
    cfg = q3.tiny_config(frontend)
    w = q3.synthetic_weights(cfg, seed=5)
    dec = Decoder(cfg, w, device="cuda:0", precision="f32")
    T = 90
    codes = q3.random_codes(cfg, 1, T, seed=5)
    ref = dec.forward(codes)[0, 0]
    pool = dec.open_streams(4)
    for sizes in ([1], [3, 1, 7, 2], [40, 50]):
        sid = pool.open()
        outs, t, k = [], 0, 0
        while t < T:
            n = min(sizes[k % len(sizes)], T - t); k += 1
            outs.append(pool.push([sid], codes[:, :, t:t + n].cuda())[0]); t += n
        got = torch.cat(outs)
        pool.close_stream(sid)
        print(frontend, sizes, float((got - ref).abs().max()))
