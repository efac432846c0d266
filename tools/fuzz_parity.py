"""Randomised differential run: the HIP decode against the CPU oracle over seeded random small architectures, batch sizes,
lengths, chunk plans and push plans.  Every case checks, in fp32 (max-abs <= 1e-3, SNR >= 60 dB) and in bf16 on the
conditioned weight profile (SNR >= 30 dB at these small widths), the un-chunked forward, chunked_decode with a random
(chunk, context), the ragged `decode` wrapper and a stream advanced by random push sizes.  Channel counts stay multiples
of 8 (the library's requirement: 16-byte rows in bf16).
`--encode N` adds N cases of the tokenizer's encode side: random small encoders (strides, widths, heads, windows), batch
sizes and audio lengths that no stride divides; fp32 embeddings within 1e-4 of the oracle's and >= 99 % of the codes equal.
Prints one line per case and a summary; exits 1 at the first violation.

    python tools/fuzz_parity.py [--cases 60] [--encode 0] [--seed 0]
"""
import argparse
import os
import random
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import decoder_oracle as oracle  # noqa: E402
import qwen3_tts_amd as q3  # noqa: E402
from qwen3_tts_amd.decoder import Decoder  # noqa: E402

F32_MAX_ABS, F32_SNR_DB, BF16_SNR_DB = 1e-3, 60.0, 30.0


def snr_db(ref, got):
    ref, got = np.asarray(ref, dtype=np.float64), np.asarray(got, dtype=np.float64)
    noise = float(np.sum((ref - got) ** 2))
    return 200.0 if noise == 0.0 else 10.0 * np.log10(float(np.sum(ref ** 2)) / noise + 1e-300)


def random_config(rng: random.Random):
    frontend = rng.choice(["split_rvq", "split_rvq", "mean_embedding"])
    heads, kv = rng.choice([(4, 4), (4, 2), (4, 1), (2, 2), (8, 4)])
    head_dim = rng.choice([16, 32, 64])
    hidden = rng.choice([64, 96, 128])
    over = dict(num_attention_heads=heads, num_key_value_heads=kv, head_dim=head_dim, hidden_size=hidden,
                latent_dim=hidden if frontend == "mean_embedding" else rng.choice([64, 96]),
                intermediate_size=rng.choice([96, 128, 192]), num_hidden_layers=rng.choice([1, 2, 3]),
                sliding_window=rng.choice([3, 5, 17, 72]), decoder_dim=rng.choice([128, 256, 384]),
                num_quantizers=rng.choice([2, 4, 8]), codebook_size=rng.choice([16, 32, 100]),
                codebook_dim=rng.choice([8, 16, 32]), rvq_out_dim=rng.choice([16, 32, 48]))
    over["num_semantic_quantizers"] = rng.choice([1, min(2, over["num_quantizers"] - 1)]) if over["num_quantizers"] > 1 else 1
    return q3.tiny_config(frontend, **over)


def check(tag, want, got, precision, fails):
    want, got = np.asarray(want), np.asarray(got)
    if want.shape != got.shape:
        fails.append(f"{tag}: shape {got.shape} != {want.shape}")
        return
    if want.size == 0:
        return
    err, snr = float(np.abs(want - got).max()), snr_db(want, got)
    ok = (err <= F32_MAX_ABS and snr >= F32_SNR_DB) if precision == "f32" else (np.isfinite(got).all() and snr >= BF16_SNR_DB)
    if not ok:
        fails.append(f"{tag} ({precision}): max|err| {err:.3e}, snr {snr:.1f} dB")


def one_case(i, rng, verbose=True):
    cfg = random_config(rng)
    B, T = rng.randint(1, 4), rng.choice([1, 2, 3, rng.randint(4, 40), rng.randint(41, 90)])
    seed = rng.randint(0, 10 ** 6)
    codes = q3.random_codes(cfg, B, T, seed=seed)
    chunk, ctx = rng.randint(2, 40), rng.randint(0, 12)
    lens = [rng.randint(1, T) for _ in range(B)]
    lens = [n + 1 if (n == cfg.num_quantizers and n < T) else n for n in lens]      # a square matrix has no orientation
    fails, t0 = [], time.time()
    for precision in ("f32", "bf16"):
        w = q3.synthetic_weights(cfg, seed=seed % 97, profile="chaotic" if precision == "f32" else "conditioned")
        dec = Decoder(cfg, w, device="cuda:0", precision=precision, max_positions=256)
        ref = oracle.forward(cfg, w, codes)
        check("forward", ref, dec.forward(codes).cpu(), precision, fails)
        check("chunked", oracle.chunked_decode(cfg, w, codes, chunk, ctx), dec.chunked_decode(codes, chunk, ctx).cpu(), precision, fails)
        items = [codes[b, :, :n].t().contiguous() for b, n in enumerate(lens) if n != cfg.num_quantizers]      # [T_i, Q]
        wavs, sr = dec.decode(items)
        want_items, _ = oracle.decode(cfg, w, items)
        for b, (a, g) in enumerate(zip(want_items, wavs)):
            check(f"decode item {b} (lengths {lens})", a, g, precision, fails)
        # a stream advanced by random push sizes equals the un-chunked forward
        pool = dec.open_streams(2)
        sid, t, parts = pool.open(), 0, []
        while t < T:
            n = min(rng.choice([1, 1, 2, 3, 7, 25]), T - t)
            parts.append(pool.push([sid], codes[:1, :, t:t + n].cuda())[0])
            t += n
        check("stream", ref[0, 0], torch.cat(parts).cpu(), precision, fails)
        pool.close()
        del dec
    if verbose or fails:
        print(f"case {i}: {cfg.frontend}, heads {cfg.num_attention_heads}/{cfg.num_key_value_heads} x {cfg.head_dim}, hidden {cfg.hidden_size}, "
              f"layers {cfg.num_hidden_layers}, window {cfg.sliding_window}, decoder_dim {cfg.decoder_dim}, Q {cfg.num_quantizers}, "
              f"B {B}, T {T}, chunk {chunk}+{ctx}: {'ok' if not fails else 'FAIL'} ({time.time() - t0:.1f} s)", flush=True)
    for f in fails:
        print("   ", f, flush=True)
    return not fails


def encode_case(i, rng, verbose=True):
    import importlib
    import encoder_oracle as eo
    E = importlib.import_module("qwen3-tts_amd.encoder")
    heads, head_dim = rng.choice([(2, 16), (4, 16), (2, 32), (4, 32), (2, 64)])
    ecfg = E.EncoderConfig(num_filters=rng.choice([16, 32]), ratios=tuple(rng.choice([(2, 3), (4, 2), (3, 2, 2), (2, 2, 3), (5, 2)])),
                           hidden_size=heads * head_dim, num_attention_heads=heads, head_dim=head_dim,
                           intermediate_size=rng.choice([64, 128]), num_hidden_layers=rng.choice([1, 2]),
                           sliding_window=rng.choice([3, 9, 250]))
    cfg = q3.tiny_config("split_rvq", num_quantizers=rng.choice([2, 4, 8]), codebook_size=rng.choice([16, 64]),
                         codebook_dim=rng.choice([8, 16]), rvq_out_dim=ecfg.hidden_size)
    seed = rng.randint(0, 10 ** 6)
    w = {**q3.synthetic_weights(cfg, seed=seed % 89), **E.synthetic_encoder_weights(ecfg, cfg.codebook_dim, seed=seed % 83)}
    B, L = rng.randint(1, 3), rng.choice([rng.randint(1, ecfg.hop), rng.randint(ecfg.hop + 1, 40 * ecfg.hop), rng.randint(40 * ecfg.hop, 120 * ecfg.hop)])
    audio = torch.randn(B, L, generator=torch.Generator().manual_seed(seed)) * 0.3
    t0 = time.time()
    want_codes, want_emb = eo.encode(ecfg, cfg, w, audio[:, None, :])
    want_emb = want_emb.permute(0, 2, 1).numpy()
    dec = Decoder(cfg, w, device="cuda:0", precision="f32", encoder=ecfg)
    codes, emb = dec.encode(audio, return_embeddings=True)
    fails = []
    if tuple(codes.shape) != tuple(want_codes.shape):
        fails.append(f"codes shape {tuple(codes.shape)} != {tuple(want_codes.shape)}")
    else:
        err = float(np.abs(want_emb - emb.cpu().numpy()).max())
        agree = float((codes.cpu() == want_codes).float().mean())
        if err > 1e-4 * max(1.0, float(np.abs(want_emb).max())):
            fails.append(f"embeddings max|err| {err:.3e}")
        if agree < 0.99 and codes.numel() * (1 - agree) > 2:        # a near-tie or two among a handful of frames is not a failure
            fails.append(f"codes equal {100 * agree:.2f} %")
    if verbose or fails:
        print(f"encode case {i}: filters {ecfg.num_filters}, strides {ecfg.ratios}, {heads} x {head_dim}, layers {ecfg.num_hidden_layers}, "
              f"window {ecfg.sliding_window}, Q {cfg.num_quantizers} x {cfg.codebook_size}, B {B}, L {L} ({ecfg.frames(L)} frames): "
              f"{'ok' if not fails else 'FAIL'} ({time.time() - t0:.1f} s)", flush=True)
    for f in fails:
        print("   ", f, flush=True)
    return not fails


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=60)
    ap.add_argument("--encode", type=int, default=0)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    rng = random.Random(args.seed)
    for i in range(args.cases):
        if not one_case(i, rng):
            sys.exit(1)
    for i in range(args.encode):
        if not encode_case(i, rng):
            sys.exit(1)
    print(f"{args.cases} decode cases and {args.encode} encode cases ok (seed {args.seed})")


if __name__ == "__main__":
    main()
