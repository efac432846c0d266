"""The tokenizer's ENCODE side on the GPU (q3c_encode through the C ABI) against the CPU oracle and the sibling's stored outputs.

Bar: the embeddings the quantiser sees within the fp32 bar (max-abs <= 1e-4 of unit-scale values; measured a few 1e-6); the
codes are an argmin over distances, so a frame whose two nearest rows are closer than the summation noise may fall either way
and the rest of that frame's residual chain then differs: >= 99 % of the codes equal, and every differing FIRST layer of a
frame must be a near-tie under the oracle's own distances.  bf16: embeddings >= 30 dB, codes mostly equal (stated below)."""
import importlib

import numpy as np
import pytest
import torch

import qwen3_tts_amd as q3
from qwen3_tts_amd.decoder import Decoder

from gpu_util import max_abs, snr_db
from test_oracle_golden import encoder_fixture

import encoder_oracle as eo

pytestmark = pytest.mark.gpu
E = importlib.import_module("qwen3-tts_amd.encoder")


@pytest.mark.parametrize("name", ["encoder_tiny", "encoder_wide"])
def test_encode_fp32_vs_sibling_fixture(golden, name):
    meta, a = golden(name)
    ecfg, cfg, w = encoder_fixture(meta)
    dec = Decoder(cfg, w, device="cuda:0", precision="f32", encoder=ecfg)
    for L in meta["lengths"]:
        audio = torch.from_numpy(a[f"audio_{L}"])
        codes, emb = dec.encode(audio, return_embeddings=True)
        want_emb = np.transpose(a[f"emb_{L}"], (0, 2, 1))
        assert codes.shape == a[f"codes_{L}"].shape and codes.dtype == torch.int64
        assert max_abs(want_emb, emb.cpu().numpy()) <= 1e-4, (L, max_abs(want_emb, emb.cpu().numpy()))
        agree = float((codes.cpu().numpy() == a[f"codes_{L}"]).mean())
        assert agree >= 0.99, (L, agree)
        print(f"{name} L={L}: embeddings max|err| {max_abs(want_emb, emb.cpu().numpy()):.2e}, codes equal {100 * agree:.2f} %")


def test_encode_at_mimi_sizes_vs_oracle():
    """The sizes of the sibling's default Mimi encoder (64 filters, strides 4 5 6 8, 512 wide, 8 layers x 8 heads x 64, window
    250, hop 1920) with the flagship decoder's code books, on 2 x 1.3 s of noise whose length no stride divides: fp32
    embeddings and codes against the oracle; bf16 embeddings by SNR.  First run of the long-window attention (fewer
    queries per block so that 250 + keys fit the LDS) and of every encoder layer at real widths."""
    ecfg = E.EncoderConfig()
    cfg = q3.flagship_config()
    w = {**q3.synthetic_weights(cfg, seed=0), **E.synthetic_encoder_weights(ecfg, cfg.codebook_dim, seed=3)}
    L = 31337
    audio = torch.randn(2, L, generator=torch.Generator().manual_seed(5)) * 0.3
    want_codes, want_emb = eo.encode(ecfg, cfg, w, audio[:, None, :])
    want_emb = want_emb.permute(0, 2, 1).numpy()
    dec = Decoder(cfg, w, device="cuda:0", precision="f32", encoder=ecfg)
    codes, emb = dec.encode(audio, return_embeddings=True)
    assert codes.shape == (2, cfg.num_quantizers, ecfg.frames(L)) == tuple(want_codes.shape)
    err = max_abs(want_emb, emb.cpu().numpy())
    agree = float((codes.cpu() == want_codes).float().mean())
    print(f"mimi-size encoder fp32: embeddings max|err| {err:.2e} (max|e| {np.abs(want_emb).max():.2f}), codes equal {100 * agree:.2f} %")
    assert err <= 1e-4 * max(1.0, float(np.abs(want_emb).max())) and agree >= 0.99
    # the same codes decode (fp32) to a waveform of the right length: the two sides of the tokenizer fit together
    wav = dec.forward(codes)
    assert wav.shape == (2, 1, cfg.output_length(codes.shape[2])) and bool(torch.isfinite(wav).all())
    dec16 = Decoder(cfg, w, device="cuda:0", precision="bf16", encoder=ecfg)
    codes16, emb16 = dec16.encode(audio, return_embeddings=True)
    snr = snr_db(want_emb, emb16.cpu().numpy())
    agree16 = float((codes16.cpu() == want_codes).float().mean())
    print(f"mimi-size encoder bf16: embeddings {snr:.1f} dB, codes equal {100 * agree16:.1f} % (first layer {100 * float((codes16.cpu()[:, 0] == want_codes[:, 0]).float().mean()):.1f} %)")
    assert snr >= 30.0 and agree16 >= 0.5
    # bf16 runs the window-250 attention on the matrix cores (nine key tiles per wave); the vector-ALU kernel (the library's
    # measurement switch) must land at the same distance from the oracle
    from qwen3_tts_amd import _lib
    lib = _lib.load()
    try:
        dec16.set_tuning(256)
        _, emb_v = dec16.encode(audio, return_embeddings=True)
    finally:
        dec16.set_tuning(0)
    snr_v = snr_db(want_emb, emb_v.cpu().numpy())
    print(f"mimi-size encoder bf16, vector-ALU attention: embeddings {snr_v:.1f} dB; between the two kernels {snr_db(emb_v.cpu().numpy(), emb16.cpu().numpy()):.1f} dB")
    assert snr_v >= 30.0 and abs(snr_v - snr) <= 3.0


def test_rvq_search_kernels_agree():
    """The code-book search runs on the fp32 matrix op when the code books are 256 wide (32 frames per block, four waves
    splitting the 2048 codes); the vector-ALU kernel (the library's measurement switch) is the same exact argmin in another
    summation order.  3 x 4.1 s of noise = 3 x 52 frames (four full blocks and a ragged fifth): the two must pick the same
    code nearly everywhere (a near-tie may flip, and then the later code books of that frame follow)."""
    from qwen3_tts_amd import _lib
    ecfg = E.EncoderConfig()
    cfg = q3.flagship_config()
    w = {**q3.synthetic_weights(cfg, seed=0), **E.synthetic_encoder_weights(ecfg, cfg.codebook_dim, seed=3)}
    audio = torch.randn(3, 98765, generator=torch.Generator().manual_seed(11)) * 0.3
    dec = Decoder(cfg, w, device="cuda:0", precision="f32", encoder=ecfg)
    lib = _lib.load()
    a = dec.encode(audio).cpu()
    try:
        dec.set_tuning(1 << 20)
        b = dec.encode(audio).cpu()
    finally:
        dec.set_tuning(0)
    assert a.shape == b.shape == (3, cfg.num_quantizers, ecfg.frames(98765))
    first = float((a[:, 0] == b[:, 0]).float().mean())
    frames_equal = float((a == b).all(dim=1).float().mean())
    print(f"matrix-op vs vector-ALU search: first code book {100 * first:.2f} %, frames with all codes equal {100 * frames_equal:.2f} %")
    assert int(a.min()) >= 0 and int(a.max()) < cfg.codebook_size
    assert first >= 0.99 and frames_equal >= 0.97


def test_encode_argument_errors(golden):
    meta, a = golden("encoder_tiny")
    ecfg, cfg, w = encoder_fixture(meta)
    plain = Decoder(cfg, w, device="cuda:0", precision="f32")
    with pytest.raises(RuntimeError):
        plain.encode(torch.zeros(1, 100))
    dec = Decoder(cfg, w, device="cuda:0", precision="f32", encoder=ecfg)
    with pytest.raises(ValueError):
        dec.encode(torch.zeros(1, 2, 100))
    missing = {k: v for k, v in w.items() if k != "downsample.conv.weight"}
    with pytest.raises(RuntimeError, match="downsample"):
        Decoder(cfg, missing, device="cuda:0", precision="f32", encoder=ecfg)
