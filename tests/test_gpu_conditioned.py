"""bf16 - the precision the headline benchmark runs at - held to ABSOLUTE bounds through the real launch sequence.

The default synthetic weights make the conv decoder amplify any perturbation by ~5.5 dB per block (SnakeBeta with slope in
[0, 2], residual branches at gain 1/2), so a bf16 run against the fp32 oracle ends near 16 dB whoever computes it, and only
per-block bounds say anything.  The "conditioned" weight profile (qwen3_tts_amd.weights.synthetic_weights) is the same
architecture with every branch still visible but no amplification: rounding errors add up instead of multiplying, PyTorch's
own bf16 run of the restatement ends at 36 dB, and the whole path - front end, transformer, upsamplers, the four blocks with
their fused units, the folded final conv - can be held to a stated tolerance:

    bf16 waveform vs the fp32 oracle, flagship widths:  SNR >= BF16_WAV_SNR_DB,  max-abs <= BF16_WAV_MAX_ABS

and the same order between launch shapes (batch of 1 against batch of 64), between streaming and bulk, and through the
chunked decode.  The oracle is oracle/decoder_oracle.py (sibling Code2Wav.forward,
transformers/models/qwen3_omni_moe/modeling_qwen3_omni_moe.py:3672-3696; SnakeBeta :3542-3580, residual unit :3592-3608)."""
import numpy as np
import pytest
import torch

import qwen3_tts_amd as q3
from qwen3_tts_amd.decoder import Decoder

from gpu_util import format_report, max_abs, oracle, snr_db, stage_report

pytestmark = pytest.mark.gpu

F32_MAX_ABS, F32_SNR_DB = 1e-3, 60.0
# Stated bf16 tolerance of the whole decode at flagship widths on the conditioned profile (measured on MI355X: 37.4 dB / 1.8e-2 at 125
# frames, 37.4 dB / 1.9e-2 at 73; every stage >= 37.2 dB; PyTorch's own bf16 run of the restatement: 36.0 dB / 2.1e-2 on 12 frames).
BF16_WAV_SNR_DB = 36.0
BF16_WAV_MAX_ABS = 3e-2
# every tapped stage (activations a consumer's SnakeBeta has been applied to, where the library fuses it)
BF16_STAGE_SNR_DB = 36.0
# two bf16 runs of the same codes through different launch shapes / the streaming kernels: both within the tolerance of the
# oracle, so at worst 3 dB below it against each other (measured: batch of 1 against batch of 64 42.6 dB, streaming against bulk
# 37.4 dB - the pushes use the fp32-operand attention arithmetic, the bulk decode the matrix cores')
BF16_BETWEEN_RUNS_SNR_DB = 35.0


@pytest.fixture(scope="module")
def cond():
    cfg = q3.flagship_config()
    w = q3.synthetic_weights(cfg, seed=0, profile="conditioned")
    return cfg, w


@pytest.fixture(scope="module")
def cond_bf16(cond):
    cfg, w = cond
    return Decoder(cfg, w, device="cuda:0", precision="bf16", max_frames_per_launch=64 * 325)


def test_conditioned_profile_is_seeded_and_differs_from_the_default():
    cfg = q3.tiny_config()
    a = q3.synthetic_weights(cfg, seed=3, profile="conditioned")
    b = q3.synthetic_weights(cfg, seed=3, profile="conditioned")
    c = q3.synthetic_weights(cfg, seed=3)
    assert all(torch.equal(a[k], b[k]) for k in a) and set(a) == set(c)
    assert not torch.equal(a["decoder.1.block.0.alpha"], c["decoder.1.block.0.alpha"])
    with pytest.raises(ValueError):
        q3.synthetic_weights(cfg, seed=3, profile="nope")


@pytest.mark.parametrize("T", [125, 73])
def test_flagship_conditioned_forward_vs_oracle_absolute_bf16_bound(cond, cond_bf16, T):
    """One 10 s utterance (and the attention-window edge) through the whole launch sequence at flagship widths: fp32 at the
    1e-3 / 60 dB bar; bf16 at the stated absolute tolerance, at every tapped stage and at the waveform.  An error that a
    kernel adds anywhere on the path - a wrong tap, a window off by one, a rounding in the wrong place - stays visible at
    the waveform here instead of drowning in amplified noise."""
    cfg, w = cond
    codes = q3.random_codes(cfg, 1, T, seed=T)
    taps = {}
    ref = oracle.forward(cfg, w, codes, taps)
    assert float(ref.pow(2).mean().sqrt()) > 0.15                             # a waveform of ordinary level, not silence
    dec32 = Decoder(cfg, w, device="cuda:0", precision="f32")
    rows, _, wav = stage_report(dec32, cfg, w, codes, (taps, ref))
    print(f"conditioned fp32 T={T}\n" + format_report(rows))
    for name, err, snr, top in rows[:-1]:
        assert snr >= F32_SNR_DB, f"{name}: {snr:.1f} dB\n" + format_report(rows)
    assert rows[-1][1] <= F32_MAX_ABS and rows[-1][2] >= F32_SNR_DB, format_report(rows)
    dec32.close()
    del dec32

    rows16, _, wav16 = stage_report(cond_bf16, cfg, w, codes, (taps, ref))
    print(f"conditioned bf16 T={T}\n" + format_report(rows16))
    for name, err, snr, top in rows16[:-1]:
        assert snr >= BF16_STAGE_SNR_DB, f"{name}: {snr:.1f} dB\n" + format_report(rows16)
    # the run above had stage taps set (no folded final conv, every stage stored); the plain forward is the bench's path
    plain = cond_bf16.forward(codes)
    for name, y in (("tapped", wav16), ("plain", plain)):
        s, e = snr_db(ref, y), max_abs(ref, y)
        print(f"conditioned bf16 T={T} {name} forward: wav snr {s:.2f} dB, max|err| {e:.3e}")
        assert s >= BF16_WAV_SNR_DB and e <= BF16_WAV_MAX_ABS, (name, s, e)


def test_conditioned_bf16_batch_shapes_and_chunked_decode_agree(cond, cond_bf16):
    """The same utterance alone (few rows: 128-row tiles, other K chunking) and inside a batch of 64 (the bench's launch shapes),
    whole and through chunked_decode(300, 25) with its skipped context rows: all within the stated tolerance of the oracle and
    of each other."""
    cfg, w = cond
    T = 125
    codes = q3.random_codes(cfg, 64, T, seed=11)
    ref = oracle.forward(cfg, w, codes[7:8])
    alone = cond_bf16.forward(codes[7:8])
    batch = cond_bf16.forward(codes)[7:8]
    s_a, s_b, s_ab = snr_db(ref, alone), snr_db(ref, batch), snr_db(alone, batch)
    print(f"conditioned bf16: alone {s_a:.2f} dB, in a batch of 64 {s_b:.2f} dB, between them {s_ab:.2f} dB")
    assert min(s_a, s_b) >= BF16_WAV_SNR_DB and s_ab >= BF16_BETWEEN_RUNS_SNR_DB
    assert max(max_abs(ref, alone), max_abs(ref, batch)) <= BF16_WAV_MAX_ABS
    # chunked: 64 x 330 frames -> chunks (0, 300) and (275, 330) with 25 context frames; item 7 against the oracle's chunks
    long = q3.random_codes(cfg, 64, 330, seed=12)
    got = cond_bf16.chunked_decode(long, 300, 25)[7:8]
    want = oracle.chunked_decode(cfg, w, long[7:8], 300, 25)
    assert got.shape == want.shape == (1, 1, cfg.chunked_output_length(330))
    s, e = snr_db(want, got), max_abs(want, got)
    n0 = cfg.output_length(300)
    print(f"conditioned bf16 chunked 300 + 30: {s:.2f} dB, max|err| {e:.3e}; second chunk alone {snr_db(want[..., n0:], got[..., n0:]):.2f} dB")
    assert s >= BF16_WAV_SNR_DB and e <= BF16_WAV_MAX_ABS
    assert snr_db(want[..., n0:], got[..., n0:]) >= BF16_WAV_SNR_DB


def test_conditioned_bf16_streams_equal_the_bulk_forward_within_tolerance(cond, cond_bf16):
    """Streaming through the DEFAULT kernels (segmented-row launches, K/V rings, the one-frame attention step and the
    multi-frame vector-ALU attention) against the bulk forward's default path (matrix-core attention) and against the oracle,
    at flagship widths: pushes of 1, 1, 5, 24 and 50 frames, so the window of 72 is crossed and the ring wraps."""
    cfg, w = cond
    T = 81
    codes = q3.random_codes(cfg, 3, T, seed=21)
    ref = oracle.forward(cfg, w, codes[:1])[:, 0]
    bulk = cond_bf16.forward(codes)[:, 0].float()
    pool = cond_bf16.open_streams(4)
    ids = pool.open_many(3)
    outs, t = [], 0
    for n in (1, 1, 5, 24, 50):
        outs.append(torch.stack(list(pool.push(ids, codes[:, :, t:t + n].cuda()))))
        t += n
    pool.close()
    got = torch.cat(outs, dim=1).float()
    assert got.shape == bulk.shape == (3, cfg.output_length(T))
    s_bulk, s_ref = snr_db(bulk, got), snr_db(ref, got[:1])
    print(f"conditioned bf16 streaming: {s_bulk:.2f} dB against the bulk forward, {s_ref:.2f} dB against the oracle "
          f"(max|err| {max_abs(ref, got[:1]):.3e})")
    assert s_bulk >= BF16_BETWEEN_RUNS_SNR_DB and s_ref >= BF16_WAV_SNR_DB
    assert max_abs(ref, got[:1]) <= BF16_WAV_MAX_ABS


def test_flagship_fp32_one_frame_stream_equals_the_bulk_forward(cond):
    """fp32 at flagship widths (16 heads x 64): a stream advanced ONE frame per push - the attention step kernel reading the
    K/V ring in place, ring wrap after 71 frames - against the bulk forward of the same codes: same arithmetic up to the
    grouping of fp32 sums."""
    cfg, w = cond
    T = 80
    codes = q3.random_codes(cfg, 2, T, seed=31)
    dec = Decoder(cfg, w, device="cuda:0", precision="f32", max_positions=256)
    bulk = dec.forward(codes)[:, 0]
    pool = dec.open_streams(2)
    ids = pool.open_many(2)
    outs = [torch.stack(list(pool.push(ids, codes[:, :, t:t + 1].cuda()))) for t in range(T)]
    got = torch.cat(outs, dim=1)
    assert got.shape == bulk.shape
    e = max_abs(bulk, got)
    print(f"flagship fp32 one-frame stream vs bulk: max|diff| {e:.3e}, {snr_db(bulk, got):.1f} dB")
    assert e <= 2e-5, e                                                        # measured 4.3e-6
    pool.close()
    dec.close()


def test_decoder_from_a_local_safetensors_checkpoint_matches_the_oracle(tmp_path):
    """Weights written to <model dir>/speech_tokenizer/{config.json, model.safetensors} (fp32, key prefix "decoder."), loaded
    with Decoder.from_local - never from a live state dict - and decoded: fp32 at the 1e-3 / 60 dB bar against the oracle on the
    same tensors; a bf16-stored checkpoint within the bf16 tolerance; warm-up runs; hub names are refused."""
    from qwen3_tts_amd import checkpoint
    cfg = q3.tiny_config("split_rvq", sliding_window=40)
    w = q3.synthetic_weights(cfg, seed=5, profile="conditioned")
    root = tmp_path / "Qwen3-TTS-12Hz-0.6B-Base"
    checkpoint.save_local(str(root / "speech_tokenizer"), cfg, w)
    dec = Decoder.from_local(str(root), device="cuda:0", precision="f32")
    assert dec.cfg == cfg
    assert dec.warmup(batch=2, frames=7) > 0.0
    codes = q3.random_codes(cfg, 2, 61, seed=8)
    ref = oracle.forward(cfg, w, codes)
    got = dec.forward(codes)
    print(f"from_local fp32: max|err| {max_abs(ref, got):.3e}, {snr_db(ref, got):.1f} dB")
    assert max_abs(ref, got) <= F32_MAX_ABS and snr_db(ref, got) >= F32_SNR_DB
    wavs, sr = dec.decode([codes[0].t(), {"audio_codes": codes[1, :, :30]}])
    assert sr == 24000 and np.array_equal(wavs[0], got[0, 0].cpu().numpy()) and wavs[1].shape == (cfg.output_length(30),)
    dec.close()
    checkpoint.save_local(str(tmp_path / "bf16" / "tok"), cfg, w, dtype=torch.bfloat16)
    dec16 = Decoder.from_local(str(tmp_path / "bf16"), device="cuda:0", precision="bf16")
    w16 = {k: v.to(torch.bfloat16).float() for k, v in w.items()}          # what the file holds
    got16 = dec16.forward(codes)
    s = snr_db(oracle.forward(cfg, w16, codes), got16)
    print(f"from_local bf16 checkpoint, bf16 decode: {s:.1f} dB")
    assert s >= BF16_WAV_SNR_DB - 1.0
    with pytest.raises(FileNotFoundError, match="hub names"):
        Decoder.from_local("Qwen/Qwen3-TTS-12Hz-0.6B-Base")
