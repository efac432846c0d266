"""The benchmark configurations on the GPU, at the flagship architecture (hidden 1024, 16 heads x 64, 8 layers, 16 code books
of 2048 x 256, conv decoder 1536 -> 96 channels):

  * one 10 s utterance (125 frames) and the attention-window edge (73 frames): whole forward against the CPU oracle, fp32 at
    the 1e-3 / 60 dB bar, bf16 with stated per-stage floors and each decoder block's own error bounded on the oracle's input;
  * full-size property runs of the bulk batch (64 x 375 frames, chunked), the streaming configuration (256 staggered streams,
    one frame per push, 100 pushes), the 10-minute utterance (7500 frames), and streaming over 7500 frames against the oracle
    on the small architecture (positions in the thousands);
  * the install hook with the real decoder, the one-rank RCCL gather, and the bf16 sine far outside +-256 revolutions.

The oracle (oracle/decoder_oracle.py) is the CPU restatement of the decode (sibling Code2Wav.forward,
transformers/models/qwen3_omni_moe/modeling_qwen3_omni_moe.py:3672-3696); sizes follow what it finishes in seconds."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import qwen3_tts_amd as q3
from qwen3_tts_amd import _lib
from qwen3_tts_amd import decoder as D
from qwen3_tts_amd.decoder import Decoder

from gpu_util import format_report, max_abs, oracle, snr_db, stage_report

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

F32_MAX_ABS, F32_SNR_DB = 1e-3, 60.0
# bf16 at the flagship widths, against the fp32 oracle.  Early stages: SNR floors.  The random-weight conv decoder amplifies
# any perturbation by ~5.5 dB per block, so the blocks are judged one at a time on the ORACLE's input (own error <= 1 % of
# the block output's range for the transposed conv + three residual units chain) and the waveform by its RMS error against
# what PyTorch's own bf16 run of the same restatement loses.
BF16_STAGE_SNR_DB = {"rvq_out": 50.0, "transformer_in": 45.0, "transformer_out": 36.0, "upsample0": 34.0, "upsample1": 34.0,
                     "conv_in+snake": 32.0,
                     # whole-chain floors behind the amplifying blocks (measured 30.1 / 25.0 / 19.9 / 15.0 / 16.2 dB): these only
                     # catch gross breakage; BF16_BLOCK_REL below is what bounds each block's own arithmetic
                     "block0+snake": 28.0, "block1+snake": 23.0, "block2+snake": 18.0, "block3+snake": 13.0, "wav": 14.0}
# The same bf16 decode through OTHER tile shapes (a batch of one against a batch of 64: K is walked in other chunk sizes, so
# fp32 sums regroup and a few bf16 roundings flip early on): the amplifying blocks turn that into two noise realisations of the
# size each has against the oracle (16 dB); measured 18.5 dB between them.
BF16_RESHAPED_SNR_DB = 14.0
BF16_BLOCK_REL = 0.01
BF16_WAV_REL = 1.25


@pytest.fixture(scope="module")
def flagship():
    cfg = q3.flagship_config()
    w = q3.synthetic_weights(cfg, seed=0)
    return cfg, w


@pytest.fixture(scope="module")
def flagship_bf16(flagship):
    cfg, w = flagship
    return Decoder(cfg, w, device="cuda:0", precision="bf16", max_frames_per_launch=64 * 325)


def _rms(x):
    return float(np.sqrt(np.mean(np.asarray(x, dtype=np.float64) ** 2)))


# ---- the bulk attention on the matrix cores -------------------------------------------------------------------------------
BF16_TRANSFORMER_SNR_DB = 36.0       # measured 37.9-38.9 dB for both attention kernels


@pytest.mark.parametrize("T", [1, 31, 72, 73, 129, 300])
def test_flagship_transformer_bf16_both_attention_kernels_vs_oracle(flagship, flagship_bf16, T):
    """The transformer stack (8 layers, 16 heads x 64, window 72) in bf16 on random hidden states against the oracle, with the
    matrix-core attention kernel the bulk decode uses and with the vector-ALU kernel (the library's measurement switch): both
    within the same floor, lengths covering one query, partial 32-query and 128-query tiles, the window edge and a full
    chunk."""
    cfg, w = flagship
    x = torch.randn(2, T, cfg.latent_dim, generator=torch.Generator().manual_seed(100 + T))
    want = oracle.transformer(cfg, x, w).numpy()
    lib = _lib.load()
    got = {}
    try:
        for name, bits in (("matrix cores", 0), ("vector ALU", 256)):
            flagship_bf16.set_tuning(bits)
            got[name] = flagship_bf16.transformer(x).cpu().numpy()
    finally:
        flagship_bf16.set_tuning(0)
    for name, y in got.items():
        assert np.isfinite(y).all(), name
        assert snr_db(want, y) >= BF16_TRANSFORMER_SNR_DB, (name, T, snr_db(want, y))
    print(f"transformer bf16 T={T}: " + ", ".join(f"{k} {snr_db(want, v):.1f} dB" for k, v in got.items()))


@pytest.mark.parametrize("window,kv_heads", [(72, 2), (40, 4), (97, 1), (250, 2)])
def test_matrix_core_attention_grouped_heads_and_windows_vs_oracle(window, kv_heads):
    """Head size 64 with grouped key / value heads and windows on both sides of the four-tile limit (97) and at the nine-tile
    kernel's 250, two layers, bf16: the whole small decode's transformer stage against the oracle, both attention kernels."""
    cfg = q3.tiny_config(hidden_size=256, latent_dim=64, num_attention_heads=4, num_key_value_heads=kv_heads, head_dim=64,
                         intermediate_size=256, sliding_window=window)
    w = q3.synthetic_weights(cfg, seed=21)
    dec = Decoder(cfg, w, device="cuda:0", precision="bf16")
    T = 300 if window > 97 else 150
    x = torch.randn(2, T, cfg.latent_dim, generator=torch.Generator().manual_seed(window))
    want = oracle.transformer(cfg, x, w).numpy()
    lib = _lib.load()
    got = {}
    try:
        for name, bits in (("matrix cores", 0), ("vector ALU", 256)):
            dec.set_tuning(bits)
            got[name] = dec.transformer(x).cpu().numpy()
    finally:
        dec.set_tuning(0)
    print(f"window {window}, {kv_heads} kv heads: " + ", ".join(f"{k} {snr_db(want, v):.1f} dB" for k, v in got.items()))
    for name, y in got.items():
        assert np.isfinite(y).all() and snr_db(want, y) >= 38.0, (name, snr_db(want, y))
    assert not np.array_equal(got["matrix cores"], got["vector ALU"])          # the switch really selects another kernel


def test_final_conv_fold_at_32_channels():
    """The fold on the 256 x 32 fused shape (a decoder whose last block has 32 channels): equal to the separate launch."""
    cfg = q3.tiny_config(decoder_dim=512)
    dec = Decoder(cfg, q3.synthetic_weights(cfg, seed=4), device="cuda:0", precision="bf16")
    codes = q3.random_codes(cfg, 2, 40, seed=6).cuda()
    lib = _lib.load()
    try:
        dec.set_tuning(512)
        want = dec.forward(codes).cpu().numpy()
    finally:
        dec.set_tuning(0)
    got = dec.forward(codes).cpu().numpy()
    assert np.isfinite(got).all() and max_abs(want, got) <= 2e-6, max_abs(want, got)
    dec.profile(True, detail=True)
    dec.forward(codes)
    names = [p["kernel"] for p in dec.profile_read()]
    dec.profile(False)
    assert "tap_sum" in names, names                                          # the folded path is the one that ran


def test_final_conv_folded_into_the_last_unit_equals_its_own_launch(flagship, flagship_bf16):
    """The bulk bf16 decode never stores the last residual unit's output: its epilogue leaves the final conv's per-tap products
    (8 floats per row) and a diagonal-sum kernel finishes the conv.  Same bf16 inputs to the conv's products, fp32 sums in
    another order: equal to the separate final-conv launch (the library's measurement switch) to fp32 rounding, on full and
    partial tiles (two items of 125 frames: 239 445 rows each) and on a 3-frame item (one partial tile)."""
    cfg, _ = flagship
    lib = _lib.load()
    for B, T in ((2, 125), (1, 3)):
        codes = q3.random_codes(cfg, B, T, seed=40 + T).cuda()
        try:
            flagship_bf16.set_tuning(512)
            want = flagship_bf16.forward(codes).cpu().numpy()
        finally:
            flagship_bf16.set_tuning(0)
        got = flagship_bf16.forward(codes).cpu().numpy()
        assert got.shape == want.shape == (B, 1, cfg.output_length(T))
        assert np.isfinite(got).all() and max_abs(want, got) <= 2e-6, max_abs(want, got)


# ---- configs[0]: one 10 s utterance, and the window edge ---------------------------------------------------------------
@pytest.mark.parametrize("T", [125, 73])
def test_flagship_forward_vs_oracle(flagship, flagship_bf16, T):
    """fp32: every stage and the waveform against the oracle at the fp32 bar (the first check of the attention kernel at
    head_dim 64 / 16 heads and of the launch sequence at real widths).  bf16: stage floors, then every decoder block alone on
    the oracle's own stage input through the operator entry points (the launches the decode uses), then the waveform."""
    cfg, w = flagship
    codes = q3.random_codes(cfg, 1, T, seed=0 if T == 125 else T)
    dec32 = Decoder(cfg, w, device="cuda:0", precision="f32")
    taps = {}
    ref = oracle.forward(cfg, w, codes, taps)
    rows, ref, wav = stage_report(dec32, cfg, w, codes, (taps, ref))
    print(f"flagship fp32 T={T}\n" + format_report(rows))
    assert wav.shape == ref.shape == (1, 1, cfg.output_length(T))
    for name, err, snr, top in rows[:-1]:
        assert snr >= F32_SNR_DB, f"{name}: {snr:.1f} dB\n" + format_report(rows)
    assert rows[-1][1] <= F32_MAX_ABS and rows[-1][2] >= F32_SNR_DB, format_report(rows)
    dec32.close()
    del dec32

    rows16, ref16, wav16 = stage_report(flagship_bf16, cfg, w, codes, (taps, ref))
    print(f"flagship bf16 T={T}\n" + format_report(rows16))
    by = {r[0]: r for r in rows16}
    for stage, floor in BF16_STAGE_SNR_DB.items():
        assert by[stage][2] >= floor, f"{stage} below {floor} dB\n" + format_report(rows16)
    assert float(wav16.abs().max()) <= 1.0 and bool(torch.isfinite(wav16).all())
    if T != 125:
        return
    # each block alone, fed the oracle's stage input
    x = taps["conv_in"]
    for i, (_cin, _cout, r) in enumerate(cfg.block_channels):
        p = f"decoder.{i + 1}.block."
        h = D.snake_beta(x.cuda(), w[p + "0.alpha"], w[p + "0.beta"], cfg.snake_eps)
        h = D.causal_trans_conv1d(h, w[p + "1.conv.weight"], w[p + "1.conv.bias"], r, "bf16")
        for j, d in enumerate(cfg.dilations):
            h = D.residual_unit(h, w, p + f"{j + 2}.", d, cfg.snake_eps, "bf16")
        want = taps[f"block{i}"]
        err, top = max_abs(want, h), float(want.abs().max())
        print(f"block{i} alone in bf16: max|err| {err:.3e} = {err / top:.3%} of max|y| {top:.2f}, snr {snr_db(want, h):.1f} dB")
        assert err <= BF16_BLOCK_REL * top, (i, err, top)
        x = want
    # the waveform against PyTorch's own bf16 run of the restatement, on a prefix the CPU finishes quickly (bf16 convolutions
    # are slow there): the first 24 frames of the same codes (a frame never influences samples that precede it by more than
    # the fixed lookahead, so the prefix of the forward is the forward of the prefix)
    Tp = 24
    n = cfg.output_length(Tp)
    torch_bf16 = oracle.forward(cfg, w, codes[:, :, :Tp], dtype=torch.bfloat16).numpy()[0, 0]
    r = ref16.numpy()[0, 0, :n]
    e_hip, e_torch = _rms(wav16.cpu().numpy()[0, 0, :n] - r), _rms(torch_bf16 - r)
    print(f"bf16 waveform rms error over the first {Tp} frames: this library {e_hip:.4f}, PyTorch bf16 {e_torch:.4f}")
    assert e_hip <= BF16_WAV_REL * e_torch, (e_hip, e_torch)


# ---- configs[1]: the bulk batch at full size ---------------------------------------------------------------------------
def test_bulk_batch_full_size_properties(flagship, flagship_bf16):
    """64 x 375 frames, chunked 300 + 25, bf16: length 718 890 per item, finite, inside [-1, 1]; an item decodes to the same
    bits alone as in the batch (batch rows are independent), and the seam of the two chunks sits where the rule puts it."""
    cfg, _ = flagship
    dec = flagship_bf16
    codes = q3.random_codes(cfg, 64, 375, seed=1).cuda()
    wav = dec.chunked_decode(codes, 300, 25)
    assert wav.shape == (64, 1, 718890) == (64, 1, cfg.chunked_output_length(375))
    assert bool(torch.isfinite(wav).all()) and float(wav.abs().max()) <= 1.0
    assert float(wav.abs().mean()) > 1e-3                                    # not silence
    # batch rows are independent: the same item at another batch position, among other neighbours, decodes to the same bits
    perm = torch.roll(torch.arange(64), 5)
    wav_p = dec.chunked_decode(codes[perm], 300, 25)
    assert torch.equal(wav_p, wav[perm])
    # alone (a batch of one takes other tile shapes, so sums regroup: bf16 rounding noise, nothing more) and chunk by chunk
    alone = dec.chunked_decode(codes[17:18], 300, 25)
    assert snr_db(wav[17:18], alone) >= BF16_RESHAPED_SNR_DB, snr_db(wav[17:18], alone)
    n0 = cfg.output_length(300)
    first = dec.forward(codes[17:18, :, :300])                               # chunk 0 = plain forward of frames 0..299
    second = dec.forward(codes[17:18, :, 275:375])                           # chunk 1 with its 25 context frames dropped
    assert torch.equal(first[0, 0], alone[0, 0, :n0])
    assert torch.equal(second[0, 0, 25 * cfg.total_upsample:], alone[0, 0, n0:])


# ---- configs[2]: 256 staggered streams, one frame per push --------------------------------------------------------------
def test_streaming_256_streams_100_pushes(flagship, flagship_bf16):
    """256 streams at staggered positions advance one frame per push for 100 pushes: every push finite and inside [-1, 1];
    four of the streams, concatenated, against the bulk un-chunked forward of the same codes (bf16 on both sides: they
    differ by which rows share a tile and by the carried rows' rounding points)."""
    cfg, _ = flagship
    dec = flagship_bf16
    S, steps, up = 256, 100, cfg.total_upsample
    pool = dec.open_streams(S)
    ids = pool.open_many(S)
    lead = [i % 8 for i in range(S)]                                         # stream i starts i % 8 frames ahead
    total = [steps + l for l in lead]
    g = torch.Generator().manual_seed(2)
    codes = torch.randint(0, cfg.codebook_size, (S, cfg.num_quantizers, steps + 8), generator=g, dtype=torch.int64).cuda()
    watch = [0, 5, 131, 255]
    got = {k: [] for k in watch}
    for j in range(1, 8):
        sub = [i for i in range(S) if lead[i] >= j]
        frames = torch.stack([codes[i, :, j - 1:j] for i in sub])
        outs = pool.push([ids[i] for i in sub], frames)
        for i, o in zip(sub, outs):
            if i in got:
                got[i].append(o)
    for t in range(steps):
        frames = torch.stack([codes[i, :, lead[i] + t: lead[i] + t + 1] for i in range(S)])
        outs = pool.push(ids, frames, check=(t == steps - 1))
        if t % 25 == 0 or t == steps - 1:
            flat = torch.cat([o.reshape(-1) for o in outs])
            assert bool(torch.isfinite(flat).all()) and float(flat.abs().max()) <= 1.0
        for i in watch:
            got[i].append(outs[i])
    for i in watch:
        mine = torch.cat(got[i])
        ref = dec.forward(codes[i:i + 1, :, : total[i]])[0, 0]
        assert mine.shape == ref.shape == (total[i] * up - cfg.trim,)
        snr = snr_db(ref, mine)
        print(f"stream {i} ({total[i]} frames) vs bulk forward: snr {snr:.1f} dB, max|diff| {max_abs(ref, mine):.3e}")
        # one-frame pushes run another attention kernel and other tile shapes than the bulk forward: two bf16 noise
        # realisations of the same decode (see BF16_RESHAPED_SNR_DB)
        assert snr >= BF16_RESHAPED_SNR_DB
        if i == watch[0]:                 # and each is as close to the fp32 oracle as the other
            want = oracle.forward(cfg, flagship[1], codes[i:i + 1, :, : total[i]].cpu())[0, 0]
            s_stream, s_bulk = snr_db(want, mine.cpu()), snr_db(want, ref.cpu())
            print(f"stream {i} vs oracle: {s_stream:.1f} dB; bulk forward vs oracle: {s_bulk:.1f} dB")
            assert s_stream >= BF16_STAGE_SNR_DB["wav"] and s_stream >= s_bulk - 2.0
    assert pool.state_bytes_per_stream < 4 << 20
    pool.close()


# ---- configs[4]: one 10-minute utterance ------------------------------------------------------------------------------
def test_longform_7500_frames_properties(flagship, flagship_bf16):
    """7500 frames through chunked_decode (25 chunks of 300 + 25, batched): 14 386 125 samples, finite, inside [-1, 1]; chunks
    3 and 24 equal the plain forward of their frames with the context samples dropped."""
    cfg, _ = flagship
    dec = flagship_bf16
    T, up = 7500, cfg.total_upsample
    codes = q3.random_codes(cfg, 1, T, seed=3).cuda()
    wav = dec.chunked_decode(codes, 300, 25)
    assert wav.shape == (1, 1, 14386125) == (1, 1, cfg.chunked_output_length(T))
    assert bool(torch.isfinite(wav).all()) and float(wav.abs().max()) <= 1.0
    plan = cfg.chunk_plan(T, 300, 25)
    assert len(plan) == 25
    offs = np.cumsum([0] + [cfg.output_length(e - s) - c * up for s, e, c in plan])
    for k in (3, 24):                     # a batch of one chunk takes other tile shapes than the batch of 25: bf16 regrouping noise
        s, e, c = plan[k]
        piece = dec.forward(codes[:, :, s:e])[0, 0, c * up:]
        got = wav[0, 0, offs[k]:offs[k + 1]]
        assert piece.shape == got.shape and snr_db(piece, got) >= BF16_RESHAPED_SNR_DB, (k, snr_db(piece, got))
    # the 24 full chunks in another order decode to the same bits (same shapes, rows independent)
    order = [int(i) for i in torch.randperm(24, generator=torch.Generator().manual_seed(0))]
    stacked = torch.cat([codes[:, :, plan[1 + j][0]:plan[1 + j][1]] for j in order], dim=0)
    pieces = dec.forward(stacked)
    for n, j in enumerate(order[:6]):
        k = 1 + j
        assert torch.equal(pieces[n, 0, plan[k][2] * up:], wav[0, 0, offs[k]:offs[k + 1]])


def test_streaming_7500_frames_vs_oracle_small_architecture():
    """Exact streaming over 7500 frames (positions far beyond the window and the chunk size, RoPE angles in the thousands of
    radians) on the small architecture in fp32, 250 frames per push, against the oracle's un-chunked forward."""
    cfg = q3.tiny_config()
    w = q3.synthetic_weights(cfg, seed=3)
    T = 7500
    codes = q3.random_codes(cfg, 1, T, seed=3)
    want = oracle.forward(cfg, w, codes)[0, 0]
    dec = Decoder(cfg, w, device="cuda:0", precision="f32", max_positions=8192)
    pool = dec.open_streams(1)
    sid = pool.open()
    parts = [pool.push([sid], codes[:, :, a:a + 250].cuda())[0] for a in range(0, T, 250)]
    got = torch.cat(parts).cpu()
    assert got.shape == want.shape == (cfg.output_length(T),)
    tail = slice(-250 * cfg.total_upsample, None)
    print(f"7500-frame stream: max|err| {max_abs(want, got):.3e}, snr {snr_db(want, got):.1f} dB; last push alone "
          f"{max_abs(want[tail], got[tail]):.3e}")
    assert max_abs(want, got) <= F32_MAX_ABS and snr_db(want, got) >= F32_SNR_DB
    assert max_abs(want[tail], got[tail]) <= F32_MAX_ABS and snr_db(want[tail], got[tail]) >= F32_SNR_DB


# ---- install hook with the real decoder ---------------------------------------------------------------------------------
class _RefTokenizer:
    """Stand-in for the upstream tokenizer object: owns a `decoder` with the weights and a reference `decode`."""

    class _Module:
        def __init__(self, cfg, weights):
            import types
            d = {k: v for k, v in cfg.to_dict().items() if k not in ("frontend", "io_proj", "rvq_out_dim")}
            d["codebook_dim"] = 2 * cfg.codebook_dim              # upstream's meaning of the name: the RVQ in/out width
            self.config = types.SimpleNamespace(**d)
            self._sd = {"decoder." + k: v.cuda() for k, v in weights.items()}

        def state_dict(self):
            return self._sd

    def __init__(self, cfg, weights):
        self.cfg, self.w = cfg, weights
        self.decoder = self._Module(cfg, weights)

    def decode(self, encoded):
        return oracle.decode(self.cfg, self.w, [torch.as_tensor(e["audio_codes"]) for e in encoded])


def test_install_hook_end_to_end_with_the_real_decoder():
    """install_decoder on a stub model (the layout reference models.py:43-59 loads) with the real Decoder: the tokenizer's
    decode then runs the HIP path and matches the oracle at the fp32 bar; uninstall restores the reference method."""
    import types
    from qwen3_tts_amd import install
    cfg = q3.tiny_config()
    w = q3.synthetic_weights(cfg, seed=9)
    tok = _RefTokenizer(cfg, w)
    model = types.SimpleNamespace(model=types.SimpleNamespace(speech_tokenizer=tok, talker=None), name="stub")
    new = install.install_decoder(model, precision="f32")
    assert type(new).__name__ == "Decoder" and hasattr(new, "_ctx") and new.cfg == cfg and new.device.type == "cuda"
    items = [{"audio_codes": q3.random_codes(cfg, 1, T, seed=T)[0].t().contiguous()} for T in (5, 40, 301)]
    wavs, sr = model.model.speech_tokenizer.decode(items)
    want, sr2 = tok.reference_decode(items)
    assert sr == sr2 == 24000 and len(wavs) == len(want)
    for g_, r_ in zip(wavs, want):
        assert g_.dtype == np.float32 and g_.shape == r_.shape
        assert max_abs(r_, g_) <= F32_MAX_ABS and snr_db(r_, g_) >= F32_SNR_DB
    install.uninstall_decoder(model)
    assert model.model.speech_tokenizer.decode.__func__ is _RefTokenizer.decode


# ---- the gather over RCCL (one rank: the only size a one-GPU box can form) ------------------------------------------------
_RCCL_WORKER = r'''
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, {root!r})
import qwen3_tts_amd as q3
from qwen3_tts_amd.decoder import Decoder
from qwen3_tts_amd.parallel import decode_long_sharded, decode_sharded
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
cfg = q3.tiny_config()
dec = Decoder(cfg, q3.synthetic_weights(cfg, 4), device="cuda:0", precision="f32")
items = [q3.random_codes(cfg, 1, T, seed=T)[0].t().contiguous() for T in (3, 9, 2, 310, 5)]
want, _ = dec.decode(items)
for dst in (0, None):
    got, sr = decode_sharded(dec, items, dst=dst)
    assert sr == 24000 and all(np.array_equal(a, b) for a, b in zip(got, want))
codes = q3.random_codes(cfg, 1, 23, seed=7)
long = decode_long_sharded(dec, codes[0].cuda(), 5, 2, dst=0)
assert np.abs(long - dec.chunked_decode(codes, 5, 2)[0, 0].cpu().numpy()).max() <= 1e-5
dist.barrier(); dist.destroy_process_group()
print("rccl ok")
'''


def test_sharded_decode_over_rccl_one_rank(tmp_path):
    """parallel.decode_sharded / decode_long_sharded with the nccl (RCCL) backend and the real decoder: lengths all_gather,
    flat-buffer gather and the split back into utterances, equal to a direct decode."""
    script = tmp_path / "worker.py"
    script.write_text(_RCCL_WORKER.format(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29581", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0 and "rccl ok" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


# ---- bf16 sine outside the hardware instruction's +-256 revolutions -------------------------------------------------------
def test_bf16_snake_with_large_phase():
    """SnakeBeta in the bf16 form with |x * exp(alpha)| up to ~1e4 rad (1600 revolutions): v_sin_f32 alone returns 0 beyond
    256 revolutions, which would silently drop the sin^2 term for trained alphas of that size.  Inputs are bf16-exact, so the
    expectation is the fp64 formula on the same inputs; the bound is bf16 output rounding plus the phase error of an fp32
    product at that magnitude."""
    g = torch.Generator().manual_seed(21)
    C, rows = 64, 4096
    x = (torch.randn(rows, C, generator=g) * 1.5).to(torch.bfloat16).float()
    alpha = torch.log(torch.linspace(1.0, 3000.0, C))
    beta = torch.randn(C, generator=g) * 0.3
    got = D.snake_beta(x.cuda(), alpha, beta, 1e-9, "bf16").cpu().double()
    a, invb = torch.exp(alpha.double()), 1.0 / (torch.exp(beta.double()) + 1e-9)
    want = x.double() + invb * torch.sin(x.double() * a) ** 2
    phase = (x.double().abs() * a)
    assert float(phase.max()) > 5e3
    # fp32 frequency (rel 6e-8) and product rounding move the phase by <= ~2e-7 * |phase| rad -> sin^2 by twice that
    bound = (want.abs() * 2.0 ** -8 + invb * (4e-7 * phase + 2e-5) + 1e-6)
    bad = (got - want).abs() > bound
    big = phase > 2000.0
    print(f"large-phase snake: max|err| {float((got - want).abs().max()):.3e}; elements beyond 2000 rad: {int(big.sum())}, "
          f"their mean sin^2 term {float(((got - x.double()) / invb)[big].mean()):.3f}")
    assert not bool(bad.any()), f"{int(bad.sum())} of {bad.numel()} outside the bound, worst {float(((got - want).abs() - bound).max()):.3e}"
    assert float(((got - x.double()) / invb)[big].mean()) > 0.3            # the sin^2 term is alive out there (mean 1/2)


# ---- another architecture through the same code: every size is a configuration field -------------------------------------
def test_narrow_transformer_architecture_vs_oracle():
    """The 12 Hz tokenizer's own decoder is believed to use a narrower transformer than the sibling's default (hidden 512,
    intermediate 1024 around a 1024-wide latent, grouped KV heads); nothing in the library is tied to the flagship sizes:
    whole forward in fp32 against the oracle, and bf16 against the stage floors."""
    cfg = q3.DecoderConfig(hidden_size=512, intermediate_size=1024, latent_dim=1024, num_attention_heads=8, num_key_value_heads=4,
                           head_dim=64, num_hidden_layers=4, codebook_dim=256, rvq_out_dim=512)
    w = q3.synthetic_weights(cfg, seed=4)
    codes = q3.random_codes(cfg, 2, 40, seed=9)
    taps = {}
    ref = oracle.forward(cfg, w, codes, taps)
    dec = Decoder(cfg, w, device="cuda:0", precision="f32")
    rows, ref, wav = stage_report(dec, cfg, w, codes, (taps, ref))
    print("narrow transformer fp32\n" + format_report(rows))
    assert rows[-1][1] <= F32_MAX_ABS and rows[-1][2] >= F32_SNR_DB, format_report(rows)
    dec.close()
    dec16 = Decoder(cfg, w, device="cuda:0", precision="bf16")
    rows16, _, wav16 = stage_report(dec16, cfg, w, codes, (taps, ref))
    print("narrow transformer bf16\n" + format_report(rows16))
    by = {r[0]: r for r in rows16}
    for stage in ("rvq_out", "transformer_in", "transformer_out", "upsample1", "conv_in+snake"):
        assert by[stage][2] >= BF16_STAGE_SNR_DB[stage], f"{stage}\n" + format_report(rows16)
    assert by["wav"][2] >= BF16_STAGE_SNR_DB["wav"] - 2.0 and float(wav16.abs().max()) <= 1.0


# ---- a chunk's left context: rows nobody reads are not computed ------------------------------------------------------------
@pytest.mark.parametrize("precision", ["f32", "bf16"])
def test_skipped_context_rows_leave_the_kept_samples_bit_identical_small(precision):
    """q3c_forward_skip on the small architecture (partial tiles, the vector-path front end): the samples from
    skip * total_upsample on equal q3c_forward's bit for bit, for skips of one frame, a few, a chunk's context and all but one;
    chunked_decode (sibling :3686-3696: each chunk recomputes 25 context frames and drops their samples) gives the same
    bits whether or not the context's rows are skipped, and matches the oracle as before."""
    cfg = q3.tiny_config()
    w = q3.synthetic_weights(cfg, seed=3)
    dec = Decoder(cfg, w, device="cuda:0", precision=precision)
    up = cfg.total_upsample
    for B, T, skips in ((2, 40, (1, 3, 25, 39)), (1, 2, (1,)), (3, 101, (25,))):
        codes = q3.random_codes(cfg, B, T, seed=T)
        full = dec.forward(codes)
        for k in skips:
            part = dec.forward(codes, skip_frames=k)
            assert part.shape == full.shape
            assert torch.equal(part[..., k * up:], full[..., k * up:]), (precision, B, T, k)
            assert not bool(part[..., : k * up].any())          # the dropped samples come back as zeros, not as stale memory
    with pytest.raises(ValueError):
        dec.forward(q3.random_codes(cfg, 1, 4, seed=0), skip_frames=4)
    codes = q3.random_codes(cfg, 2, 345, seed=9)
    a = dec.chunked_decode(codes, 100, 25)
    dec.skip_context = False
    b = dec.chunked_decode(codes, 100, 25)
    dec.skip_context = True
    assert torch.equal(a, b)
    if precision == "f32":
        want = oracle.chunked_decode(cfg, w, codes, 100, 25)
        assert max_abs(want, a) <= F32_MAX_ABS and snr_db(want, a) >= F32_SNR_DB
    # decode() groups the (utterance, chunk) pieces by padded length AND context, so the launches' batch sizes - hence tile
    # shapes and summation orders - differ from a run without skipping: judged against the oracle, not bit for bit
    items = [q3.random_codes(cfg, 1, t, seed=t)[0] for t in (7, 130, 260, 101)]
    got, sr = dec.decode(items, chunk_size=100, left_context_size=25)
    assert sr == cfg.sample_rate
    for item, x in zip(items, got):
        want = oracle.chunked_decode(cfg, w, item[None], 100, 25)[0, 0].numpy()
        assert x.shape == want.shape
        if precision == "f32":
            assert max_abs(want, x) <= F32_MAX_ABS and snr_db(want, x) >= F32_SNR_DB
        else:       # (bf16 arithmetic is bounded by the tests above and in test_gpu_parity.py; here: every sample written)
            assert np.isfinite(x).all() and np.abs(x).max() <= 1.0 and np.corrcoef(want, x)[0, 1] > 0.9
    dec.close()


def test_skipped_context_rows_wide_blocks_fp32():
    """fp32 at the flagship's conv-decoder widths (768 / 384 / 192-channel blocks run their units as two launches in fp32, each
    with its own first row; the k = 1 conv must not fold the batch into its rows then)."""
    cfg = q3.tiny_config("split_rvq", decoder_dim=1536)
    w = q3.synthetic_weights(cfg, seed=5)
    dec = Decoder(cfg, w, device="cuda:0", precision="f32")
    up = cfg.total_upsample
    codes = q3.random_codes(cfg, 2, 31, seed=4)
    full = dec.forward(codes)
    for k in (25, 30):
        part = dec.forward(codes, skip_frames=k)
        assert torch.equal(part[..., k * up:], full[..., k * up:]), k
    dec.close()


def test_skipped_context_rows_flagship_bf16(flagship, flagship_bf16):
    """The same at the flagship widths in bf16 (fused units on the raw stream, the folded final conv, the two-launch units of
    the 768- and 384-channel blocks, 256-row tiles): a 100-frame chunk with 25 context frames, batch of 3."""
    cfg, _w = flagship
    dec = flagship_bf16
    up = cfg.total_upsample
    codes = q3.random_codes(cfg, 3, 100, seed=21)
    full = dec.forward(codes)
    part = dec.forward(codes, skip_frames=25)
    assert torch.equal(part[..., 25 * up:], full[..., 25 * up:])
    part = dec.forward(codes, skip_frames=99)
    assert torch.equal(part[..., 99 * up:], full[..., 99 * up:])
    codes = q3.random_codes(cfg, 2, 375, seed=22)
    a = dec.chunked_decode(codes, cfg.chunk_size, cfg.left_context_size)
    dec.skip_context = False
    b = dec.chunked_decode(codes, cfg.chunk_size, cfg.left_context_size)
    dec.skip_context = True
    assert a.shape[-1] == cfg.chunked_output_length(375, cfg.chunk_size, cfg.left_context_size)
    assert torch.equal(a, b)
