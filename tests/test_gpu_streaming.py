"""Streaming decode on the GPU: the concatenated pushes of a stream must equal the un-chunked forward of all its frames
(sibling Code2Wav.forward, modeling_qwen3_omni_moe.py:3672-3684), whatever the push sizes and whichever other streams
share the launches.  fp32 bar: max-abs <= 1e-3 and SNR >= 60 dB against the CPU oracle (the same bar as the bulk path).
Against the library's own un-chunked forward a stream differs by which rows share a tile and, for one-frame pushes, by the
order in which the attention sums are formed (another kernel serves them): rounding noise of fp32 sums, which the decoder
amplifies to at most a few 1e-5 at the waveform (measured 2.2e-5); SAME_PATH_TOL bounds it."""

SAME_PATH_TOL = 5e-5
import numpy as np
import pytest
import torch

import qwen3_tts_amd as q3
from qwen3_tts_amd import _lib
from qwen3_tts_amd.decoder import Decoder

from gpu_util import max_abs, oracle, snr_db

pytestmark = pytest.mark.gpu


def _decoder(cfg, precision="f32", seed=0, **kw):
    w = q3.synthetic_weights(cfg, seed=seed)
    return Decoder(cfg, w, device="cuda:0", precision=precision, **kw), w


def _stream_all(pool, sid, codes, sizes):
    """Push codes [Q, T] of one stream in pieces of the given sizes (cycled); returns the concatenated samples."""
    out, t, i = [], 0, 0
    T = codes.shape[-1]
    while t < T:
        n = min(sizes[i % len(sizes)], T - t)
        out.append(pool.push([sid], codes[None, :, t:t + n])[0])
        t += n
        i += 1
    return torch.cat(out)


@pytest.mark.parametrize("frontend", ["split_rvq", "mean_embedding"])
def test_single_stream_equals_unchunked_forward(frontend):
    cfg = q3.tiny_config(frontend)
    dec, w = _decoder(cfg)
    T = 90
    codes = q3.random_codes(cfg, 1, T, seed=5)
    ref = dec.forward(codes)[0, 0]
    want = oracle.forward(cfg, w, codes)[0, 0]
    pool = dec.open_streams(4)
    for sizes in ([1], [3, 1, 7, 2], [40, 50]):
        sid = pool.open()
        got = _stream_all(pool, sid, codes[0].cuda(), sizes)
        pool.close_stream(sid)
        assert got.shape == ref.shape == (cfg.output_length(T),)
        assert max_abs(got, ref) <= SAME_PATH_TOL, (sizes, max_abs(got, ref))
        assert max_abs(got.cpu(), want) <= 1e-3 and snr_db(want, got.cpu()) >= 60.0


def test_six_hundred_frames_one_frame_at_a_time():
    """Longer than the attention window, the chunk size of chunked_decode and several tiles of every layer."""
    cfg = q3.tiny_config("split_rvq")
    dec, w = _decoder(cfg)
    T = 600
    codes = q3.random_codes(cfg, 1, T, seed=2)
    ref = dec.forward(codes)[0, 0]
    pool = dec.open_streams(1)
    sid = pool.open()
    got = _stream_all(pool, sid, codes[0].cuda(), [1])
    assert got.shape == ref.shape
    assert max_abs(got, ref) <= SAME_PATH_TOL
    want = oracle.forward(cfg, w, codes)[0, 0]
    assert max_abs(got.cpu(), want) <= 1e-3 and snr_db(want, got.cpu()) >= 60.0


@pytest.mark.parametrize("precision", ["f32", "bf16"])
def test_one_frame_pushes_at_head_size_64_with_grouped_heads(precision):
    """The one-frame attention kernel at the flagship head size (its compile-time-sized form: 16-byte value pieces over key
    groups, table rows fetched ahead), with grouped key/value heads, over more frames than the key ring holds (wrap-around):
    fp32 against the oracle at the fp32 bar; bf16 against the library's bulk forward (which rounds the same operands)."""
    cfg = q3.tiny_config("split_rvq", head_dim=64, num_attention_heads=4, num_key_value_heads=2, sliding_window=72)
    dec, w = _decoder(cfg, precision=precision, seed=7)
    T = 170
    codes = q3.random_codes(cfg, 2, T, seed=11)
    ref = dec.forward(codes)[:, 0]
    pool = dec.open_streams(2)
    ids = [pool.open(), pool.open()]
    got = torch.cat([torch.stack(pool.push(ids, codes[:, :, t:t + 1].cuda())) for t in range(T)], dim=1)
    assert got.shape == ref.shape
    if precision == "f32":
        want = oracle.forward(cfg, w, codes)[:, 0]
        assert max_abs(got, ref) <= SAME_PATH_TOL, max_abs(got, ref)
        assert max_abs(got.cpu(), want) <= 1e-3 and snr_db(want, got.cpu()) >= 60.0
    else:       # two bf16 runs through different tile shapes: two noise realisations behind the amplifying blocks (measured 17.6 dB)
        assert snr_db(ref, got) >= 14.0, snr_db(ref, got)
    dec.close()


def test_staggered_streams_share_launches():
    """Streams at different positions, joining and leaving, in changing order, with slot reuse."""
    cfg = q3.tiny_config("split_rvq")
    dec, _ = _decoder(cfg)
    rng = np.random.default_rng(0)
    T = {0: 30, 1: 47, 2: 12, 3: 25}
    codes = {k: q3.random_codes(cfg, 1, t, seed=10 + k)[0].cuda() for k, t in T.items()}
    refs = {k: dec.forward(codes[k][None])[0, 0] for k in T}
    pool = dec.open_streams(3)
    sid, done, outs = {}, {k: 0 for k in T}, {k: [] for k in T}
    waiting = [0, 1, 2, 3]
    step = 0
    while any(done[k] < T[k] for k in T):
        while waiting and len(sid) < 3 and (step % 2 == 0 or not sid):      # stagger the openings
            k = waiting.pop(0)
            sid[k] = pool.open()
        live = [k for k in sid if done[k] < T[k]]
        n = int(rng.integers(1, 5))
        group = [k for k in live if T[k] - done[k] >= n]
        if not group:
            n, group = 1, live
        rng.shuffle(group)
        batch = torch.stack([codes[k][:, done[k]:done[k] + n] for k in group])
        for k, piece in zip(group, pool.push([sid[k] for k in group], batch, check=True)):
            outs[k].append(piece)
            done[k] += n
        for k in list(sid):
            if done[k] >= T[k]:
                pool.close_stream(sid.pop(k))
        step += 1
    for k in T:
        got = torch.cat(outs[k])
        assert got.shape == refs[k].shape
        assert max_abs(got, refs[k]) <= SAME_PATH_TOL, (k, max_abs(got, refs[k]))


def test_streaming_wide_channels_bf16_matches_bulk_bf16():
    """bf16 contexts round at the same places in both paths; the fused residual-unit kernels and the 8-wave tiles run."""
    cfg = q3.tiny_config("split_rvq", decoder_dim=1536, latent_dim=256, hidden_size=256, intermediate_size=512)
    dec, _ = _decoder(cfg, precision="bf16")
    S, T = 3, 7
    codes = q3.random_codes(cfg, S, T, seed=3).cuda()
    ref = dec.forward(codes)[:, 0]
    pool = dec.open_streams(S)
    ids = pool.open_many(S)
    parts = [pool.push(ids, codes[:, :, a:b]) for a, b in ((0, 2), (2, 3), (3, 7))]
    for s in range(S):
        got = torch.cat([p[s] for p in parts])
        assert got.shape == ref[s].shape
        err = max_abs(got, ref[s])
        assert err <= 2e-2 and snr_db(ref[s], got) >= 40.0, (s, err, snr_db(ref[s], got))


def test_stream_errors():
    cfg = q3.tiny_config("split_rvq")
    dec, _ = _decoder(cfg, max_positions=16)
    pool = dec.open_streams(2)
    a = pool.open()
    b = pool.open()
    with pytest.raises(RuntimeError):
        pool.open()
    with pytest.raises(ValueError):
        pool.push([a, a], q3.random_codes(cfg, 2, 1, seed=0))
    with pytest.raises(ValueError):
        pool.push([a], q3.random_codes(cfg, 1, 17, seed=0))           # beyond max_positions
    with pytest.raises(ValueError):
        pool.push([a], torch.zeros((1, cfg.num_quantizers + 1, 2), dtype=torch.int64))
    bad = q3.random_codes(cfg, 1, 2, seed=0)
    bad[0, 1, 1] = cfg.codebook_size
    with pytest.raises(IndexError):
        pool.push([b], bad, check=True)
    pool.close_stream(a)
    with pytest.raises(KeyError):
        pool.push([a], q3.random_codes(cfg, 1, 1, seed=0))
    # a slot id outside the pool is caught on the device
    ids = torch.tensor([7], dtype=torch.int32, device="cuda:0")
    pool.push_raw(ids, q3.random_codes(cfg, 1, 1, seed=0).cuda())
    with pytest.raises(_lib.Q3cError):
        dec._check(dec._lib.q3c_check_errors(dec._ctx, dec._stream()))


def test_flagship_bf16_streams_track_the_bulk_forward():
    """Flagship widths in bf16: the segmented-row kernels at the channel counts of the headline benchmark.  Streaming and
    bulk round the same bf16 intermediates except where fp32 reductions group differently, so they agree far more
    closely (measured 37 dB) than bf16 agrees with fp32 (15-18 dB with these random weights); bound: 30 dB.  The bulk
    forward runs with the vector-ALU attention here (the library's measurement switch), the arithmetic the streaming pushes
    use: the bulk decode's matrix-core attention rounds queries, keys and probabilities to bf16, one more bf16-sized
    difference that these random weights amplify to the 16 dB every bf16 run has against the oracle (both attention kernels
    are held to the oracle in test_gpu_configs.py)."""
    cfg = q3.flagship_config()
    dec, _ = _decoder(cfg, precision="bf16", seed=1)
    T = 30
    codes = q3.random_codes(cfg, 2, T, seed=9)
    lib = _lib.load()
    dec.set_tuning(256)
    try:
        ref = dec.forward(codes)[:, 0].float()
    finally:
        dec.set_tuning(0)
    pool = dec.open_streams(4)
    ids = pool.open_many(2)
    outs, t = [], 0
    for n in (1, 5, 24):
        outs.append(torch.stack(list(pool.push(ids, codes[:, :, t:t + n].cuda()))))
        t += n
    got = torch.cat(outs, dim=1).float()
    assert got.shape == ref.shape == (2, cfg.output_length(T))
    rel = float((got - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
    print(f"flagship bf16 streaming vs bulk: rms(diff) / rms = {rel:.4f}")
    assert rel <= 10 ** (-30 / 20), rel



def test_graph_replayed_push_equals_plain_push():
    """q3c_stream_push_graph: the push recorded once per buffer set and replayed must produce the bits of the plain launch
    sequence, step after step (state carried between replays), for two alternating code buffers and a second (S, n)."""
    cfg = q3.tiny_config("split_rvq")
    dec, _ = _decoder(cfg, precision="bf16")
    S = 5
    pa, pb = dec.open_streams(S), dec.open_streams(S)
    ia = torch.tensor(pa.open_many(S), dtype=torch.int32, device="cuda")
    ib = torch.tensor(pb.open_many(S), dtype=torch.int32, device="cuda")
    for n, steps in ((1, 12), (3, 4)):
        bufs = [torch.empty((S, cfg.num_quantizers, n), dtype=torch.int64, device="cuda") for _ in range(2)]
        out_g = torch.empty((S, n * cfg.total_upsample), dtype=torch.float32, device="cuda")
        for t in range(steps):
            codes = q3.random_codes(cfg, S, n, seed=100 * n + t).cuda()
            bufs[t % 2].copy_(codes)
            got = pa.push_raw(ia, bufs[t % 2], out_g, graph=True)
            want = pb.push_raw(ib, codes)
            assert got.data_ptr() == out_g.data_ptr()
            assert torch.equal(got, want), (n, t, max_abs(got, want))
    # a push that was never prepared is refused instead of allocating behind the caller's back
    import ctypes as C
    rc = dec._lib.q3c_stream_push(pa._pool, C.c_void_p(ia.data_ptr()), 2, C.c_void_p(bufs[0].data_ptr()), 7, C.c_void_p(out_g.data_ptr()),
                                  C.c_void_p((pa._ws.data_ptr() + 255) // 256 * 256), 1 << 20, dec._stream())
    assert rc == -3 and b"q3c_stream_workspace_bytes" in dec._lib.q3c_last_error(dec._ctx)
