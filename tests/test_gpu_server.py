"""The REST / SSE surface (qwen3-tts_amd/server.py) on the GPU: the reference's routes, schemas and headers
(reference server.py:145-171, 176-277, 425-486; schemas.py:10-35), with one `audio` event per decoded chunk.  The talker LM
is a deterministic stand-in (`hashed_code_source`), so every response can be recomputed: whole utterances against the CPU
oracle's decode converted by the PCM_16 oracle, streamed chunks against the same un-chunked decode."""
import base64
import io
import json
import struct

import numpy as np
import pytest
import torch

import qwen3_tts_amd as q3
from qwen3_tts_amd import server
from qwen3_tts_amd.decoder import Decoder

import pcm16_oracle as po
from gpu_util import oracle

pytestmark = pytest.mark.gpu

REF = {"ref_audio": "sample(1).mp3", "ref_text": "reference transcript"}


@pytest.fixture(scope="module")
def stack():
    from fastapi.testclient import TestClient
    cfg = q3.tiny_config()
    w = q3.synthetic_weights(cfg, seed=11)
    dec = Decoder(cfg, w, device="cuda:0", precision="f32")
    src = server.hashed_code_source(cfg)
    eng = server.Engine(dec, src, stream_slots=4, chunk_frames=25)
    client = TestClient(server.create_app(eng))
    yield cfg, w, src, eng, client
    eng.close()


def _pcm_of_wav_file(data: bytes, sr: int = 24000) -> np.ndarray:
    assert data[:4] == b"RIFF" and data[8:12] == b"WAVE" and data[12:16] == b"fmt "
    fmt, ch, rate, _br, _ba, bits = struct.unpack("<HHIIHH", data[20:36])
    assert (fmt, ch, rate, bits) == (1, 1, sr, 16) and data[36:40] == b"data"
    n = struct.unpack("<I", data[40:44])[0]
    assert n == len(data) - 44
    return np.frombuffer(data[44:], dtype="<i2")


def _want_pcm(cfg, w, codes_list, chunked=True):
    """The oracle's decode of each [Q, T] code matrix, as PCM_16."""
    out = []
    for c in codes_list:
        wav = (oracle.chunked_decode(cfg, w, c[None]) if chunked else oracle.forward(cfg, w, c[None]))[0, 0].numpy()
        out.append(po.float_to_pcm16(wav))
    return out


def _close(a, b):
    """The fp32 decode's bar against the oracle is 1e-3 (33 steps of 16-bit PCM; measured ~5e-5): here at most 4 steps
    apart, more than one step for under 1 % of the samples, equal for most."""
    assert a.shape == b.shape
    d = np.abs(a.astype(np.int32) - b.astype(np.int32))
    assert d.max() <= 4 and (d > 1).mean() < 0.01 and (d != 0).mean() < 0.3, (d.max(), (d > 1).mean(), (d != 0).mean())


def test_health_info_load(stack):
    _cfg, _w, _src, _eng, client = stack
    for path in ("/", "/health"):
        r = client.get(path)
        assert r.status_code == 200 and r.json() == {"status": "ok", "models_loaded": ["base_0.6b"]}
    i = client.get("/info").json()
    assert set(i) == {"loaded_models", "available_models", "available_speakers", "supported_languages"}
    assert "base_0.6b" in i["available_models"] and "Sohee" in i["available_speakers"] and i["supported_languages"][0] == "Auto"
    assert client.post("/load/base_1.7b").json() == {"success": True, "message": "Model base_1.7b loaded successfully"}
    assert "base_1.7b" in client.get("/health").json()["models_loaded"]
    r = client.post("/load/nonsense")
    assert r.status_code == 400 and "Unknown model type" in r.json()["detail"]


def test_voice_clone_single_block_is_a_wav_file(stack):
    cfg, w, src, _eng, client = stack
    text = "Hello there. This is a test"
    r = client.post("/tts/voice_clone", json={"text": text, "seed": 7, **REF})
    assert r.status_code == 200 and r.headers["content-type"] == "audio/wav"
    assert r.headers["content-disposition"] == "attachment; filename=output.wav"
    assert float(r.headers["x-generation-time"]) >= 0 and r.headers["access-control-expose-headers"] == "X-Generation-Time"
    got = _pcm_of_wav_file(r.content)
    _close(got, _want_pcm(cfg, w, [src(text=text, seed=7)])[0])
    from qwen3_tts_amd import audio
    direct, _ = audio.decode_pcm16(_eng.decoder, [src(text=text, seed=7)])       # and exactly the library's own decode
    assert np.array_equal(got, direct[0])


def test_voice_clone_splits_long_text_and_batches_the_sentences(stack):
    cfg, w, src, eng, client = stack
    sentences = [f"Sentence number {i} goes here" for i in range(7)]
    text = ". ".join(sentences) + "."
    before = eng.batcher.batches
    r = client.post("/tts/voice_clone", json={"text": text, "seed": 100, **REF})
    assert r.status_code == 200
    assert eng.batcher.batches == before + 1                      # seven sentences, one batched decode
    parts = server.split_into_sentences(text)
    assert len(parts) == 7
    want = np.concatenate(_want_pcm(cfg, w, [src(text=s, seed=100 + i) for i, s in enumerate(parts)]))
    _close(_pcm_of_wav_file(r.content), want)
    r2 = client.post("/tts/voice_clone", json={"text": text, "seed": 100, "split_sentences": False, **REF})
    _close(_pcm_of_wav_file(r2.content), _want_pcm(cfg, w, [src(text=text, seed=100)])[0])


def test_voice_clone_list_input_answers_json(stack):
    cfg, w, src, _eng, client = stack
    texts = ["first item", "the second item is longer"]
    r = client.post("/tts/voice_clone", json={"text": texts, "seed": 3, **REF})
    assert r.status_code == 200
    j = r.json()
    assert j["success"] is True and j["audio_count"] == 2 and j["sample_rate"] == 24000 and j["message"] == "Generated 2 audio(s)"
    assert j["generation_time"] >= 0
    want = _want_pcm(cfg, w, [src(text=t, seed=3) for t in texts])
    for b64, wnt in zip(j["audio_data"], want):
        _close(_pcm_of_wav_file(base64.b64decode(b64)), wnt)


def _events(body: str):
    out = []
    for block in body.strip().split("\n\n"):
        lines = block.split("\n")
        assert lines[0].startswith("event: ") and lines[1].startswith("data: ")
        out.append((lines[0][7:], json.loads(lines[1][6:])))
    return out


def test_sse_sends_one_audio_event_per_chunk(stack):
    cfg, w, src, eng, client = stack
    text = "x" * 80                                              # 72 frames -> chunks of 25, 25, 22
    r = client.post("/tts/voice_clone/sse", json={"text": text, "seed": 5, **REF})
    assert r.status_code == 200 and r.headers["content-type"].startswith("text/event-stream")
    assert r.headers["cache-control"] == "no-cache"
    ev = _events(r.text)
    assert ev[0] == ("meta", {"status": "generating", "text": text})
    audio_ev = [d for n, d in ev if n == "audio"]
    T = src(text=text, seed=5).shape[1]
    assert len(audio_ev) == -(-T // eng.chunk_frames) == 3
    assert [d["chunk_index"] for d in audio_ev] == [0, 1, 2] and all(d["sample_rate"] == 24000 for d in audio_ev)
    assert ev[-1][0] == "done" and ev[-1][1]["total_chunks"] == 3 and ev[-1][1]["total_time"] >= audio_ev[-1]["generation_time"] - 1e-3
    got = np.concatenate([_pcm_of_wav_file(base64.b64decode(d["audio"])) for d in audio_ev])
    # the pieces concatenate to the UN-chunked decode of all frames: no seams, no recomputed context
    _close(got, _want_pcm(cfg, w, [src(text=text, seed=5)], chunked=False)[0])
    assert len(got) == T * cfg.total_upsample - cfg.trim
    # the slot went back to the pool
    assert len(eng.pool._frames) == 0


def test_sse_without_streaming_is_the_reference_stream(stack):
    cfg, w, src, _eng, client = stack
    r = client.post("/tts/voice_clone/sse?streaming=false", json={"text": ["only the first text is used", "ignored"], "seed": 1, **REF})
    ev = _events(r.text)
    assert [n for n, _ in ev] == ["meta", "audio", "done"] and ev[1][1]["chunk_index"] == 0 and ev[2][1]["total_chunks"] == 1
    _close(_pcm_of_wav_file(base64.b64decode(ev[1][1]["audio"])), _want_pcm(cfg, w, [src(text="only the first text is used", seed=1)])[0])


def test_errors_become_http_500_with_the_message(stack):
    _cfg, _w, _src, eng, client = stack
    good = eng.code_source

    def broken(**kw):
        raise RuntimeError("talker exploded")
    eng.code_source = broken
    try:
        r = client.post("/tts/voice_clone", json={"text": "hi", **REF})
        assert r.status_code == 500 and r.json()["detail"] == "talker exploded"
    finally:
        eng.code_source = good
    r = client.post("/tts/voice_clone", json={"text": "hi"})      # schema: ref_audio / ref_text are required
    assert r.status_code == 422
    r = client.post("/tts/voice_clone", json={"text": "hi", "generation_params": {"temperature": 3.0}, **REF})
    assert r.status_code == 422


def test_concurrent_sse_requests_share_pushes_and_each_gets_its_own_audio():
    """24 SSE requests at once (reference route: server.py:425-486, which serves them one by one): their chunks advance through
    SHARED pushes - one launch sequence per tick for every stream with a chunk pending - so the number of pushes is far below
    requests x chunks, and every request's concatenated audio is still the oracle's un-chunked decode of its own codes."""
    import threading
    cfg = q3.tiny_config()
    w = q3.synthetic_weights(cfg, seed=11)
    dec = Decoder(cfg, w, device="cuda:0", precision="f32")
    src = server.hashed_code_source(cfg)
    eng = server.Engine(dec, src, stream_slots=32, chunk_frames=10, batch_wait_ms=20.0)
    try:
        N = 24
        texts = [("request %02d " % i) + "y" * (20 + 3 * i) for i in range(N)]        # 27 .. 90 frames: 3 .. 9 chunks each
        results, errors = [None] * N, []
        gate = threading.Barrier(N)

        def run(i):
            try:
                gate.wait()
                req = server.VoiceCloneRequest(text=texts[i], seed=i, **REF)
                results[i] = _events("".join(server.voice_clone_sse(eng, req)))
            except Exception as e:                                      # noqa: BLE001
                errors.append((i, repr(e)))

        threads = [threading.Thread(target=run, args=(i,)) for i in range(N)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(120)
        assert not errors, errors
        total_chunks = 0
        for i in range(N):
            ev = results[i]
            codes = src(text=texts[i], seed=i)
            n_chunks = -(-codes.shape[1] // eng.chunk_frames)
            total_chunks += n_chunks
            audio_ev = [d for n, d in ev if n == "audio"]
            assert ev[0][0] == "meta" and ev[-1][0] == "done" and ev[-1][1]["total_chunks"] == n_chunks == len(audio_ev)
            assert [d["chunk_index"] for d in audio_ev] == list(range(n_chunks))
            got = np.concatenate([_pcm_of_wav_file(base64.b64decode(d["audio"])) for d in audio_ev])
            _close(got, _want_pcm(cfg, w, [codes], chunked=False)[0])
        print(f"{N} concurrent SSE requests, {total_chunks} chunks: {eng.streams.pushes} pushes in {eng.streams.ticks} ticks")
        assert eng.streams.chunks == total_chunks
        assert eng.streams.pushes <= total_chunks // 4, (eng.streams.pushes, total_chunks)     # measured: ~1/8
        assert len(eng.pool._frames) == 0                                # every slot went back
    finally:
        eng.close()


def test_sse_full_pool_is_http_503_and_bad_chunks_stay_with_their_request():
    from fastapi.testclient import TestClient
    cfg = q3.tiny_config()
    dec = Decoder(cfg, q3.synthetic_weights(cfg, seed=11), device="cuda:0", precision="f32", max_positions=320)
    src = server.hashed_code_source(cfg)
    eng = server.Engine(dec, src, stream_slots=1, chunk_frames=25, slot_wait_s=0.05)
    try:
        client = TestClient(server.create_app(eng))
        held = eng.acquire_stream()
        r = client.post("/tts/voice_clone/sse", json={"text": "hello", "seed": 1, **REF})
        assert r.status_code == 503 and "busy" in r.json()["detail"]
        eng.release_stream(held)
        r = client.post("/tts/voice_clone/sse", json={"text": "hello again", "seed": 1, **REF})
        assert r.status_code == 200 and _events(r.text)[-1][0] == "done"
        # an utterance longer than a stream may run (max_positions = 320 frames) falls back to one chunked whole-utterance piece
        long_text = "z" * 400                                            # 360 frames
        ev = _events(client.post("/tts/voice_clone/sse", json={"text": long_text, "seed": 2, **REF}).text)
        assert [n for n, _ in ev] == ["meta", "audio", "done"] and ev[-1][1]["total_chunks"] == 1
        assert len(eng.pool._frames) == 0
    finally:
        eng.close()
    # a chunk with a code outside the code book is refused in its caller's thread; the stream beside it is untouched
    eng = server.Engine(dec, src, stream_slots=2, chunk_frames=25)
    try:
        a, b = eng.streams.open(), eng.streams.open()
        good = q3.random_codes(cfg, 1, 5, seed=3)[0]
        bad = good.clone()
        bad[1, 2] = cfg.codebook_size
        with pytest.raises(IndexError):
            eng.streams.submit(a, bad)
        with pytest.raises(ValueError):
            eng.streams.submit(a, torch.zeros((cfg.num_quantizers + 1, 3), dtype=torch.int64))
        fa, fb = eng.streams.submit(a, good), eng.streams.submit(b, good)
        assert np.array_equal(fa.result(30), fb.result(30)) and fa.result().dtype == np.int16
        assert fa.result().shape == (5 * cfg.total_upsample - cfg.trim,)
        eng.streams.close_stream(a)
        eng.streams.close_stream(b)
    finally:
        eng.close()


def test_engine_built_from_the_model_manager_serves_the_models_own_codes(tmp_path):
    """`server.build_engine`: the model manager loads a (stub) Qwen3TTSModel from a local directory, installs the real decoder
    into it, and the model's own generation - codes handed to the tokenizer's decode - is the server's code source.  The
    voice-clone route then answers with the oracle's decode of exactly those codes, and the model called directly (the
    reference's own server path) returns the same samples through the installed decode."""
    from fastapi.testclient import TestClient
    from qwen3_tts_amd import models
    from test_host_cpu import _StubDecoderModule, _StubTokenizer, _StubTTS
    cfg = q3.tiny_config()
    w = q3.synthetic_weights(cfg, seed=13)
    src = server.hashed_code_source(cfg)

    class Model(_StubTTS):
        def __init__(self):
            super().__init__(_StubTokenizer(_StubDecoderModule(cfg, {k: v.cuda() for k, v in w.items()}, "")))

        def generate_voice_clone(self, text, **kw):
            codes = src(text=text, seed=torch.initial_seed())                    # the "talker LM": seeded by the request
            return self.model.speech_tokenizer.decode([{"audio_codes": codes.t().contiguous()}])       # [T, Q] items, as upstream

    root = tmp_path / "models" / "Qwen3-TTS-12Hz-0.6B-Base"
    root.mkdir(parents=True)
    env = {"TTS_DEVICE": "cuda:0", "TTS_DTYPE": "float32", "TTS_STREAM_SLOTS": "2", "TTS_CHUNK_FRAMES": "10"}
    mm = models.TTSModelManager(factory=lambda path, **kw: Model(), env=env, models_root=str(tmp_path / "models"))
    eng = server.build_engine(mm, env)
    try:
        assert eng.pool.capacity == 2 and eng.chunk_frames == 10 and eng.get_loaded_models() == ["base_0.6b"]
        client = TestClient(server.create_app(eng))
        text = "A sentence for the stub model"
        r = client.post("/tts/voice_clone", json={"text": text, "seed": 21, **REF})
        assert r.status_code == 200, r.text
        _close(_pcm_of_wav_file(r.content), _want_pcm(cfg, w, [src(text=text, seed=21)])[0])
        # the reference's own path: the model called directly decodes through the installed MI355X decode
        torch.manual_seed(21)
        wavs, sr = mm.get_model("base_0.6b").generate_voice_clone(text=text)
        assert sr == 24000
        want = oracle.chunked_decode(cfg, w, src(text=text, seed=21)[None])[0, 0].numpy()
        assert np.abs(wavs[0] - want).max() <= 1e-3
    finally:
        eng.close()
