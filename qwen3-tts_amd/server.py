"""REST / SSE surface of the reference server on top of the MI355X decode.

Routes, request / response schemas and headers follow the reference:
  GET  /, /health                  -> HealthResponse                            (server.py:145-152, schemas.py:32-35)
  GET  /info                       -> loaded / available models, speakers ...   (server.py:155-162)
  POST /load/{model_type}                                                       (server.py:165-171)
  POST /tts/voice_clone            -> audio/wav (string input) or JSON (list)   (server.py:176-420, 116-140)
  POST /tts/voice_clone/sse        -> text/event-stream: meta, audio..., done   (server.py:425-486)

What differs, and why it is still a drop-in for the clients: the reference's SSE route decodes the whole utterance and
sends ONE `audio` event with `chunk_index: 0` (server.py:463-473).  Here every group of `chunk_frames` code frames is
pushed through a streaming slot (`StreamPool`: causal state carried on the GPU, so the pieces concatenate to exactly the
un-chunked decode) and leaves as its own `audio` event with `chunk_index` k; `done` carries `total_chunks`.  The
reference's web client already appends any number of `audio` events (web/index.html:609-617).  Whole-utterance routes
hand their codes to a `DecodeBatcher`, so the sentences of one request and concurrent requests share launch sequences
(the reference decodes them one by one, server.py:343-398).

The text -> codes side (talker LM, speaker prompt) is NOT part of this package: the engine takes a `code_source`
callable.  With the `qwen-tts` package that is the model's own generation with this decoder installed
(`install.install_decoder`); the tests use a deterministic stand-in.  The handlers below are plain functions on an
`Engine`; `create_app` only binds them to FastAPI routes."""
from __future__ import annotations

import json
import re
import threading
import time
from typing import Any, Callable, Dict, Iterator, List, Optional, Sequence, Tuple, Union

import numpy as np
import torch
from pydantic import BaseModel, Field

from . import audio
from .batcher import DecodeBatcher, StreamBatcher

# generation defaults and tables of the reference (config.py:75-107)
DEFAULT_MAX_NEW_TOKENS = 2048
DEFAULT_TEMPERATURE = 0.9
DEFAULT_TOP_K = 50
DEFAULT_TOP_P = 1.0
DEFAULT_REPETITION_PENALTY = 1.05
AVAILABLE_SPEAKERS = ["Vivian", "Serena", "Uncle_Fu", "Dylan", "Eric", "Ryan", "Aiden", "Ono_Anna", "Sohee"]
SUPPORTED_LANGUAGES = ["Auto", "Chinese", "English", "Japanese", "Korean", "German", "French", "Russian", "Portuguese",
                       "Spanish", "Italian"]
MODEL_KEYS = ["custom_voice_1.7b", "voice_design_1.7b", "base_1.7b", "custom_voice_0.6b", "base_0.6b", "custom_voice",
              "voice_design", "base"]


# ---- schemas (reference schemas.py:10-35: same fields, defaults and bounds) -------------------------------------------
class GenerationParams(BaseModel):
    max_new_tokens: int = Field(default=DEFAULT_MAX_NEW_TOKENS, ge=1, le=4096)
    temperature: float = Field(default=DEFAULT_TEMPERATURE, ge=0.0, le=2.0)
    top_k: int = Field(default=DEFAULT_TOP_K, ge=1, le=100)
    top_p: float = Field(default=DEFAULT_TOP_P, ge=0.0, le=1.0)
    repetition_penalty: float = Field(default=DEFAULT_REPETITION_PENALTY, ge=1.0, le=2.0)
    do_sample: bool = True


class VoiceCloneRequest(BaseModel):
    text: Union[str, List[str]] = Field(..., description="Text to synthesize")
    language: Union[str, List[str]] = Field(default="Auto", description="Language")
    ref_audio: Union[str, List[str]] = Field(..., description="Reference audio path or URL")
    ref_text: Union[str, List[str]] = Field(..., description="Reference audio transcript")
    x_vector_only_mode: bool = Field(default=True)
    split_sentences: Optional[bool] = Field(default=None)
    seed: Optional[int] = Field(default=None)
    generation_params: Optional[GenerationParams] = None


class HealthResponse(BaseModel):
    status: str
    models_loaded: List[str]


# ---- request shaping (reference server.py:74-105) ---------------------------------------------------------------------
def split_into_sentences(text: str) -> List[str]:
    """Split at runs of . ! ? followed by white space; never returns an empty list (server.py:74-91)."""
    parts = [s.strip() for s in re.split(r"[.!?]+[\s]+", text) if s.strip()]
    return parts if parts else [text.strip()]


def get_generation_kwargs(params: Optional[GenerationParams] = None) -> Dict[str, Any]:
    p = params or GenerationParams()
    return {"max_new_tokens": p.max_new_tokens, "temperature": p.temperature, "top_k": p.top_k, "top_p": p.top_p,
            "repetition_penalty": p.repetition_penalty, "do_sample": p.do_sample}


def should_split(text: str, split_sentences: Optional[bool]) -> bool:
    """None = auto: split only very long text (> 500 characters) or many sentences (> 5) (server.py:205-211)."""
    if split_sentences is not None:
        return bool(split_sentences)
    return len(text) > 500 or len(split_into_sentences(text)) > 5


def sse_event(name: str, data: Dict[str, Any], ascii_only: bool = False) -> str:
    return f"event: {name}\ndata: {json.dumps(data, ensure_ascii=ascii_only)}\n\n"


def _first(v: Union[str, Sequence[str]]) -> str:
    return v if isinstance(v, str) else v[0]


# ---- the engine: codes in, audio out ----------------------------------------------------------------------------------
CodeSource = Callable[..., torch.Tensor]
"""code_source(text=..., language=..., ref_audio=..., ref_text=..., x_vector_only_mode=..., seed=..., **generation kwargs)
-> int64 codes [num_quantizers, T]: the talker LM's output for one text."""


class _LockedPcm16Decoder:
    """What DecodeBatcher needs from a decoder, with 16-bit PCM results (converted on the GPU) and the engine's lock held
    during the decode: a q3c context must not be used from two threads at once, and the batcher's worker thread shares
    it with the SSE handlers' streaming pushes."""

    def __init__(self, decoder, lock: threading.Lock):
        self._dec, self._lock, self.cfg = decoder, lock, decoder.cfg

    def _normalize_items(self, encoded):
        return self._dec._normalize_items(encoded)

    def decode(self, encoded):
        with self._lock:
            return audio.decode_pcm16(self._dec, encoded)


class StreamSlotsBusy(RuntimeError):
    """Every streaming slot stayed busy for `slot_wait_s`: the SSE route answers 503."""


class Engine:
    """One decoder on one GPU behind the server's handlers.

    decoder:      qwen3_tts_amd.decoder.Decoder
    code_source:  text -> codes (see CodeSource); called once per sentence / list item, in the handler's thread
    stream_slots: concurrent SSE requests (each holds one streaming slot while it runs)
    chunk_frames: code frames per SSE `audio` event (25 frames = 2 s of audio)"""

    def __init__(self, decoder, code_source: CodeSource, model_key: str = "base_0.6b", stream_slots: int = 64, chunk_frames: int = 25,
                 batch_wait_ms: float = 2.0, slot_wait_s: float = 30.0):
        self.decoder = decoder
        self.code_source = code_source
        self.models: List[str] = [model_key]
        self.chunk_frames = int(chunk_frames)
        self.slot_wait_s = float(slot_wait_s)
        self._pool_lock = threading.Lock()         # one context, one pool: every use of the decoder holds this lock
        self.batcher = DecodeBatcher(_LockedPcm16Decoder(decoder, self._pool_lock), max_wait_ms=batch_wait_ms)
        self.pool = decoder.open_streams(stream_slots)
        # concurrent SSE requests share pushes: one launch sequence per tick for every stream with a chunk pending
        self.streams = StreamBatcher(self.pool, self._pool_lock, max_wait_ms=batch_wait_ms, to_pcm16=audio.to_pcm16)
        self._slots = threading.BoundedSemaphore(int(stream_slots))

    def close(self) -> None:
        self.batcher.close()
        self.streams.close()
        self.pool.close()

    # model manager surface (reference models.py:108-128)
    def get_loaded_models(self) -> List[str]:
        return list(self.models)

    def load_model(self, model_type: str) -> None:
        if model_type not in MODEL_KEYS:
            raise ValueError(f"Unknown model type: {model_type}. Available: {MODEL_KEYS}")
        if model_type not in self.models:
            self.models.append(model_type)

    def get_model(self, model_key: str) -> "Engine":
        self.load_model(model_key)
        return self

    # whole utterances: one batched decode for all texts of the call -> (int16 waveforms, sample rate)
    def synthesize(self, texts: Sequence[str], seeds: Sequence[Optional[int]], **kw) -> Tuple[List[np.ndarray], int]:
        codes = [self.code_source(text=t, seed=s, **kw) for t, s in zip(texts, seeds)]
        return self.batcher.decode(codes)

    def acquire_stream(self) -> int:
        """Claim a streaming slot for one SSE request BEFORE its response starts: when all `stream_slots` are busy the request
        waits up to `slot_wait_s` (the reference serves such requests one after the other) and then fails with
        StreamSlotsBusy, which the route answers as HTTP 503 instead of a truncated event stream."""
        if not self._slots.acquire(timeout=self.slot_wait_s):
            raise StreamSlotsBusy(f"all {self.pool.capacity} streaming slots are busy")
        try:
            return self.streams.open()
        except Exception:
            self._slots.release()
            raise

    def release_stream(self, sid: int) -> None:
        try:
            self.streams.close_stream(sid)
        finally:
            self._slots.release()

    # one utterance, chunk by chunk: yields int16 arrays whose concatenation is the whole utterance
    def synthesize_chunks(self, text: str, seed: Optional[int], sid: Optional[int] = None, **kw) -> Iterator[np.ndarray]:
        """`sid`: a slot from acquire_stream, which the CALLER releases; None claims one here and releases it at the end."""
        own = sid is None
        if own:
            sid = self.acquire_stream()
        try:
            codes = torch.as_tensor(self.code_source(text=text, seed=seed, **kw))
            Q = self.decoder.cfg.num_quantizers
            if codes.dim() != 2:
                raise ValueError(f"code source returned shape {tuple(codes.shape)}, expected [{Q}, T]")
            if codes.shape[0] != Q:
                codes = codes.t()
            T = int(codes.shape[1])
            if T < 1:
                raise ValueError("No audio generated")
            if T > self.decoder.max_positions:
                # longer than a stream may run: the whole utterance through the chunked decode, as one piece (the
                # reference's own stream is one `audio` event, server.py:463-473)
                pcms, _ = self.batcher.decode([codes])
                yield pcms[0]
                return
            for a in range(0, T, self.chunk_frames):
                pcm = self.streams.advance(sid, codes[:, a:a + self.chunk_frames])
                if pcm.size:
                    yield pcm
        finally:
            if own:
                self.release_stream(sid)


# ---- handlers (framework-free: create_app binds them) -------------------------------------------------------------------
def health(engine: Engine) -> HealthResponse:
    return HealthResponse(status="ok", models_loaded=engine.get_loaded_models())


def info(engine: Engine) -> Dict[str, Any]:
    return {"loaded_models": engine.get_loaded_models(), "available_models": list(MODEL_KEYS),
            "available_speakers": AVAILABLE_SPEAKERS, "supported_languages": SUPPORTED_LANGUAGES}


def model_key_for(model_size: str) -> str:
    return f"base_{model_size}" if model_size in ("0.6b", "1.7b") else "base"


def voice_clone(engine: Engine, request: VoiceCloneRequest, model_size: str = "0.6b") -> Dict[str, Any]:
    """-> {"single": True, "wav": bytes, "generation_time": s} for a string input (the reference answers audio/wav),
    {"single": False, "json": {...}} for a list input (base64 WAVs in JSON, server.py:131-140)."""
    engine.get_model(model_key_for(model_size))
    gen_kwargs = get_generation_kwargs(request.generation_params)
    t0 = time.time()
    if isinstance(request.text, str):
        text = request.text
        common = dict(language=request.language, ref_audio=request.ref_audio, ref_text=request.ref_text,
                      x_vector_only_mode=True, **gen_kwargs)           # x-vector mode forced, as the reference does (server.py:221,283)
        if not should_split(text, request.split_sentences):
            wavs, sr = engine.synthesize([text], [request.seed], **common)
        else:
            sentences = split_into_sentences(text)
            seed0 = request.seed if request.seed is not None else int(time.time() * 1000) % (2 ** 31)
            # the reference seeds sentence i with seed + i and decodes them one by one (server.py:345-379); here all
            # sentences of the request go through ONE batched decode
            parts, sr = engine.synthesize(sentences, [seed0 + i for i in range(len(sentences))], **common)
            if not parts:
                raise ValueError("No audio generated")
            wavs = [np.concatenate(parts)] if len(parts) > 1 else parts
        gen_time = time.time() - t0
        return {"single": True, "wav": audio.wav_file_bytes(wavs[0], sr), "sample_rate": sr, "generation_time": gen_time}
    texts = list(request.text)
    n = len(texts)

    def per_item(v):
        return [v] * n if isinstance(v, str) else list(v)

    langs, refs, reft = per_item(request.language), per_item(request.ref_audio), per_item(request.ref_text)
    codes = [engine.code_source(text=texts[i], seed=request.seed, language=langs[i], ref_audio=refs[i], ref_text=reft[i],
                                x_vector_only_mode=request.x_vector_only_mode, **gen_kwargs) for i in range(n)]
    pcms, sr = engine.batcher.decode(codes) if n else ([], engine.decoder.cfg.sample_rate)
    gen_time = time.time() - t0
    return {"single": False, "json": {"success": True, "message": f"Generated {len(pcms)} audio(s)", "sample_rate": sr,
                                      "audio_count": len(pcms), "audio_data": [audio.audio_to_base64(p, sr) for p in pcms],
                                      "generation_time": gen_time}}


def voice_clone_sse(engine: Engine, request: VoiceCloneRequest, model_size: str = "0.6b", streaming: bool = True) -> Iterator[str]:
    """The event stream of /tts/voice_clone/sse: `meta`, then one `audio` event per decoded chunk (`chunk_index` k, a
    base64 PCM_16 WAV file, `sample_rate`, `generation_time` so far), then `done` with `total_chunks`.  With
    streaming=False the whole utterance is one chunk, which is exactly the reference's stream."""
    engine.get_model(model_key_for(model_size))
    gen_kwargs = get_generation_kwargs(request.generation_params)
    text = _first(request.text)
    kw = dict(language=_first(request.language), ref_audio=_first(request.ref_audio), ref_text=_first(request.ref_text),
              x_vector_only_mode=request.x_vector_only_mode, **gen_kwargs)
    sr = engine.decoder.cfg.sample_rate
    # the slot is claimed here, before the response starts: a full pool is an HTTP error, not a truncated event stream
    sid = engine.acquire_stream() if streaming else None

    def events() -> Iterator[str]:
        t0 = time.time()
        yield sse_event("meta", {"status": "generating", "text": text})
        k = 0
        if streaming:
            for pcm in engine.synthesize_chunks(text, request.seed, sid=sid, **kw):
                yield sse_event("audio", {"chunk_index": k, "audio": audio.audio_to_base64(pcm, sr), "sample_rate": sr,
                                          "generation_time": round(time.time() - t0, 3)})
                k += 1
        else:
            pcms, _ = engine.batcher.decode([engine.code_source(text=text, seed=request.seed, **kw)])
            yield sse_event("audio", {"chunk_index": 0, "audio": audio.audio_to_base64(pcms[0], sr), "sample_rate": sr,
                                      "generation_time": round(time.time() - t0, 3)})
            k = 1
        yield sse_event("done", {"total_time": round(time.time() - t0, 3), "total_chunks": k}, ascii_only=True)

    return _SlotGuard(events(), engine, sid) if sid is not None else events()


class _SlotGuard:
    """The event stream of a request that holds a streaming slot: the slot goes back exactly once - when the stream ends,
    fails, is closed by the server (client gone), or is dropped without ever having been started."""

    def __init__(self, it: Iterator[str], engine: "Engine", sid: int):
        self._it, self._engine, self._sid = it, engine, sid

    def __iter__(self):
        return self

    def __next__(self) -> str:
        try:
            return next(self._it)
        except BaseException:
            self.close()
            raise

    def close(self) -> None:
        sid, self._sid = self._sid, None
        if sid is not None:
            try:
                self._it.close()
            finally:
                self._engine.release_stream(sid)

    __del__ = close


# ---- FastAPI binding -----------------------------------------------------------------------------------------------------
def create_app(engine: Engine):
    """The reference's application object (server.py:56-69) with the routes above."""
    from fastapi import FastAPI, HTTPException
    from fastapi.middleware.cors import CORSMiddleware
    from fastapi.responses import JSONResponse, Response, StreamingResponse

    app = FastAPI(title="Qwen3-TTS API Server", description="Text-to-Speech API using Qwen3-TTS models", version="1.0.0")
    app.add_middleware(CORSMiddleware, allow_origins=["*"], allow_credentials=True, allow_methods=["*"], allow_headers=["*"])

    @app.get("/", response_model=HealthResponse)
    def root():
        return health(engine)

    @app.get("/health", response_model=HealthResponse)
    def health_check():
        return health(engine)

    @app.get("/info")
    def get_info():
        return info(engine)

    @app.post("/load/{model_type}")
    def load_model(model_type: str):
        try:
            engine.load_model(model_type)
            return {"success": True, "message": f"Model {model_type} loaded successfully"}
        except Exception as e:
            raise HTTPException(status_code=400, detail=str(e))

    # plain `def` handlers: FastAPI runs them on its thread pool, so concurrent requests reach the batcher together
    # (the reference's `async def` handlers block the event loop and serve one request at a time, server.py:177)
    @app.post("/tts/voice_clone")
    def generate_voice_clone(request: VoiceCloneRequest, model_size: str = "0.6b"):
        try:
            res = voice_clone(engine, request, model_size)
        except Exception as e:
            raise HTTPException(status_code=500, detail=str(e))
        if res["single"]:
            return Response(content=res["wav"], media_type="audio/wav",
                            headers={"Content-Disposition": "attachment; filename=output.wav",
                                     "X-Generation-Time": f"{res['generation_time']:.3f}",
                                     "Access-Control-Expose-Headers": "X-Generation-Time"})
        return JSONResponse(res["json"])

    @app.post("/tts/voice_clone/sse")
    def tts_voice_clone_sse(request: VoiceCloneRequest, model_size: str = "0.6b", streaming: bool = True):
        try:
            stream = voice_clone_sse(engine, request, model_size, streaming)
        except StreamSlotsBusy as e:
            raise HTTPException(status_code=503, detail=str(e))
        except Exception as e:
            raise HTTPException(status_code=500, detail=str(e))
        return StreamingResponse(stream, media_type="text/event-stream",
                                 headers={"Cache-Control": "no-cache", "Connection": "keep-alive",
                                          "Access-Control-Expose-Headers": "X-Generation-Time"})

    return app


def hashed_code_source(cfg, frames_per_char: float = 0.9) -> CodeSource:
    """A deterministic stand-in for the talker LM (tests, load tests): codes drawn from a generator seeded by the text and
    the seed, about 12.5 frames per 14 characters."""
    import hashlib

    def source(text: str, seed: Optional[int] = None, **_kw) -> torch.Tensor:
        h = int.from_bytes(hashlib.sha256(f"{seed}|{text}".encode("utf-8")).digest()[:8], "little") % (2 ** 62)
        g = torch.Generator().manual_seed(h)
        T = max(2, int(round(len(text) * frames_per_char)))
        return torch.randint(0, cfg.codebook_size, (cfg.num_quantizers, T), generator=g, dtype=torch.int64)

    return source


def build_engine(manager=None, env: Optional[Dict[str, str]] = None) -> Engine:
    """The engine of a real deployment: the model manager loads the default model from its LOCAL directory (models.py), installs
    this decode into it, and the model's own generation becomes the code source (install.ModelCodeSource).  Environment, as the
    reference's config.py reads it: TTS_DEVICE, TTS_DTYPE, TTS_DEFAULT_MODEL, TTS_USE_WARMUP, TTS_MODELS_DIR / MODEL_* paths;
    plus TTS_STREAM_SLOTS (concurrent SSE requests, default 64) and TTS_CHUNK_FRAMES (frames per SSE audio event, default 25)."""
    import os
    from .install import ModelCodeSource
    from .models import TTSModelManager
    env = os.environ if env is None else env
    slots, chunk = int(env.get("TTS_STREAM_SLOTS", "64")), int(env.get("TTS_CHUNK_FRAMES", "25"))
    if manager is None:
        manager = TTSModelManager(env=env)
    key = manager.default_model if manager.default_model != "all" else "base"      # the voice-clone routes serve a base model
    model = manager.get_model(key)
    decoder = manager.get_decoder(key)
    return Engine(decoder, ModelCodeSource(model, decoder.cfg), model_key=key, stream_slots=slots, chunk_frames=chunk)


def main() -> None:
    """`python -m qwen3_tts_amd.server`: what `python server.py` is for the reference (server.py:575-577; TTS_HOST / TTS_PORT as in
    config.py:8-9)."""
    import os
    import uvicorn
    uvicorn.run(create_app(build_engine()), host=os.getenv("TTS_HOST", "0.0.0.0"), port=int(os.getenv("TTS_PORT", "8000")))


if __name__ == "__main__":
    main()
