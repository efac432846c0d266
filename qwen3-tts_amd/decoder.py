"""Host side of the MI355X decode: `Decoder.decode(codes) -> (wavs, sr)`.

Mirrors what the reference reaches through `Qwen3TTSModel.generate_voice_clone(...) -> (wavs, sr)`
(reference server.py:245-251; `wavs[i]` is a 1-D float32 numpy array, `sr` 24000: server.py:379-389,
API_GUIDE.md:152) and the tensor-level `forward` / `chunked_decode` of the sibling Code2Wav
(transformers/models/qwen3_omni_moe/modeling_qwen3_omni_moe.py:3672-3696).  All arithmetic runs in
libq3c.so (hand-written gfx950 kernels); torch is used for device memory and streams only.
"""
from __future__ import annotations

import ctypes as C
import os
import threading
from typing import Dict, List, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from . import _lib
from .config import DecoderConfig
from .weights import validate_weights

CodesLike = Union[torch.Tensor, np.ndarray, Dict[str, object], Sequence[object]]


def rope_inv_freq(cfg: DecoderConfig) -> torch.Tensor:
    """inv_freq = 1 / theta^(2i/d), computed by torch in fp32 exactly as the sibling's rotary embedding does
    (modeling_qwen3_omni_moe.py:2419)."""
    d = cfg.head_dim
    return 1.0 / (cfg.rope_theta ** (torch.arange(0, d, 2, dtype=torch.float) / d))


_STAGE: Dict[int, tuple] = {}          # device index -> (two pinned staging buffers, two events, a lock)
_STAGE_GUARD = threading.Lock()
_STAGE_PIECE = 128 << 20               # bytes per staging buffer
_COPY_THREADS = 4
_copy_pool = None


def _stage_buffers(dev: torch.device) -> tuple:
    """The device's pinned staging buffers, events and the lock that serialises their users (two threads copying out on one
    GPU - a batcher worker beside a gather, two decoders - would otherwise overwrite each other's pieces)."""
    with _STAGE_GUARD:
        if dev.index not in _STAGE:
            _STAGE[dev.index] = ([torch.empty(_STAGE_PIECE, dtype=torch.uint8, pin_memory=True) for _ in range(2)],
                                 [torch.cuda.Event() for _ in range(2)], threading.Lock())
        return _STAGE[dev.index]


def _parallel_copy(dst: np.ndarray, src: np.ndarray) -> None:
    """dst[:] = src for large byte arrays, split over a few threads: one core's memcpy out of the pinned staging buffer
    (~10 GB/s) is slower than the PCIe copy that fills it, and numpy releases the GIL while it copies."""
    global _copy_pool
    n = dst.shape[0]
    if n < (8 << 20):
        dst[:] = src
        return
    if _copy_pool is None:
        from concurrent.futures import ThreadPoolExecutor
        _copy_pool = ThreadPoolExecutor(max_workers=_COPY_THREADS, thread_name_prefix="q3tts-d2h")
    step = -(-n // _COPY_THREADS)

    def part(i):
        dst[i:i + step] = src[i:i + step]
    list(_copy_pool.map(part, range(0, n, step)))


def to_host(flat: torch.Tensor) -> np.ndarray:
    """Device samples (float32 or int16) -> one numpy array, through two pinned staging buffers per GPU: the copy of piece
    k + 1 from the GPU runs while piece k moves from its staging buffer into the result (a pageable copy of the whole
    tensor is several times slower, and pinning a result-sized buffer per call costs more than it saves)."""
    flat = flat.reshape(-1)
    n = int(flat.numel())
    res = np.empty(n, dtype=np.int16 if flat.dtype == torch.int16 else np.float32)
    if n == 0:
        return res
    if n <= (1 << 20) or not flat.is_cuda:              # small results: not worth the staging
        res[:] = flat.cpu().numpy()
        return res
    bytes_dev = flat.view(torch.uint8)
    res_b = res.view(np.uint8)
    nb = int(bytes_dev.numel())
    piece = _STAGE_PIECE
    dev = flat.device
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    stage, stage_ev, lock = _stage_buffers(dev)
    stream = torch.cuda.current_stream(dev)
    bounds = [(a, min(a + piece, nb)) for a in range(0, nb, piece)]
    with lock, torch.cuda.device(dev):
        for k, (a, b) in enumerate(bounds[:2]):
            stage[k][: b - a].copy_(bytes_dev[a:b], non_blocking=True)
            stage_ev[k].record(stream)
        for k, (a, b) in enumerate(bounds):
            j = k & 1
            stage_ev[j].synchronize()
            _parallel_copy(res_b[a:b], stage[j][: b - a].numpy())
            if k + 2 < len(bounds):
                a2, b2 = bounds[k + 2]
                stage[j][: b2 - a2].copy_(bytes_dev[a2:b2], non_blocking=True)
                stage_ev[j].record(stream)
    return res


_COPY_STREAMS: Dict[int, "torch.cuda.Stream"] = {}


def to_host_by_launch(flat: torch.Tensor, total: int, ends: Sequence[int], events: Sequence["torch.cuda.Event"],
                      p_dev: np.ndarray, p_host: np.ndarray, p_n: np.ndarray) -> np.ndarray:
    """Device samples laid out launch by launch -> one float32 numpy array in the caller's order, WHILE the later launches
    still run: launch k's samples end at element `ends[k]` of `flat` and are complete when `events[k]` has fired; piece i is
    `p_n[i]` samples at device offset `p_dev[i]` (ascending) that belong at host offset `p_host[i]`.  The copies go through
    the device's two pinned staging buffers on a stream of their own, each waiting only for the launches it covers, and
    every staged range is scattered to its pieces' places by a few threads."""
    import bisect
    res = np.empty(total, dtype=np.float32)
    n = int(flat.numel())
    if n == 0:
        return res
    p_dev_l = p_dev.tolist()

    def scatter(src: np.ndarray, a: int, b: int) -> None:
        i = max(0, bisect.bisect_right(p_dev_l, a) - 1)
        jobs = []
        while i < len(p_dev_l) and p_dev_l[i] < b:
            lo, hi = max(a, p_dev_l[i]), min(b, p_dev_l[i] + int(p_n[i]))
            if hi > lo:
                jobs.append((int(p_host[i]) + lo - p_dev_l[i], lo - a, hi - lo))
            i += 1
        if len(jobs) > 4 and b - a >= (2 << 20):
            global _copy_pool
            if _copy_pool is None:
                from concurrent.futures import ThreadPoolExecutor
                _copy_pool = ThreadPoolExecutor(max_workers=_COPY_THREADS, thread_name_prefix="q3tts-d2h")
            per = -(-len(jobs) // _COPY_THREADS)

            def part(g):
                for h, o, c in jobs[g:g + per]:
                    res[h:h + c] = src[o:o + c]
            list(_copy_pool.map(part, range(0, len(jobs), per)))
        else:
            for h, o, c in jobs:
                res[h:h + c] = src[o:o + c]

    if n <= (1 << 20):                                      # small results: not worth the staging
        scatter(flat.cpu().numpy(), 0, n)
        return res
    dev = flat.device
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    stage, stage_ev, lock = _stage_buffers(dev)
    with _STAGE_GUARD:
        if dev.index not in _COPY_STREAMS:
            _COPY_STREAMS[dev.index] = torch.cuda.Stream(device=dev)
        copy_stream = _COPY_STREAMS[dev.index]
    flat.record_stream(copy_stream)
    piece = _STAGE_PIECE // 4
    bounds = [(a, min(a + piece, n)) for a in range(0, n, piece)]
    ends_l = list(ends)

    def issue(k: int, j: int) -> None:
        a, b = bounds[k]
        copy_stream.wait_event(events[min(bisect.bisect_left(ends_l, b), len(events) - 1)])      # the launch that writes element b - 1
        with torch.cuda.stream(copy_stream):
            stage[j].view(torch.float32)[: b - a].copy_(flat[a:b], non_blocking=True)
            stage_ev[j].record(copy_stream)

    with lock, torch.cuda.device(dev):
        for k in range(min(2, len(bounds))):
            issue(k, k)
        for k, (a, b) in enumerate(bounds):
            j = k & 1
            stage_ev[j].synchronize()
            scatter(stage[j].numpy().view(np.float32), a, b)
            if k + 2 < len(bounds):
                issue(k + 2, j)
    return res


class Decoder:
    """One decoder instance = one q3c context on one GPU."""

    def __init__(self, cfg: DecoderConfig, weights: Dict[str, torch.Tensor], device: Union[str, torch.device] = "cuda:0",
                 precision: str = "bf16", max_positions: int = 8192, max_frames_per_launch: int = 64 * 325,
                 encoder=None, encoder_max_positions: int = 16384):
        """`encoder` (an `encoder.EncoderConfig`) adds the tokenizer's encode side to the context: `weights` must then hold
        the encoder's tensors too (`encoder.encoder_param_specs`) and the quantiser's two `input_proj`."""
        if not torch.cuda.is_available():
            raise RuntimeError("qwen3_tts_amd.Decoder needs a ROCm GPU (MI355X); there is no CPU fallback")
        self.cfg = cfg
        self.device = torch.device(device)
        self.precision = precision
        self.max_positions = int(max_positions)
        self.max_frames_per_launch = int(max_frames_per_launch)
        self.skip_context = True        # chunked decodes do not compute the conv-decoder rows under a chunk's dropped context
        self._lib = _lib.load()
        self._ctx = C.c_void_p()
        self._ws: Optional[torch.Tensor] = None
        self._taps: Dict[int, torch.Tensor] = {}
        validate_weights(cfg, weights)
        qc = _lib.make_config(cfg, precision, max_positions)
        with torch.cuda.device(self.device):
            rc = self._lib.q3c_create(C.byref(qc), C.byref(self._ctx))
            self._check(rc)
            w = dict(weights)
            w["rope.inv_freq"] = rope_inv_freq(cfg)
            self.encoder_cfg = encoder
            if encoder is not None:
                ec = _lib.make_encoder_config(encoder, encoder_max_positions)
                self._check(self._lib.q3c_encoder_configure(self._ctx, C.byref(ec)))
                d = encoder.head_dim
                w["encoder_rope.inv_freq"] = 1.0 / (encoder.rope_theta ** (torch.arange(0, d, 2, dtype=torch.float) / d))
            for name, t in w.items():
                t = t.detach().to(device="cpu", dtype=torch.float32).contiguous()
                shape = (C.c_int64 * t.dim())(*t.shape)
                self._check(self._lib.q3c_load_weight(self._ctx, name.encode(), C.c_void_p(t.data_ptr()), shape, t.dim()))
            self._check(self._lib.q3c_finalize(self._ctx))
        if os.environ.get("Q3C_TUNING"):       # measurement aid: this context's A/B switches (q3c_debug_set_tuning)
            self.set_tuning(int(os.environ["Q3C_TUNING"]))

    @classmethod
    def from_local(cls, path: str, device: Union[str, torch.device] = "cuda:0", precision: str = "bf16",
                   rename: Optional[Dict[str, str]] = None, **kwargs) -> "Decoder":
        """A decoder from a LOCAL checkpoint directory (config.json + *.safetensors; see checkpoint.load_local): the local-path
        half of `Qwen3TTSModel.from_pretrained(model_path, ...)` (reference models.py:43-48) for the decoder alone.  A hub name
        or any path that is not a directory here raises FileNotFoundError; nothing is ever downloaded."""
        from .checkpoint import load_local
        cfg, w = load_local(path, rename)
        return cls(cfg, w, device=device, precision=precision, **kwargs)

    def warmup(self, batch: int = 1, frames: int = 25, stream_pool: Optional["StreamPool"] = None, streams: int = 0,
               frames_per_push: int = 1) -> float:
        """Take every first-use cost out of the first request, as the reference's warm-up generation does for its compiled model
        (reference models.py:68-106): one decode of `batch` x `frames` random codes (loads the code object, sets the kernels'
        LDS attributes, allocates the workspace and the pinned host staging buffers) and, with a stream pool, one graph-replayed
        push of `streams` x `frames_per_push` on scratch slots (plan, descriptor table, graph instantiation).  -> seconds."""
        import time
        t0 = time.time()
        g = torch.Generator().manual_seed(0)
        codes = torch.randint(0, self.cfg.codebook_size, (batch, self.cfg.num_quantizers, frames), generator=g, dtype=torch.int64)
        self.decode([c for c in codes])
        with torch.cuda.device(self.device):
            _stage_buffers(self.device)
        if stream_pool is not None and streams > 0:
            stream_pool.warmup(streams, frames_per_push)
        torch.cuda.synchronize(self.device)
        return time.time() - t0

    # ------------------------------------------------------------------------------------------------
    def _check(self, rc: int) -> None:
        if rc != 0:
            msg = self._lib.q3c_last_error(self._ctx).decode() if self._ctx else "no context"
            if rc == -6:
                raise IndexError(msg)              # torch's embedding raises IndexError for a bad index
            raise _lib.Q3cError(rc, msg)

    def close(self) -> None:
        if getattr(self, "_ctx", None):
            for pool in list(getattr(self, "_pools", [])):
                pool.close()
            self._lib.q3c_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def open_streams(self, capacity: int) -> "StreamPool":
        """A pool of `capacity` stream slots whose causal state lives in this GPU's memory (see StreamPool)."""
        pool = StreamPool(self, capacity)
        self.__dict__.setdefault("_pools", []).append(pool)
        return pool

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self) -> C.c_void_p:
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _workspace(self, batch: int, frames: int) -> torch.Tensor:
        need = C.c_size_t()
        self._check(self._lib.q3c_workspace_bytes(self._ctx, batch, frames, C.byref(need)))
        if self._ws is None or self._ws.numel() < need.value:
            self._ws = None
            self._ws = torch.empty(need.value + 256, dtype=torch.uint8, device=self.device)
        return self._ws

    def workspace_bytes(self, batch: int, frames: int) -> int:
        need = C.c_size_t()
        self._check(self._lib.q3c_workspace_bytes(self._ctx, batch, frames, C.byref(need)))
        return int(need.value)

    def _codes_tensor(self, codes) -> torch.Tensor:
        codes = torch.as_tensor(codes)
        if codes.dim() != 3 or codes.shape[1] != self.cfg.num_quantizers:
            raise ValueError(f"Expected {self.cfg.num_quantizers} layer of codes, got {tuple(codes.shape)}")
        return codes.to(device=self.device, dtype=torch.int64).contiguous()

    # ------------------------------------------------------------------------------------------------
    def forward(self, codes: torch.Tensor, check_codes: bool = False, skip_frames: int = 0) -> torch.Tensor:
        """codes [B, Q, T] int64 -> wav [B, 1, T*1920 - 555] float32 on the device, clamped to [-1, 1].
        skip_frames > 0 (a chunk's left context, q3c_forward_skip): the samples of the first skip_frames frames come back as
        zeros and the rows of the conv decoder that only they depend on are not computed; the other samples are bit for
        bit those of skip_frames = 0."""
        codes = self._codes_tensor(codes)
        B, _, T = codes.shape
        if T < 1 or B < 1:
            raise ValueError("empty codes")
        if not 0 <= skip_frames < T:
            raise ValueError(f"skip_frames must be in [0, {T}), got {skip_frames}")
        n = self.cfg.output_length(T)
        wav = torch.empty((B, 1, n), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            ws = self._workspace(B, T)
            base = (ws.data_ptr() + 255) // 256 * 256
            if skip_frames:
                wav[..., : skip_frames * self.cfg.total_upsample].zero_()      # the library leaves them unwritten
                self._check(self._lib.q3c_forward_skip(self._ctx, C.c_void_p(codes.data_ptr()), B, T, skip_frames,
                                                       C.c_void_p(wav.data_ptr()), C.c_void_p(base),
                                                       ws.numel() - (base - ws.data_ptr()), self._stream()))
            else:
                self._check(self._lib.q3c_forward(self._ctx, C.c_void_p(codes.data_ptr()), B, T, C.c_void_p(wav.data_ptr()),
                                                  C.c_void_p(base), ws.numel() - (base - ws.data_ptr()), self._stream()))
            if check_codes:
                self._check(self._lib.q3c_check_errors(self._ctx, self._stream()))
        return wav

    __call__ = forward

    def forward_scatter(self, codes: torch.Tensor, spans: torch.Tensor, out: torch.Tensor, skip_frames: int = 0) -> None:
        """`forward(codes, skip_frames)` whose kept samples land in place (q3c_forward_scatter): spans is a device int64 tensor
        [B, 2] of (offset into the flat float32 device tensor `out`, count); item b's samples from skip_frames * total_upsample
        on, `count` of them, are written at out[offset:], nothing else is.  No per-item copies afterwards."""
        codes = self._codes_tensor(codes)
        B, _, T = codes.shape
        if not 0 <= skip_frames < T:
            raise ValueError(f"skip_frames must be in [0, {T}), got {skip_frames}")
        if spans.dtype != torch.int64 or tuple(spans.shape) != (B, 2) or not spans.is_contiguous() or spans.device != self.device:
            raise ValueError("spans must be a contiguous int64 [B, 2] tensor on the decoder's device")
        if out.dtype != torch.float32 or not out.is_contiguous() or out.device != self.device:
            raise ValueError("out must be a contiguous float32 tensor on the decoder's device")
        with torch.cuda.device(self.device):
            ws = self._workspace(B, T)
            base = (ws.data_ptr() + 255) // 256 * 256
            self._check(self._lib.q3c_forward_scatter(self._ctx, C.c_void_p(codes.data_ptr()), B, T, skip_frames, C.c_void_p(spans.data_ptr()),
                                                      C.c_void_p(out.data_ptr()), C.c_void_p(base), ws.numel() - (base - ws.data_ptr()),
                                                      self._stream()))

    def forward_static(self, codes: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
        """`forward` for callers that must not allocate device memory: codes [B, Q, T] int64 and `out` [B, 1, T*1920 - 555]
        float32 are device tensors the caller already owns, the workspace is the library's static array
        (q3c_forward_static; batch x frames limited by it), and nothing is allocated or freed.  Calls that overlap in time on
        one GPU share that array and race."""
        B, Q, T = codes.shape
        if Q != self.cfg.num_quantizers:
            raise ValueError(f"Expected {self.cfg.num_quantizers} layer of codes, got {tuple(codes.shape)}")
        if codes.dtype != torch.int64 or not codes.is_contiguous() or codes.device != self.device:
            raise ValueError("forward_static takes contiguous int64 codes already on the decoder's device")
        if out.dtype != torch.float32 or not out.is_contiguous() or out.device != self.device or \
                out.numel() != B * self.cfg.output_length(T):
            raise ValueError("forward_static needs a contiguous float32 output of B * (T*total_upsample - trim) samples on the device")
        with torch.cuda.device(self.device):
            self._check(self._lib.q3c_forward_static(self._ctx, C.c_void_p(codes.data_ptr()), B, T, C.c_void_p(out.data_ptr()),
                                                     self._stream()))
        return out

    def chunked_decode(self, codes: torch.Tensor, chunk_size: int = 300, left_context_size: int = 25) -> torch.Tensor:
        """Same result layout as the sibling's chunked_decode (:3686-3696), but the chunks - which are independent,
        each recomputing its own left context - are batched: all chunks of equal length go through one launch
        sequence."""
        codes = self._codes_tensor(codes)
        B, _, T = codes.shape
        up = self.cfg.total_upsample
        plan = self.cfg.chunk_plan(T, chunk_size, left_context_size)
        total = self.cfg.chunked_output_length(T, chunk_size, left_context_size)
        out = torch.empty((B, 1, total), dtype=torch.float32, device=self.device)
        offsets, off = [], 0
        for s, e, ctx in plan:
            offsets.append(off)
            off += self.cfg.output_length(e - s) - ctx * up
        # chunks of equal length AND equal context share a launch sequence: the context's samples are dropped below, so the
        # launch skips the rows only they depend on (forward's skip_frames)
        groups: Dict[Tuple[int, int], List[int]] = {}
        for j, (s, e, ctx) in enumerate(plan):
            groups.setdefault((e - s, ctx), []).append(j)
        # every launch writes its chunks' kept samples straight to their places in `out` (forward_scatter): one small table of
        # (destination, count) per launch row, all launches' tables uploaded in one transfer
        launches, table = [], []
        for (length, ctx), idxs in groups.items():
            per = max(1, self.max_frames_per_launch // (length * B))
            keep = self.cfg.output_length(length) - ctx * up
            for i0 in range(0, len(idxs), per):
                sub = idxs[i0:i0 + per]
                launches.append((ctx, sub, len(table)))
                for j in sub:
                    table += [(b * total + offsets[j], keep) for b in range(B)]
        spans = torch.tensor(table, dtype=torch.int64).reshape(-1, 2).to(self.device)
        flat = out.view(-1)
        for ctx, sub, t0 in launches:
            stacked = torch.cat([codes[..., plan[j][0]:plan[j][1]] for j in sub], dim=0) if len(sub) > 1 \
                else codes[..., plan[sub[0]][0]:plan[sub[0]][1]]
            self._forward_kept(stacked, ctx, spans[t0:t0 + len(sub) * B], flat)
        return out

    def _forward_kept(self, codes: torch.Tensor, ctx: int, spans: torch.Tensor, flat: torch.Tensor) -> None:
        """One launch sequence whose rows drop `ctx` context frames and keep spans[:, 1] samples each, placed at spans[:, 0] of
        `flat`.  With skip_context off the context's conv-decoder rows are computed as well (same samples, bit for bit)."""
        if self.skip_context or ctx == 0:
            self.forward_scatter(codes, spans, flat, skip_frames=ctx)
        else:                                    # measurement aid: the whole forward, then the placement
            wav = self.forward(codes)
            up = self.cfg.total_upsample
            for n, (dst, cnt) in enumerate(spans.cpu().tolist()):
                flat[dst:dst + cnt] = wav[n, 0, ctx * up: ctx * up + cnt]

    def transformer(self, hidden: torch.Tensor) -> torch.Tensor:
        """The transformer stack alone: hidden states [B, T, latent_dim] (fp32) -> the same shape."""
        x = hidden.to(self.device, torch.float32).contiguous()
        B, T, D = x.shape
        if D != self.cfg.latent_dim:
            raise ValueError(f"Expected {self.cfg.latent_dim} channels, got {D}")
        y = torch.empty_like(x)
        ws = self._workspace(B, T)
        with torch.cuda.device(self.device):
            base = (ws.data_ptr() + 255) // 256 * 256
            self._check(self._lib.q3c_transformer(self._ctx, C.c_void_p(x.data_ptr()), B, T, C.c_void_p(y.data_ptr()),
                                                  C.c_void_p(base), ws.numel() - (base - ws.data_ptr()), self._stream()))
        return y

    def rvq_dequant(self, codes: torch.Tensor) -> torch.Tensor:
        """Ordered fp32 gather-and-sum of each residual group: codes [B, Q, T] -> [B, T, G, codebook_dim]."""
        codes = self._codes_tensor(codes)
        B, _, T = codes.shape
        G = 2 if self.cfg.num_quantizers > self.cfg.num_semantic_quantizers else 1
        out = torch.empty((B, T, G, self.cfg.codebook_dim), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            self._check(self._lib.q3c_rvq_dequant_f32(self._ctx, C.c_void_p(codes.data_ptr()), B, T, C.c_void_p(out.data_ptr()),
                                                      self._stream()))
            self._check(self._lib.q3c_check_errors(self._ctx, self._stream()))
        return out

    def rvq_encode(self, embeddings: torch.Tensor) -> torch.Tensor:
        """The quantiser's encode direction: embeddings [B, T, D_in] (fp32, channels-last) -> codes [B, Q, T] int64 on the
        device.  D_in is the input width of the groups' `input_proj` when the weights held one, else the code-book row
        width."""
        x = embeddings.to(self.device, torch.float32).contiguous()
        if x.dim() != 3:
            raise ValueError(f"Expected [B, T, D] embeddings, got {tuple(x.shape)}")
        B, T, Din = x.shape
        codes = torch.empty((B, self.cfg.num_quantizers, T), dtype=torch.int64, device=self.device)
        with torch.cuda.device(self.device):
            self._check(self._lib.q3c_rvq_encode(self._ctx, C.c_void_p(x.data_ptr()), B, T, Din, C.c_void_p(codes.data_ptr()),
                                                 self._stream()))
        return codes

    # ------------------------------------------------------------------------------------------------
    def encode(self, audio: torch.Tensor, return_embeddings: bool = False):
        """The tokenizer's encode side: mono 24 kHz audio [B, L] (or [B, 1, L]) -> codes [B, Q, frames] int64 on the device,
        frames = `encoder_cfg.frames(L)` (reference server.py:228-232 reaches it through create_voice_clone_prompt; arithmetic
        of the sibling MimiModel.encode).  With return_embeddings also what the quantiser encoded, [B, frames, hidden] fp32."""
        if self.encoder_cfg is None:
            raise RuntimeError("this Decoder was built without an encoder configuration")
        a = audio.to(self.device, torch.float32)
        if a.dim() == 3:
            if a.shape[1] != 1:
                raise ValueError(f"Expected mono audio [B, 1, L], got {tuple(a.shape)}")
            a = a[:, 0]
        if a.dim() != 2 or a.shape[1] < 1:
            raise ValueError(f"Expected audio of shape [B, L], got {tuple(audio.shape)}")
        a = a.contiguous()
        B, L = a.shape
        T = self.encoder_cfg.frames(L)
        codes = torch.empty((B, self.cfg.num_quantizers, T), dtype=torch.int64, device=self.device)
        emb = torch.empty((B, T, self.encoder_cfg.hidden_size), dtype=torch.float32, device=self.device) if return_embeddings else None
        with torch.cuda.device(self.device):
            need = C.c_size_t()
            self._check(self._lib.q3c_encode_workspace_bytes(self._ctx, B, L, C.byref(need)))
            ws = self._ws                             # one workspace per decoder, shared with the decode direction: grown, never shrunk
            if ws is None or ws.numel() < need.value + 256:
                self._ws = None
                ws = self._ws = torch.empty(need.value + 256, dtype=torch.uint8, device=self.device)
            base = (ws.data_ptr() + 255) // 256 * 256
            self._check(self._lib.q3c_encode(self._ctx, C.c_void_p(a.data_ptr()), B, L, C.c_void_p(codes.data_ptr()),
                                             C.c_void_p(emb.data_ptr()) if emb is not None else None, C.c_void_p(base), need.value,
                                             self._stream()))
        return (codes, emb) if return_embeddings else codes

    def _normalize_items(self, encoded: CodesLike) -> List[torch.Tensor]:
        """Accepts a [T, Q] / [Q, T] matrix, a [B, T, Q] / [B, Q, T] tensor, {"audio_codes": ...}, or a list of
        those; returns per-utterance [Q, T_i] int64 tensors."""
        Q = self.cfg.num_quantizers

        def one(x) -> List[torch.Tensor]:
            if isinstance(x, dict):
                if "audio_codes" not in x:
                    raise ValueError("dict input needs an 'audio_codes' entry")
                return one(x["audio_codes"])
            t = torch.as_tensor(x)
            if t.dim() == 3:
                out = []
                for i in range(t.shape[0]):
                    out += one(t[i])
                return out
            if t.dim() != 2:
                raise ValueError(f"Expected 2-D or 3-D codes, got {tuple(t.shape)}")
            if t.shape[1] == Q:            # [T, Q] (a square [Q, Q] matrix is read this way too)
                t = t.t()
            elif t.shape[0] != Q:
                raise ValueError(f"Expected {Q} layer of codes, got {tuple(t.shape)}")
            return [t.to(torch.int64)]

        if isinstance(encoded, (list, tuple)):
            items: List[torch.Tensor] = []
            for e in encoded:
                items += one(e)
            return items
        return one(encoded)

    def decode_device(self, encoded: CodesLike, chunk_size: Optional[int] = None, left_context_size: Optional[int] = None,
                      pad_to: Optional[int] = None, _by_launch: bool = False):
        """The decode of `decode` without the copy to the host: -> (flat float32 device tensor holding every
        utterance's samples back to back, start offset per utterance, length per utterance).
        `_by_launch` (used by `decode`): the device tensor holds the samples launch by launch instead - what a launch
        produces is contiguous, so it can leave for the host while the next launches run - and a fourth value describes the
        layout: (end offset per launch, event per launch, and per piece its device offset, host offset and length)."""
        cs = self.cfg.chunk_size if chunk_size is None else chunk_size
        lc = self.cfg.left_context_size if left_context_size is None else left_context_size
        items = self._normalize_items(encoded)
        up, Q = self.cfg.total_upsample, self.cfg.num_quantizers
        lengths, starts, jobs = [], [], []      # jobs: (item, s, e, ctx, offset_in_item)
        total = 0
        for i, c in enumerate(items):
            T = int(c.shape[1])
            if T < 1:
                raise ValueError("empty code sequence")
            n = self.cfg.chunked_output_length(T, cs, lc)
            starts.append(total)
            lengths.append(n)
            total += n
            off = 0
            for s, e, ctx in self.cfg.chunk_plan(T, cs, lc):
                jobs.append((i, s, e, ctx, off))
                off += self.cfg.output_length(e - s) - ctx * up
        out = torch.empty(total, dtype=torch.float32, device=self.device)
        if not items:
            return (out, starts, lengths, ([], [], np.zeros(0, np.int64), np.zeros(0, np.int64), np.zeros(0, np.int64))) if _by_launch \
                else (out, starts, lengths)
        flat_codes, bases = None, None
        if len(items) > 1 and (all(c.device.type == "cpu" for c in items) or all(c.device == self.device for c in items)):
            # one tensor on the device (for host codes: one copy), then views
            # (a zero column at the end: the source of every padding frame when a launch's codes are gathered in one go)
            flat_codes = torch.cat(list(items) + [torch.zeros((Q, 1), dtype=torch.int64, device=items[0].device)], dim=1).to(self.device)
            dev_items, bases, o = [], [], 0
            for c in items:
                dev_items.append(flat_codes[:, o:o + int(c.shape[1])])
                bases.append(o)
                o += int(c.shape[1])
        else:
            dev_items = [c.to(self.device) for c in items]
        # Launches: (padded length, context frames to skip, pieces).  Pieces of one launch share the context (its rows are
        # skipped, forward's skip_frames) and are padded with code 0 to the launch's length.
        launches: List[Tuple[int, int, List[int]]] = []
        if pad_to is None:
            # Padding frames are decoded for nothing, but finer buckets mean more and smaller launches.  Measured on 2 000
            # utterances of 2-30 s (tools/mixed_ab.py): 10-frame buckets 1.6 % faster than 25-frame ones, 5-frame ones 1 % slower;
            # pieces sorted by length and cut into full launches of neighbours no faster than 25-frame buckets (a full launch
            # of short pieces spans too many lengths).  10 frames when that leaves the buckets 4096 frames on average, else 25.
            total_frames = sum(e - s for _i, s, e, _c, _o in jobs)
            nb10 = len({(-(-(e - s) // 10) * 10, ctx) for _i, s, e, ctx, _o in jobs})
            pad_to = 10 if total_frames >= 4096 * nb10 else 25
        buckets: Dict[Tuple[int, int], List[int]] = {}
        for j, (_i, s, e, ctx, _off) in enumerate(jobs):
            buckets.setdefault((-(-(e - s) // pad_to) * pad_to, ctx), []).append(j)
        for (Lb, skip), idxs in sorted(buckets.items()):
            per = max(1, self.max_frames_per_launch // Lb)
            for i0 in range(0, len(idxs), per):
                launches.append((Lb, skip, idxs[i0:i0 + per]))
        # largest first: the samples of the LAST launch are the only ones whose way to the host nothing can hide
        launches.sort(key=lambda l: -l[0] * len(l[2]))
        # where every piece's kept samples go: (offset into `out`, count) per launch row, one table for all launches
        span_rows, span_off = [], []
        host_rows, launch_ends, dev_o = [], [], 0
        for Lb, _skip, sub in launches:
            span_off.append(len(span_rows))
            for j in sub:
                i, s, e, ctx, off = jobs[j]
                cnt = self.cfg.output_length(e - s) - ctx * up
                span_rows.append((dev_o if _by_launch else starts[i] + off, cnt))
                host_rows.append(starts[i] + off)
                dev_o += cnt
            launch_ends.append(dev_o)
        spans_host = np.asarray(span_rows, dtype=np.int64).reshape(-1, 2)
        idx_dev, idx_off = None, []
        if flat_codes is not None:
            # every launch's gather indices built up front and copied to the device in ONE transfer: a blocking copy per launch
            # would make the host wait for the launch before it, and the GPU for the host's next preparation
            zero_col = int(flat_codes.shape[1]) - 1
            total_idx = sum(len(sub) * Lb for Lb, _skip, sub in launches)
            idx_all = np.full(total_idx, zero_col, dtype=np.int64)
            o = 0
            for Lb, _skip, sub in launches:
                idx_off.append(o)
                for n, j in enumerate(sub):
                    i, s, e, _ctx, _off = jobs[j]
                    idx_all[o + n * Lb: o + n * Lb + (e - s)] = np.arange(bases[i] + s, bases[i] + e)
                o += len(sub) * Lb
            # the gather indices and the placement table cross PCIe as ONE transfer
            both = torch.from_numpy(np.concatenate([idx_all, spans_host.reshape(-1)])).to(self.device)
            idx_dev, spans_dev = both[:total_idx], both[total_idx:].view(-1, 2)
        else:
            spans_dev = torch.from_numpy(spans_host).to(self.device)
        launch_events: List["torch.cuda.Event"] = []
        for k, (Lb, skip, sub) in enumerate(launches):
            if idx_dev is not None:
                idx = idx_dev[idx_off[k]: idx_off[k] + len(sub) * Lb].view(len(sub), Lb)
                batch = flat_codes[:, idx].permute(1, 0, 2).contiguous()
            else:
                batch = torch.zeros((len(sub), Q, Lb), dtype=torch.int64, device=self.device)
                for n, j in enumerate(sub):
                    i, s, e, _ctx, _off = jobs[j]
                    batch[n, :, : e - s] = dev_items[i][:, s:e]
            self._forward_kept(batch, skip, spans_dev[span_off[k]: span_off[k] + len(sub)], out)
            if _by_launch:
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(self.device))
                launch_events.append(ev)
        if _by_launch:      # no error check here: it waits for the stream, and the caller's copies are meant to run under the launches
            return out, starts, lengths, (launch_ends, launch_events, spans_host[:, 0].copy(), np.asarray(host_rows, dtype=np.int64),
                                          spans_host[:, 1].copy())
        with torch.cuda.device(self.device):
            self._check(self._lib.q3c_check_errors(self._ctx, self._stream()))
        return out, starts, lengths

    def decode(self, encoded: CodesLike, chunk_size: Optional[int] = None, left_context_size: Optional[int] = None,
               pad_to: Optional[int] = None) -> Tuple[List[np.ndarray], int]:
        """-> (list of float32 1-D numpy waveforms, 24000).  Utterances of different length are decoded together:
        every (utterance, chunk) piece is padded with code 0 to a multiple of `pad_to` frames (None: 25, or 10 when
        there are enough pieces to fill the finer buckets), pieces of equal padded
        length share a launch sequence, and each piece keeps only its own prefix of the output (a frame never
        influences samples that precede it by more than the fixed 555-sample lookahead)."""
        out, starts, lengths, (ends, events, p_dev, p_host, p_n) = self.decode_device(encoded, chunk_size, left_context_size, pad_to,
                                                                                      _by_launch=True)
        try:
            host = to_host_by_launch(out, sum(lengths), ends, events, p_dev, p_host, p_n)
        finally:            # the device-side error flag (a code outside the code book), read once everything has run
            with torch.cuda.device(self.device):
                self._check(self._lib.q3c_check_errors(self._ctx, self._stream()))
        return [host[s:s + n] for s, n in zip(starts, lengths)], self.cfg.sample_rate

    def _to_host(self, flat: torch.Tensor) -> np.ndarray:
        """Device samples (float32 or int16) -> one numpy array (see `to_host`)."""
        return to_host(flat)

    # ------------------------------------------------------------------------------------------------
    def profile(self, on: bool, detail: bool = False) -> None:
        """Bracket every kernel launch of the following forwards with HIP events (see q3c_profile_enable);
        detail=True keeps one entry per (kernel, layer shape) instead of one per kernel."""
        self._check(self._lib.q3c_profile_enable(self._ctx, (2 if detail else 1) if on else 0))

    def profile_read(self) -> List[dict]:
        """Per-kernel launches, summed milliseconds and algorithmic flops / bytes since profile(True)."""
        cap = 512
        arr = (_lib.Q3cProfileEntry * cap)()
        n = C.c_int32()
        self._check(self._lib.q3c_profile_read(self._ctx, arr, cap, C.byref(n)))
        return [dict(kernel=arr[i].kernel.decode(), launches=int(arr[i].launches), ms=float(arr[i].ms),
                     flops=float(arr[i].flops), bytes=float(arr[i].bytes)) for i in range(n.value)]

    def set_tuning(self, bits: int) -> None:
        """Measurement switches of THIS decoder's context (include/q3c.h, q3c_debug_set_tuning); 0 = defaults."""
        self._check(self._lib.q3c_debug_set_tuning(self._ctx, int(bits)))

    def set_tap(self, stage: int, elements: int) -> torch.Tensor:
        """Debug/parity aid: capture a stage's output of the following forwards as fp32 (see include/q3c.h)."""
        buf = torch.zeros(elements, dtype=torch.float32, device=self.device)
        self._check(self._lib.q3c_set_tap(self._ctx, stage, C.c_void_p(buf.data_ptr()), elements))
        self._taps[stage] = buf
        return buf

    def clear_taps(self) -> None:
        for stage in list(self._taps):
            self._lib.q3c_set_tap(self._ctx, stage, None, 0)
        self._taps.clear()


class StreamPool:
    """Incremental decode of many concurrent streams.

    The reference has no incremental decode: it runs the whole utterance through the tokenizer's decode and sends one
    SSE `audio` event (reference server.py:299-335), and the sibling's chunked_decode (:3686-3696) recomputes 25 frames
    per chunk.  Here a stream is advanced a few frames per push and the concatenation of its pushes equals the
    un-chunked forward of all its frames (:3672-3684): the rows every causal layer looks back into are carried in a
    per-slot state record in device memory, and one push serves any set of slots, at any positions, in one launch
    sequence.

        pool = decoder.open_streams(256)
        sid = pool.open()
        wav = pool.push([sid, ...], codes[S, Q, n])      # list of 1-D float32 device tensors
        pool.close_stream(sid)

    A stream's first push returns n*1920 - 555 samples (the decode's fixed lookahead), later pushes n*1920.
    """

    def __init__(self, decoder: "Decoder", capacity: int):
        self.dec = decoder
        self.capacity = int(capacity)
        self._lib = decoder._lib
        self._pool = C.c_void_p()
        self._ws: Optional[torch.Tensor] = None
        self._free = list(range(self.capacity - 1, -1, -1))
        self._frames: Dict[int, int] = {}            # open slot -> frames pushed so far
        self._ids_dev: Dict[tuple, torch.Tensor] = {}    # id tuples already on the device (a stable set of streams pushes again and again)
        with torch.cuda.device(decoder.device):
            decoder._check(self._lib.q3c_stream_pool_create(decoder._ctx, self.capacity, C.byref(self._pool)))

    def close(self) -> None:
        if getattr(self, "_pool", None):
            self._lib.q3c_stream_pool_destroy(self._pool)
            self._pool = C.c_void_p()
            if self in getattr(self.dec, "_pools", []):
                self.dec._pools.remove(self)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def state_bytes_per_stream(self) -> int:
        n = C.c_size_t()
        self.dec._check(self._lib.q3c_stream_state_bytes(self._pool, C.byref(n)))
        return int(n.value)

    def open(self) -> int:
        """Claims a slot and zeroes its state; returns the stream id."""
        return self.open_many(1)[0]

    def open_many(self, count: int) -> List[int]:
        if count > len(self._free):
            raise RuntimeError(f"stream pool is full ({self.capacity} slots)")
        ids = [self._free.pop() for _ in range(count)]
        dev = torch.tensor(ids, dtype=torch.int32, device=self.dec.device)
        with torch.cuda.device(self.dec.device):
            self.dec._check(self._lib.q3c_stream_reset(self._pool, C.c_void_p(dev.data_ptr()), count, self.dec._stream()))
        for i in ids:
            self._frames[i] = 0
        return ids

    def close_stream(self, sid: int) -> None:
        if sid not in self._frames:
            raise KeyError(f"stream {sid} is not open")
        del self._frames[sid]
        self._free.append(sid)

    def frames_pushed(self, sid: int) -> int:
        return self._frames[sid]

    def _device_ids(self, ids: tuple) -> torch.Tensor:
        """int32 device tensor of a tuple of stream ids; the 32 most recently used tuples stay on the device, so a set of
        streams that pushes tick after tick pays the (blocking, pageable) upload once."""
        dev = self._ids_dev.pop(ids, None)
        if dev is None:
            dev = torch.tensor(ids, dtype=torch.int32, device=self.dec.device)
            while len(self._ids_dev) >= 32:
                self._ids_dev.pop(next(iter(self._ids_dev)))
        self._ids_dev[ids] = dev                     # (re)inserted last: dicts keep insertion order
        return dev

    def push_raw(self, ids_dev: torch.Tensor, codes: torch.Tensor, out: Optional[torch.Tensor] = None, graph: bool = False) -> torch.Tensor:
        """The launch sequence alone: ids_dev int32 [S] and codes int64 [S, Q, n] on the device -> [S, n*1920] float32
        (a stream's first push starts with 555 blank samples).  No host bookkeeping and no synchronisation.
        graph=True replays the push from a HIP graph cached for these exact tensors (q3c_stream_push_graph): meant for a
        serving loop that refills a fixed set of `ids_dev` / `codes` / `out` buffers."""
        dec = self.dec
        S, Q, n = codes.shape
        up = dec.cfg.total_upsample
        if out is None:
            out = torch.empty((S, n * up), dtype=torch.float32, device=dec.device)
        with torch.cuda.device(dec.device):
            need = C.c_size_t()
            dec._check(self._lib.q3c_stream_workspace_bytes(self._pool, S, n, C.byref(need)))
            if self._ws is None or self._ws.numel() < need.value + 256:
                self._ws = None
                self._ws = torch.empty(need.value + 256, dtype=torch.uint8, device=dec.device)
            base = (self._ws.data_ptr() + 255) // 256 * 256
            fn = self._lib.q3c_stream_push_graph if graph else self._lib.q3c_stream_push
            dec._check(fn(self._pool, C.c_void_p(ids_dev.data_ptr()), S, C.c_void_p(codes.data_ptr()), n, C.c_void_p(out.data_ptr()),
                          C.c_void_p(base), self._ws.numel() - (base - self._ws.data_ptr()), dec._stream()))
        return out

    def push(self, ids: Sequence[int], codes: torch.Tensor, check: bool = False) -> List[torch.Tensor]:
        """Advance the streams `ids` (distinct, open) by n frames each: codes [S, Q, n].  Returns one 1-D float32
        device tensor per stream with that stream's new samples."""
        dec = self.dec
        ids = [int(i) for i in ids]
        codes = dec._codes_tensor(codes)
        S, _, n = codes.shape
        if S != len(ids) or n < 1:
            raise ValueError(f"{len(ids)} stream ids for codes of shape {tuple(codes.shape)}")
        if len(set(ids)) != S:
            raise ValueError("stream ids in one push must be distinct")
        for i in ids:
            if i not in self._frames:
                raise KeyError(f"stream {i} is not open")
            if self._frames[i] + n > dec.max_positions:
                raise ValueError(f"stream {i} would exceed max_positions = {dec.max_positions} frames")
        wav = self.push_raw(self._device_ids(tuple(ids)), codes)
        if check:
            with torch.cuda.device(dec.device):
                dec._check(self._lib.q3c_check_errors(dec._ctx, dec._stream()))
        out = []
        for b, i in enumerate(ids):
            out.append(wav[b, dec.cfg.trim:] if self._frames[i] == 0 else wav[b])
            self._frames[i] += n
        return out


# ---- operator-level entry points (device fp32 tensors, channels-last) ------------------------------------
def causal_conv1d(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], dilation: int = 1,
                  precision: str = "f32") -> torch.Tensor:
    """x [B, L, Cin] (device fp32) -> [B, L, Cout]; weight [Cout, Cin, k] in the torch layout."""
    lib = _lib.load()
    B, L, cin = x.shape
    cout, _, k = weight.shape
    x = x.contiguous().float()
    w = weight.detach().cpu().float().contiguous()
    b = bias.detach().cpu().float().contiguous() if bias is not None else None
    y = torch.empty((B, L, cout), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        rc = lib.q3c_causal_conv1d(C.c_void_p(x.data_ptr()), B, L, cin, cout, k, dilation, C.c_void_p(w.data_ptr()),
                                   C.c_void_p(b.data_ptr()) if b is not None else None, C.c_void_p(y.data_ptr()),
                                   _lib.PRECISION[precision], C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream))
    if rc:
        raise _lib.Q3cError(rc, "q3c_causal_conv1d failed")
    return y


def causal_trans_conv1d(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], stride: int,
                        precision: str = "f32") -> torch.Tensor:
    """x [B, L, Cin] -> [B, (L-1)*s - k + 2s, Cout]; weight [Cin, Cout, k] in the torch layout."""
    lib = _lib.load()
    B, L, cin = x.shape
    _, cout, k = weight.shape
    lout = (L - 1) * stride - k + 2 * stride
    x = x.contiguous().float()
    w = weight.detach().cpu().float().contiguous()
    b = bias.detach().cpu().float().contiguous() if bias is not None else None
    y = torch.empty((B, max(lout, 0), cout), dtype=torch.float32, device=x.device)
    if lout <= 0:
        return y
    with torch.cuda.device(x.device):
        rc = lib.q3c_causal_tconv1d(C.c_void_p(x.data_ptr()), B, L, cin, cout, k, stride, C.c_void_p(w.data_ptr()),
                                    C.c_void_p(b.data_ptr()) if b is not None else None, C.c_void_p(y.data_ptr()),
                                    _lib.PRECISION[precision], C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream))
    if rc:
        raise _lib.Q3cError(rc, "q3c_causal_tconv1d failed")
    return y


def snake_beta(x: torch.Tensor, alpha: torch.Tensor, beta: torch.Tensor, eps: float = 1e-9, precision: str = "f32") -> torch.Tensor:
    """x [..., C] (device fp32, channels-last).  precision "bf16": bf16 storage and the hardware sine, as in the bf16 decode."""
    lib = _lib.load()
    Cn = x.shape[-1]
    x = x.contiguous().float()
    a = alpha.detach().cpu().float().contiguous()
    b = beta.detach().cpu().float().contiguous()
    y = torch.empty_like(x)
    with torch.cuda.device(x.device):
        rc = lib.q3c_snake_beta(C.c_void_p(x.data_ptr()), x.numel() // Cn, Cn, C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()),
                                eps, C.c_void_p(y.data_ptr()), _lib.PRECISION[precision],
                                C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream))
    if rc:
        raise _lib.Q3cError(rc, "q3c_snake_beta failed")
    return y


def _host(t: torch.Tensor) -> torch.Tensor:
    return t.detach().cpu().float().contiguous()


def residual_unit(x: torch.Tensor, w: dict, prefix: str, dilation: int, eps: float = 1e-9, precision: str = "f32") -> torch.Tensor:
    """DecoderResidualUnit on x [B, L, C] (device fp32, channels-last); `w` holds the unit's parameters under the
    torch names (`act1.alpha`, `conv1.conv.weight`, ...) behind `prefix`."""
    lib = _lib.load()
    B, L, Cn = x.shape
    x = x.contiguous().float()
    names = ["act1.alpha", "act1.beta", "conv1.conv.weight", "conv1.conv.bias", "act2.alpha", "act2.beta", "conv2.conv.weight",
             "conv2.conv.bias"]
    hw = [_host(torch.as_tensor(w[prefix + n])) for n in names]
    y = torch.empty_like(x)
    with torch.cuda.device(x.device):
        rc = lib.q3c_residual_unit(C.c_void_p(x.data_ptr()), B, L, Cn, dilation, *[C.c_void_p(t.data_ptr()) for t in hw], eps,
                                   C.c_void_p(y.data_ptr()), _lib.PRECISION[precision],
                                   C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream))
    if rc:
        raise _lib.Q3cError(rc, "q3c_residual_unit failed")
    return y


def convnext_block(x: torch.Tensor, w: dict, prefix: str, eps: float = 1e-6, precision: str = "f32") -> torch.Tensor:
    """ConvNeXtBlock on x [B, L, C] (device fp32, channels-last); parameters under the torch names behind `prefix`."""
    lib = _lib.load()
    B, L, Cn = x.shape
    x = x.contiguous().float()
    names = ["dwconv.conv.weight", "dwconv.conv.bias", "norm.weight", "norm.bias", "pwconv1.weight", "pwconv1.bias",
             "pwconv2.weight", "pwconv2.bias", "gamma"]
    hw = [_host(torch.as_tensor(w[prefix + n])) for n in names]
    y = torch.empty_like(x)
    with torch.cuda.device(x.device):
        rc = lib.q3c_convnext_block(C.c_void_p(x.data_ptr()), B, L, Cn, *[C.c_void_p(t.data_ptr()) for t in hw], eps,
                                    C.c_void_p(y.data_ptr()), _lib.PRECISION[precision],
                                    C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream))
    if rc:
        raise _lib.Q3cError(rc, "q3c_convnext_block failed")
    return y
