"""Cross-request batching of decodes.

The reference decodes one sentence at a time inside one request at a time (`server.py:343-398`: a loop over sentences,
each `generate_voice_clone` call ending in its own decode; the handlers block the event loop while they run).  The
decode kernels want the opposite: utterances of any length from any number of requests in ONE launch sequence
(`Decoder.decode` already pads and buckets a ragged list).  `DecodeBatcher` is the piece between the two: request
handlers on any thread call `submit()` / `decode()`, a single worker thread owns the decoder, gathers what has
arrived within a short window and decodes it together, and every caller gets exactly its own waveforms back.

Only host logic lives here; the arithmetic is `Decoder.decode`'s.  A request's samples do not depend on its batch
neighbours beyond fp32 rounding (another mix pads to other lengths, which regroups a few fp32 reductions: <= 1e-5 in the
fp32 path, `tests/test_gpu_parity.py::test_batcher_equals_direct_decode`)."""
from __future__ import annotations

import queue
import threading
import time
from concurrent.futures import Future
from typing import List, Optional, Tuple


class DecodeBatcher:
    """decoder: anything with `decode(list_of_code_tensors) -> (list_of_waveforms, sample_rate)` and
    `_normalize_items(encoded) -> list of [Q, T] tensors` (`qwen3_tts_amd.decoder.Decoder`).

    max_wait_ms: how long the worker keeps collecting after the first request of a batch arrived.
    max_frames:  a batch is closed once it holds this many code frames (12.5 per audio-second).
    """

    def __init__(self, decoder, max_wait_ms: float = 2.0, max_frames: int = 64 * 375, max_requests: int = 1024):
        self._dec = decoder
        self._wait = max_wait_ms / 1e3
        self._max_frames = int(max_frames)
        self._max_requests = int(max_requests)
        self._q: "queue.Queue[Optional[Tuple[list, Future]]]" = queue.Queue()
        self._closed = False
        self.batches = 0              # launch groups run so far
        self.requests = 0             # requests served so far
        self.utterances = 0
        self._worker = threading.Thread(target=self._run, name="q3tts-decode-batcher", daemon=True)
        self._worker.start()

    # ---- caller side ---------------------------------------------------------------------------------
    def submit(self, encoded) -> Future:
        """Queue one request (same input forms as `Decoder.decode`).  Malformed input raises here, in the caller's
        thread, as a direct call would (`ValueError` for a wrong code-book count); the future resolves to
        `(list of float32 waveforms, sample_rate)`."""
        if self._closed:
            raise RuntimeError("DecodeBatcher is closed")
        items = self._dec._normalize_items(encoded)
        bins = getattr(getattr(self._dec, "cfg", None), "codebook_size", None)
        for c in items:
            if int(c.shape[1]) < 1:
                raise ValueError("empty code sequence")
            # codes that live on the host are range-checked here (as torch's embedding would raise in a direct call): a bad
            # request then fails alone and at once, instead of failing its whole window on the device and making every
            # neighbour decode a second time.  Device tensors are left to the device-side check (no synchronisation here).
            if bins is not None and not c.is_cuda and c.numel() and (int(c.min()) < 0 or int(c.max()) >= bins):
                raise IndexError(f"code out of range [0, {bins})")
        fut: Future = Future()
        self._q.put((items, fut))
        return fut

    def decode(self, encoded):
        """Blocking form with `Decoder.decode`'s signature and result."""
        return self.submit(encoded).result()

    def close(self) -> None:
        if not self._closed:
            self._closed = True
            self._q.put(None)
            self._worker.join()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # ---- worker ----------------------------------------------------------------------------------------
    def _collect(self, first) -> Tuple[List[Tuple[list, Future]], bool]:
        batch = [first]
        frames = sum(int(c.shape[1]) for c in first[0])
        deadline = time.monotonic() + self._wait
        stop = False
        while frames < self._max_frames and len(batch) < self._max_requests:
            left = deadline - time.monotonic()
            try:
                nxt = self._q.get(timeout=left) if left > 0 else self._q.get_nowait()
            except queue.Empty:
                break
            if nxt is None:
                stop = True
                break
            batch.append(nxt)
            frames += sum(int(c.shape[1]) for c in nxt[0])
        return batch, stop

    def _serve(self, batch: List[Tuple[list, Future]]) -> None:
        live = [(items, fut) for items, fut in batch if fut.set_running_or_notify_cancel()]
        if not live:
            return
        flat = [c for items, _ in live for c in items]
        try:
            wavs, sr = self._dec.decode(flat)
        except Exception as e:                      # one bad request must not fail its neighbours: isolate it
            if len(live) == 1:
                live[0][1].set_exception(e)
            else:
                for items, fut in live:
                    try:
                        fut.set_result(self._dec.decode(items))
                    except Exception as e1:
                        fut.set_exception(e1)
            self.batches += 1
            self.requests += len(live)
            return
        k = 0
        for items, fut in live:
            fut.set_result((wavs[k:k + len(items)], sr))
            k += len(items)
        self.batches += 1
        self.requests += len(live)
        self.utterances += len(flat)

    def _run(self) -> None:
        while True:
            first = self._q.get()
            if first is None:
                break
            batch, stop = self._collect(first)
            self._serve(batch)
            if stop:
                break
        # requests that raced with close()
        while True:
            try:
                rest = self._q.get_nowait()
            except queue.Empty:
                break
            if rest is not None:
                rest[1].set_exception(RuntimeError("DecodeBatcher is closed"))


class StreamBatcher:
    """Concurrent streams advance in ONE push per tick.

    The reference's SSE route (`server.py:425-486`) serves one request at a time and decodes its whole utterance at once.  With
    the streaming decode a request is advanced chunk by chunk - and the push kernels serve any set of streams, at any
    positions, in one launch sequence.  Request handlers on any thread call `advance(sid, codes)`; the single worker thread
    gathers what is pending, issues one `pool.push` per distinct chunk length for ALL of those streams, converts the new
    samples to 16-bit PCM on the GPU, brings them to the host in one copy and hands every stream its own piece.  A stream has at
    most one chunk pending (its handler waits for the piece before it sends the next), so a tick never needs to wait longer
    than until every open stream has reported - or `max_wait_ms`, whichever comes first.

    pool: `qwen3_tts_amd.decoder.StreamPool`; lock: held around every use of the pool's decoder context (the context must not
    be used from two threads at once, and whole-utterance decodes share it); to_pcm16(device float tensor) -> device int16."""

    def __init__(self, pool, lock: Optional[threading.Lock] = None, max_wait_ms: float = 2.0, to_pcm16=None):
        self._pool = pool
        self._lock = lock or threading.Lock()
        self._wait = max_wait_ms / 1e3
        self._to_pcm16 = to_pcm16
        self._q: "queue.Queue[Optional[Tuple[int, object, Future]]]" = queue.Queue()
        self._closed = False
        self._open = 0                 # streams opened through this batcher and not yet closed
        self._count_lock = threading.Lock()
        self.pushes = 0                # pool.push calls so far
        self.chunks = 0                # stream chunks served so far
        self.ticks = 0
        self._worker = threading.Thread(target=self._run, name="q3tts-stream-batcher", daemon=True)
        self._worker.start()

    # ---- caller side ---------------------------------------------------------------------------------------------------
    def open(self) -> int:
        with self._lock:
            sid = self._pool.open()
        with self._count_lock:
            self._open += 1
        return sid

    def close_stream(self, sid: int) -> None:
        with self._lock:
            self._pool.close_stream(sid)
        with self._count_lock:
            self._open -= 1

    def submit(self, sid: int, codes) -> Future:
        """Queue the next chunk of stream `sid`: codes [Q, n] int64 (host or device).  Malformed chunks raise here, in the
        caller's thread - wrong code-book count (ValueError), a code outside the code book (IndexError, as the decode's own
        check raises), a stream past max_positions (ValueError) - so a bad request never reaches a push that other streams
        share.  The future resolves to the chunk's samples (int16 numpy, or float32 without a converter)."""
        import torch
        if self._closed:
            raise RuntimeError("StreamBatcher is closed")
        dec = self._pool.dec
        c = torch.as_tensor(codes)
        if c.dim() != 2 or c.shape[0] != dec.cfg.num_quantizers or c.shape[1] < 1:
            raise ValueError(f"Expected [{dec.cfg.num_quantizers}, n >= 1] codes, got {tuple(c.shape)}")
        if c.dtype != torch.int64:
            c = c.to(torch.int64)
        lo, hi = int(c.min()), int(c.max())
        if lo < 0 or hi >= dec.cfg.codebook_size:
            raise IndexError(f"a code index was outside [0, {dec.cfg.codebook_size})")
        if self._pool.frames_pushed(sid) + int(c.shape[1]) > dec.max_positions:
            raise ValueError(f"stream {sid} would exceed max_positions = {dec.max_positions} frames")
        fut: Future = Future()
        self._q.put((int(sid), c, fut))
        return fut

    def advance(self, sid: int, codes):
        return self.submit(sid, codes).result()

    def close(self) -> None:
        if not self._closed:
            self._closed = True
            self._q.put(None)
            self._worker.join()

    # ---- worker ----------------------------------------------------------------------------------------------------------
    def _collect(self, first):
        batch, seen = [first], {first[0]}
        deadline = time.monotonic() + self._wait
        stop, held = False, []
        while True:
            with self._count_lock:
                if len(batch) >= self._open:      # every open stream has reported: nothing more can arrive for this tick
                    break
            left = deadline - time.monotonic()
            try:
                nxt = self._q.get(timeout=left) if left > 0 else self._q.get_nowait()
            except queue.Empty:
                break
            if nxt is None:
                stop = True
                break
            if nxt[0] in seen:                    # a second chunk of one stream (a caller that does not wait): next tick
                held.append(nxt)
                break
            seen.add(nxt[0])
            batch.append(nxt)
        return batch, held, stop

    def _serve(self, batch) -> None:
        import torch
        live = [(sid, c, fut) for sid, c, fut in batch if fut.set_running_or_notify_cancel()]
        if not live:
            return
        self.ticks += 1
        groups = {}
        for item in live:
            groups.setdefault(int(item[1].shape[1]), []).append(item)
        for n, items in sorted(groups.items()):
            try:
                with self._lock:
                    codes = torch.stack([c for _sid, c, _f in items])
                    outs = self._pool.push([sid for sid, _c, _f in items], codes)
                    self.pushes += 1
                    flat = torch.cat(outs)
                    if self._to_pcm16 is not None:
                        flat = self._to_pcm16(flat)
                    host = flat.cpu().numpy()
                o = 0
                for (_sid, _c, fut), piece in zip(items, outs):
                    k = int(piece.numel())
                    fut.set_result(host[o:o + k].copy())
                    o += k
                self.chunks += len(items)
            except Exception as e:                # a failed push fails the chunks that were in it, nobody else's
                for _sid, _c, fut in items:
                    if not fut.done():
                        fut.set_exception(e)

    def _run(self) -> None:
        carry = []
        while True:
            first = carry.pop(0) if carry else self._q.get()
            if first is None:
                break
            batch, held, stop = self._collect(first)
            carry += held
            self._serve(batch)
            if stop:
                break
        for item in carry:
            item[2].set_exception(RuntimeError("StreamBatcher is closed"))
        while True:
            try:
                rest = self._q.get_nowait()
            except queue.Empty:
                break
            if rest is not None:
                rest[2].set_exception(RuntimeError("StreamBatcher is closed"))
