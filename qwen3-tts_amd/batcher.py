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
        for c in items:
            if int(c.shape[1]) < 1:
                raise ValueError("empty code sequence")
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
