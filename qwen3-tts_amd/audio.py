"""The output format of the reference's server: every waveform leaves it as a PCM_16 WAV file, base64-encoded inside
JSON / SSE events or as an `audio/wav` body (`server.py:108-140`, `API_GUIDE.md:138-161`).  The float -> 16-bit
conversion runs on the GPU (`q3c_wav_to_pcm16`: lrintf(x * 32767), libsndfile's rule), so half the bytes cross PCIe;
the 44-byte RIFF header and the base64 text are host work."""
from __future__ import annotations

import base64
import ctypes as C
import struct
from typing import List, Tuple

import numpy as np
import torch

from . import _lib


def to_pcm16(wav: torch.Tensor) -> torch.Tensor:
    """Device float32 samples (any shape) -> int16 tensor of the same shape, on the device."""
    if not wav.is_cuda:
        raise ValueError("to_pcm16 converts device tensors (the host side has no conversion path)")
    lib = _lib.load()
    x = wav.contiguous().float()
    y = torch.empty(x.shape, dtype=torch.int16, device=x.device)
    with torch.cuda.device(x.device):
        rc = lib.q3c_wav_to_pcm16(C.c_void_p(x.data_ptr()), x.numel(), C.c_void_p(y.data_ptr()),
                                  C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream))
    if rc:
        raise _lib.Q3cError(rc, "q3c_wav_to_pcm16 failed")
    return y


def decode_pcm16(decoder, encoded, **kw) -> Tuple[List[np.ndarray], int]:
    """`Decoder.decode` with 16-bit PCM results: -> (list of int16 1-D arrays, sample_rate)."""
    flat, starts, lengths = decoder.decode_device(encoded, **kw)
    host = decoder._to_host(to_pcm16(flat))
    return [host[s:s + n] for s, n in zip(starts, lengths)], decoder.cfg.sample_rate


def wav_file_bytes(pcm: np.ndarray, sample_rate: int) -> bytes:
    """A mono PCM_16 WAV file: canonical 44-byte RIFF header + little-endian samples."""
    pcm = np.ascontiguousarray(pcm, dtype="<i2")
    data = pcm.tobytes()
    return (b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVE" + b"fmt " +
            struct.pack("<IHHIIHH", 16, 1, 1, sample_rate, sample_rate * 2, 2, 16) + b"data" + struct.pack("<I", len(data)) + data)


def audio_to_base64(pcm: np.ndarray, sample_rate: int) -> str:
    """The `audio` field of the reference's JSON / SSE payloads (`server.py:108-113`) from int16 samples."""
    return base64.b64encode(wav_file_bytes(pcm, sample_rate)).decode("utf-8")
