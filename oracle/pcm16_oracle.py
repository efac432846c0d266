"""TEST INFRASTRUCTURE, not product code: CPU restatement of the float -> 16-bit PCM conversion the reference applies
to every decoded waveform, `soundfile.write(buffer, wav, sample_rate, format="WAV")` (reference `server.py:108-113,
117-121`; soundfile's default subtype for WAV is PCM_16).

PARITY UNPINNED: `soundfile` (and libsndfile under it) is not installed in this container and the reference holds no
audio fixture, so the rule below is libsndfile's published one (src/pcm.c, `f2s_array` with float normalisation on:
`lrintf(src * 0x7FFF)`, round half to even, no clipping) restated, not checked against the library.  The decode clamps
to [-1, 1], where clipping cannot occur.  The WAV container layout IS pinned: `wav_file_bytes` must equal what
`scipy.io.wavfile.write` (in the container) produces for the same int16 samples (tests/test_oracle_golden.py)."""
import struct

import numpy as np


def float_to_pcm16(x: np.ndarray) -> np.ndarray:
    """lrintf(x * 32767) in fp32, round half to even; values outside [-1, 1] saturate (libsndfile would wrap)."""
    v = np.asarray(x, dtype=np.float32) * np.float32(32767.0)
    return np.clip(np.rint(v), -32768, 32767).astype(np.int16)


def wav_file_bytes(pcm: np.ndarray, sample_rate: int) -> bytes:
    """Canonical 44-byte RIFF/WAVE header (PCM, mono, 16 bit) + little-endian samples."""
    pcm = np.ascontiguousarray(pcm, dtype="<i2")
    data = pcm.tobytes()
    return (b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVE" + b"fmt " +
            struct.pack("<IHHIIHH", 16, 1, 1, sample_rate, sample_rate * 2, 2, 16) + b"data" + struct.pack("<I", len(data)) + data)
