"""CPU probe: how much a weight profile amplifies perturbations through the decode.  Runs the oracle in fp32 and in PyTorch bf16
on the same codes and prints the SNR of every tapped stage and of the waveform (python tools/conditioning_probe.py [tiny|flagship] T)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import decoder_oracle as oracle  # noqa: E402
import qwen3_tts_amd as q3  # noqa: E402


def snr(a, b):
    a = a.double(); b = b.double()
    return 10 * np.log10(float((a ** 2).sum()) / max(float(((a - b) ** 2).sum()), 1e-300))


def main():
    size = sys.argv[1] if len(sys.argv) > 1 else "tiny"
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 24
    cfg = q3.flagship_config() if size == "flagship" else q3.tiny_config()
    codes = q3.random_codes(cfg, 1, T, seed=0)
    for profile in ("chaotic", "conditioned"):
        w = q3.synthetic_weights(cfg, seed=0, profile=profile)
        t0 = time.time()
        taps32, taps16 = {}, {}
        ref = oracle.forward(cfg, w, codes, taps32)
        low = oracle.forward(cfg, w, codes, taps16, dtype=torch.bfloat16)
        line = ", ".join(f"{k} {snr(taps32[k], taps16[k]):.1f}" for k in taps32 if k != "rvq_sums")
        print(f"{size} {profile}: {line}, wav {snr(ref, low):.1f} dB, max|err| {float((ref - low).abs().max()):.3e}, "
              f"rms wav {float(ref.pow(2).mean().sqrt()):.3f}, clipped {float((ref.abs() >= 1).float().mean()):.3%} ({time.time() - t0:.1f} s)")


if __name__ == "__main__":
    main()
