"""First contact of a new tile shape of the fused 96-channel residual unit with the GPU, one careful step at a time: the unit
through the operator entry point with the tuning bit Q3C_CHECK_BIT set and clear (default 1 << 22: four-wave tiles instead of
twelve-wave ones) on the same input, compared with each other and with the oracle, the first dilation first; stops at the first
mismatch.  Usage: python tools/unit_shape_check.py [dilations...]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import decoder_oracle as oracle  # noqa: E402
from qwen3_tts_amd import _lib  # noqa: E402
from qwen3_tts_amd import decoder as D  # noqa: E402

lib = _lib.load()
BIT = int(os.environ.get("Q3C_CHECK_BIT", str(1 << 22)))
g = torch.Generator().manual_seed(5)
rnd = lambda *shape, s=1.0: torch.randn(*shape, generator=g) * s
C, B, L = 96, 4, 2 * 100352 + 37                 # >= 384 * 512 rows in all; a partial last tile per item
dils = [int(v) for v in sys.argv[1:]] or [1]
for d in dils:
    w = {"ru.act1.alpha": rnd(C, s=0.3), "ru.act1.beta": rnd(C, s=0.3), "ru.conv1.conv.weight": rnd(C, C, 7, s=(7 * C) ** -0.5),
         "ru.conv1.conv.bias": rnd(C, s=0.1), "ru.act2.alpha": rnd(C, s=0.3), "ru.act2.beta": rnd(C, s=0.3),
         "ru.conv2.conv.weight": rnd(C, C, 1, s=0.5 * C ** -0.5), "ru.conv2.conv.bias": rnd(C, s=0.1)}
    x = rnd(B, C, L)
    xd = x.permute(0, 2, 1).contiguous().cuda()
    four = D.residual_unit(xd, w, "ru.", d, precision="bf16")
    torch.cuda.synchronize()
    print(f"dilation {d}: default shape done", flush=True)
    lib.q3c_debug_set_operator_context(BIT, None, 0)
    try:
        six = D.residual_unit(xd, w, "ru.", d, precision="bf16")
        torch.cuda.synchronize()
    finally:
        lib.q3c_debug_set_operator_context(0, None, 0)
    print(f"dilation {d}: shape with the bit set done", flush=True)
    diff = float((six - four).abs().max())
    want = oracle.residual_unit(x, w, "ru.", d).permute(0, 2, 1)
    e4, e6 = float((four.cpu() - want).abs().max()), float((six.cpu() - want).abs().max())
    top = float(want.abs().max())
    print(f"dilation {d}: max|with bit - default| {diff:.3e}; vs oracle: default {e4:.3e}, with bit {e6:.3e} of max|y| {top:.2f}; finite {bool(torch.isfinite(six).all())}", flush=True)
    if not (bool(torch.isfinite(six).all()) and e6 <= 0.01 * max(top, 1.0) and diff <= 0.01 * max(top, 1.0)):
        print("MISMATCH - stopping", flush=True)
        sys.exit(1)
print("shapes agree")
