"""Random row counts for the twelve-wave tiles of the fused 96-channel unit: B x L >= 786 432 rows with L mod 768 anything
(partial last tiles), dilations 1 / 3 / 9; each output must equal, bit for bit, the four-wave tiles' (tuning bit 1 << 22) on the same input.
Usage: python tools/unit_shape_fuzz.py [cases] [seed]"""
import os
import random
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from qwen3_tts_amd import _lib  # noqa: E402
from qwen3_tts_amd import decoder as D  # noqa: E402

lib = _lib.load()
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
g = torch.Generator().manual_seed(7)
rnd = lambda *shape, s=1.0: torch.randn(*shape, generator=g) * s
C = 96
for i in range(cases):
    B = rng.choice([1, 2, 3, 5, 8])
    L = -(-786432 // B) + rng.randint(0, 3000)
    d = rng.choice([1, 3, 9])
    w = {"ru.act1.alpha": rnd(C, s=0.3), "ru.act1.beta": rnd(C, s=0.3), "ru.conv1.conv.weight": rnd(C, C, 7, s=(7 * C) ** -0.5),
         "ru.conv1.conv.bias": rnd(C, s=0.1), "ru.act2.alpha": rnd(C, s=0.3), "ru.act2.beta": rnd(C, s=0.3),
         "ru.conv2.conv.weight": rnd(C, C, 1, s=0.5 * C ** -0.5), "ru.conv2.conv.bias": rnd(C, s=0.1)}
    x = torch.randn(B, L, C, device="cuda")
    twelve = D.residual_unit(x, w, "ru.", d, precision="bf16")
    lib.q3c_debug_set_operator_context(1 << 22, None, 0)
    try:
        four = D.residual_unit(x, w, "ru.", d, precision="bf16")
    finally:
        lib.q3c_debug_set_operator_context(0, None, 0)
    same = bool(torch.equal(twelve, four)) and bool(torch.isfinite(twelve).all())
    print(f"case {i}: B {B}, L {L} (L mod 768 = {L % 768}), dilation {d}: {'identical' if same else 'DIFFERENT'}", flush=True)
    if not same:
        sys.exit(1)
    del x, twelve, four
print(f"{cases} cases identical")
