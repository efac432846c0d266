"""Where a tile's time goes: phase timestamps (s_memtime) of every 64th workgroup of one contraction launch.
Phases: 0 start, 1 first tiles landed, 2 main loop done, 3 h tile written (fused unit), 4 second contraction done,
5 epilogue done.  Runs single layers of the flagship decoder through the operator entry points."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qwen3_tts_amd import _lib
from qwen3_tts_amd import decoder as D

lib = _lib.load()
SLOTS = 4096
TUNING = int(os.environ.get("Q3C_TUNING", "0"))
buf = torch.zeros((SLOTS, 8), dtype=torch.int64, device="cuda")
g = torch.Generator().manual_seed(0)


def report(name, fused):
    torch.cuda.synchronize()
    t = buf.cpu()
    t = t[(t[:, 0] != 0) & (t[:, 5] != 0)]
    d = (t[:, 1:6] - t[:, 0:5]).double()
    tot = (t[:, 5] - t[:, 0]).double()
    names = ["prologue (first tiles)", "main loop", "inner activation -> h tile", "second contraction", "epilogue"]
    wall = (t[:, 7] - t[:, 6]).double()               # 100 MHz ticks
    mhz = (tot.sum() / wall.sum() * 100.0).item() if wall.sum() > 0 else float("nan")
    print(f"{name}: {t.shape[0]} sampled tiles, {tot.mean():.0f} cycles per tile (p10 {tot.quantile(0.1):.0f}, p90 {tot.quantile(0.9):.0f}), "
          f"shader clock {mhz:.0f} MHz")
    for i, n in enumerate(names):
        if not fused and i in (2, 3):
            continue
        print(f"    {n:28s} {d[:, i].mean():8.0f} cycles  {100 * d[:, i].mean() / tot.mean():5.1f} %")
    buf.zero_()


def unit(Cn, dil, B, L):
    w = {"ru.act1.alpha": torch.randn(Cn, generator=g) * 0.3, "ru.act1.beta": torch.randn(Cn, generator=g) * 0.3,
         "ru.conv1.conv.weight": torch.randn(Cn, Cn, 7, generator=g) * (7 * Cn) ** -0.5, "ru.conv1.conv.bias": torch.zeros(Cn),
         "ru.act2.alpha": torch.randn(Cn, generator=g) * 0.3, "ru.act2.beta": torch.randn(Cn, generator=g) * 0.3,
         "ru.conv2.conv.weight": torch.randn(Cn, Cn, 1, generator=g) * Cn ** -0.5, "ru.conv2.conv.bias": torch.zeros(Cn)}
    x = torch.randn(B, L, Cn, device="cuda")
    D.residual_unit(x, w, "ru.", dil, precision="bf16")          # warm-up (allocations, code load)
    lib.q3c_debug_set_operator_context(TUNING, C.c_void_p(buf.data_ptr()), SLOTS)
    D.residual_unit(x, w, "ru.", dil, precision="bf16")
    lib.q3c_debug_set_operator_context(TUNING, None, 0)
    report(f"residual unit, {Cn} channels, dilation {dil}, {B} x {L} rows", True)


def conv(cin, cout, k, dil, B, L):
    w = torch.randn(cout, cin, k, generator=g) * (k * cin) ** -0.5
    x = torch.randn(B, L, cin, device="cuda")
    D.causal_conv1d(x, w, torch.zeros(cout), dil, "bf16")
    lib.q3c_debug_set_operator_context(TUNING, C.c_void_p(buf.data_ptr()), SLOTS)
    D.causal_conv1d(x, w, torch.zeros(cout), dil, "bf16")
    lib.q3c_debug_set_operator_context(TUNING, None, 0)
    report(f"conv k{k} d{dil} {cin}->{cout}, {B} x {L} rows", False)


def tconv(cin, cout, stride, B, L):
    w = torch.randn(cin, cout, 2 * stride, generator=g) * (2 * cin) ** -0.5
    x = torch.randn(B, L, cin, device="cuda")
    D.causal_trans_conv1d(x, w, torch.zeros(cout), stride, "bf16")
    lib.q3c_debug_set_operator_context(TUNING, C.c_void_p(buf.data_ptr()), SLOTS)
    D.causal_trans_conv1d(x, w, torch.zeros(cout), stride, "bf16")
    lib.q3c_debug_set_operator_context(TUNING, None, 0)
    report(f"transposed conv k{2 * stride} s{stride} {cin}->{cout} (single output), {B} x {L} rows", False)


unit(96, 1, 16, 575445)
unit(96, 9, 16, 575445)
unit(192, 3, 16, 191815)
conv(384, 384, 7, 3, 16, 47955)
conv(384, 384, 1, 1, 16, 47955)
tconv(384, 192, 4, 16, 47955)
tconv(192, 96, 3, 16, 191816)
