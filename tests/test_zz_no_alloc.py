"""q3c_forward_static (the entry point that allocates nothing: its workspace is a static device array inside the library)
against q3c_forward through the C ABI: bit-identical output on two calls in a row, the second on a smaller input, in both
precisions; sizes beyond the static workspace are refused with a clear error; no device memory is allocated by the calls."""
import ctypes as C

import pytest
import torch

import qwen3_tts_amd as q3
from qwen3_tts_amd import _lib
from qwen3_tts_amd.decoder import Decoder

pytestmark = pytest.mark.gpu


def _static(dec, codes):
    lib = dec._lib
    B, _, T = codes.shape
    out = torch.full((B, 1, dec.cfg.output_length(T)), float("nan"), dtype=torch.float32, device=dec.device)
    torch.cuda.synchronize()
    before = torch.cuda.mem_get_info()[0]
    rc = lib.q3c_forward_static(dec._ctx, C.c_void_p(codes.data_ptr()), B, T, C.c_void_p(out.data_ptr()),
                                C.c_void_p(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    after = torch.cuda.mem_get_info()[0]
    return rc, out, before - after


@pytest.mark.parametrize("precision", ["f32", "bf16"])
def test_forward_static_equals_forward_twice_second_smaller(precision):
    cfg = q3.tiny_config()
    dec = Decoder(cfg, q3.synthetic_weights(cfg, seed=12), device="cuda:0", precision=precision)
    for B, T, seed in ((3, 40, 1), (2, 9, 2)):                 # the second call is smaller: it finds the first one's leftovers
        codes = q3.random_codes(cfg, B, T, seed=seed).cuda()
        want = dec.forward(codes)
        rc, got, grew = _static(dec, codes)
        assert rc == 0, dec._lib.q3c_last_error(dec._ctx).decode()
        assert grew <= 0, f"device memory in use grew by {grew} bytes across q3c_forward_static"
        assert got.shape == want.shape and not bool(torch.isnan(got).any())
        assert torch.equal(got, want)
        assert torch.equal(dec.forward_static(codes, torch.empty_like(want)), want)      # the Python mirror of the same call


def test_forward_static_flagship_and_limit():
    """Flagship widths inside the limit (bf16, 1 x 24 frames), then a size beyond it: refused, nothing written."""
    cfg = q3.flagship_config()
    dec = Decoder(cfg, q3.synthetic_weights(cfg, seed=0), device="cuda:0", precision="bf16")
    assert dec.workspace_bytes(1, 24) <= _lib.STATIC_SCRATCH_BYTES < dec.workspace_bytes(4, 300)
    codes = q3.random_codes(cfg, 1, 24, seed=5).cuda()
    want = dec.forward(codes)
    rc, got, grew = _static(dec, codes)
    assert rc == 0 and grew <= 0 and torch.equal(got, want)
    big = q3.random_codes(cfg, 4, 300, seed=6).cuda()
    rc, out, _ = _static(dec, big)
    assert rc == -4                                               # Q3C_ERR_WORKSPACE
    msg = dec._lib.q3c_last_error(dec._ctx).decode()
    assert "static" in msg and str(_lib.STATIC_SCRATCH_BYTES) in msg
    assert bool(torch.isnan(out).all())
    dec.profile(True)
    rc, _, _ = _static(dec, codes)
    assert rc == -3                                               # Q3C_ERR_STATE: profiling would create events
    dec.profile(False)
