"""CPU-only checks of the host side: the C ABI library loads and exports what include/q3c.h declares, the config
struct matches the header, host-side input handling, and the multi-rank sharding/gather path over gloo."""
import ctypes
import os
import re
import subprocess
import sys
import tempfile

import numpy as np
import pytest
import torch

import qwen3_tts_amd as q3
from qwen3_tts_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "q3c.h")


def _declared():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(q3c_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()                                   # raises if libq3c.so is missing: no fallback exists
    declared = _declared()
    assert declared, "no declarations parsed from include/q3c.h"
    for name in declared:
        assert hasattr(lib, name), f"libq3c.so lacks {name}"
    assert sorted(_lib.EXPORTS) == declared
    assert lib.q3c_version().decode().startswith("q3c")


def test_config_struct_matches_header():
    src = '#include <stdio.h>\n#include "q3c.h"\nint main(void){printf("%zu %zu\\n", sizeof(q3c_config), sizeof(q3c_profile_entry));return 0;}\n'
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "t.c")
        open(c, "w").write(src)
        exe = os.path.join(d, "t")
        subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), c, "-o", exe], check=True)
        a, b = map(int, subprocess.run([exe], check=True, capture_output=True, text=True).stdout.split())
    assert a == ctypes.sizeof(_lib.Q3cConfig) and b == ctypes.sizeof(_lib.Q3cProfileEntry)


def test_create_rejects_bad_config_without_gpu():
    lib = _lib.load()
    cfg = _lib.make_config(q3.tiny_config(), "bf16", 512)
    cfg.num_key_value_heads = 3                           # 4 heads over 3 kv heads
    ctx = ctypes.c_void_p()
    rc = lib.q3c_create(ctypes.byref(cfg), ctypes.byref(ctx))
    assert rc == -1 and b"head" in lib.q3c_last_error(ctx)
    lib.q3c_destroy(ctx)
    with pytest.raises(ValueError):
        _lib.make_config(q3.tiny_config(), "fp8", 512)


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_decoder_fails_loudly_without_gpu():
    from qwen3_tts_amd.decoder import Decoder
    cfg = q3.tiny_config()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        Decoder(cfg, q3.synthetic_weights(cfg, 0))


def test_input_normalisation():
    from qwen3_tts_amd.decoder import Decoder
    d = object.__new__(Decoder)
    d.cfg = q3.tiny_config()                              # Q = 4
    Q = d.cfg.num_quantizers
    tq = torch.arange(7 * Q).reshape(7, Q)
    for given in (tq, tq.t(), {"audio_codes": tq}, [tq], [{"audio_codes": tq.numpy()}], tq[None], tq.t()[None]):
        items = d._normalize_items(given)
        assert len(items) == 1 and items[0].shape == (Q, 7) and torch.equal(items[0], tq.t())
    assert len(d._normalize_items(torch.zeros(3, 5, Q, dtype=torch.long))) == 3
    with pytest.raises(ValueError):
        d._normalize_items(torch.zeros(5, Q + 1, dtype=torch.long))
    with pytest.raises(ValueError):
        d._normalize_items({"codes": tq})


def test_config_derived_sizes_and_chunk_plan():
    c = q3.flagship_config()
    assert (c.total_upsample, c.trim, c.output_length(125)) == (1920, 555, 239445)
    assert c.chunk_plan(375) == [(0, 300, 0), (275, 375, 25)]
    assert c.chunk_plan(25, 10, 25) == [(0, 10, 0), (0, 20, 10), (0, 25, 20)]      # context = start while start - 25 <= 0
    assert c.chunked_output_length(375) == 718890 and c.chunked_output_length(7500) == 14386125
    m = c.macs_per_frame()
    assert abs(m["total"] - 2560.1e6) < 1e6 and abs(sum(v for k, v in m.items() if k != "total") - m["total"]) < 1
    with pytest.raises(ValueError):
        q3.DecoderConfig(frontend="mean_embedding")       # needs io_proj=False


def test_lpt_and_chunk_assignment():
    from qwen3_tts_amd.parallel import chunk_assign, lpt_assign
    lengths = [375, 25, 100, 300, 301, 50, 200, 10]
    shards = lpt_assign(lengths, 3)
    assert sorted(i for s in shards for i in s) == list(range(len(lengths)))
    loads = [sum(lengths[i] for i in s) for s in shards]
    assert max(loads) - min(loads) <= max(lengths)
    assert lpt_assign(lengths, 3) == shards               # deterministic
    assert chunk_assign(25, 8)[0] == [0, 8, 16, 24] and sum(len(x) for x in chunk_assign(25, 8)) == 25


# ---- two ranks over gloo -------------------------------------------------------------------------------
_WORKER = r'''
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "oracle"))
import qwen3_tts_amd as q3
from qwen3_tts_amd.parallel import decode_sharded, gather_waveforms
import decoder_oracle as oracle

rank = int(os.environ["RANK"])
dist.init_process_group("gloo", rank=rank, world_size=2)
cfg = q3.tiny_config()
w = q3.synthetic_weights(cfg, 4)

class CpuStandIn:            # the sharding logic only needs .cfg, .decode and .device
    def __init__(self): self.cfg, self.device = cfg, torch.device("cpu")
    def decode(self, items): return oracle.decode(cfg, w, [torch.as_tensor(c) for c in items])

items = [q3.random_codes(cfg, 1, T, seed=T)[0].t().contiguous() for T in (3, 9, 2, 6, 5)]
want, _ = oracle.decode(cfg, w, items)
out = decode_sharded(CpuStandIn(), items, dst=0)
if rank == 0:
    wavs, sr = out
    assert sr == 24000 and len(wavs) == len(items)
    for a, b in zip(wavs, want):
        assert a.dtype == np.float32 and np.array_equal(a, b)
else:
    assert out is None
everyone = decode_sharded(CpuStandIn(), items, dst=None)
for a, b in zip(everyone[0], want):
    assert np.array_equal(a, b)
# a rank with nothing to send
local = [torch.arange(5, dtype=torch.float32)] if rank == 1 else []
got = gather_waveforms(local, [0] if rank == 1 else [], 1, dst=0)
if rank == 0:
    assert np.array_equal(got[0], np.arange(5, dtype=np.float32))
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
'''


def test_sharded_decode_two_ranks_gloo():
    with tempfile.TemporaryDirectory() as d:
        script = os.path.join(d, "worker.py")
        open(script, "w").write(_WORKER.format(root=ROOT))
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29571", WORLD_SIZE="2", OMP_NUM_THREADS="2")
        procs = [subprocess.Popen([sys.executable, script], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                                  stderr=subprocess.STDOUT, text=True) for r in range(2)]
        outs = [p.communicate(timeout=240)[0] for p in procs]
        for r, (p, o) in enumerate(zip(procs, outs)):
            assert p.returncode == 0, f"rank {r} failed:\n{o[-3000:]}"
