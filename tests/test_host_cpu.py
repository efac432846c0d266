"""CPU-only checks of the host side: the C ABI library loads and exports what include/q3c.h declares, the config
struct matches the header, host-side input handling, and the multi-rank sharding/gather path over gloo."""
import ctypes
import json
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
    src = '#include <stdio.h>\n#include "q3c.h"\nint main(void){printf("%zu %zu %zu\\n", sizeof(q3c_config), sizeof(q3c_profile_entry), sizeof(q3c_encoder_config));return 0;}\n'
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "t.c")
        open(c, "w").write(src)
        exe = os.path.join(d, "t")
        subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), c, "-o", exe], check=True)
        a, b, e = map(int, subprocess.run([exe], check=True, capture_output=True, text=True).stdout.split())
    assert a == ctypes.sizeof(_lib.Q3cConfig) and b == ctypes.sizeof(_lib.Q3cProfileEntry) and e == ctypes.sizeof(_lib.Q3cEncoderConfig)


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


def test_encoder_configuration_without_gpu():
    """q3c_encoder_configure validates on the host; frame counts follow the chain of rounded-up strides of the conv encoder
    (every strided conv pads its input on the right to a whole stride) and agree between Python, the C ABI and the oracle's
    convolutions."""
    import importlib
    import sys
    E = importlib.import_module("qwen3-tts_amd.encoder")
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import encoder_oracle as eo
    lib = _lib.load()
    ecfg = E.EncoderConfig()
    assert ecfg.hop == 1920
    ctx = ctypes.c_void_p()
    cfg = _lib.make_config(q3.tiny_config(), "f32", 512)
    assert lib.q3c_create(ctypes.byref(cfg), ctypes.byref(ctx)) == 0
    ec = _lib.make_encoder_config(ecfg, 4096)
    assert lib.q3c_encoder_configure(ctx, ctypes.byref(ec)) == 0
    for samples in (1, 1919, 1920, 1921, 24000, 31337, 720000):
        n = ctypes.c_int64()
        assert lib.q3c_encode_frames(ctx, samples, ctypes.byref(n)) == 0
        assert n.value == ecfg.frames(samples) >= -(-samples // ecfg.hop)
    bad = _lib.make_encoder_config(ecfg, 4096)
    bad.sliding_window = 400
    assert lib.q3c_encoder_configure(ctx, ctypes.byref(bad)) == -7 and b"sliding_window" in lib.q3c_last_error(ctx)
    bad = _lib.make_encoder_config(ecfg, 4096)
    bad.hidden_size = 520
    assert lib.q3c_encoder_configure(ctx, ctypes.byref(bad)) == -1
    lib.q3c_destroy(ctx)
    # the oracle's convolutions give the same lengths at the small sizes
    t = E.tiny_encoder_config()
    w = E.synthetic_encoder_weights(t, 8, seed=0)
    for samples in (1, 5, 6, 7, 101):
        x = eo.seanet_encoder(t, w, torch.zeros(1, 1, samples))
        y = eo.causal_conv1d(x, w["downsample.conv.weight"], None, stride=t.downsample_stride, pad_mode="replicate")
        assert y.shape[-1] == t.frames(samples), (samples, y.shape)


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
from qwen3_tts_amd.parallel import decode_long_sharded, decode_sharded, gather_waveforms
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

class DeviceStandIn(CpuStandIn):     # the form the GPU decoder offers: samples stay in one flat tensor until the gather
    def decode_device(self, items):
        wavs, _ = oracle.decode(cfg, w, [torch.as_tensor(c) for c in items])
        lens = [len(x) for x in wavs]
        starts = [int(v) for v in np.cumsum([0] + lens[:-1])]
        return torch.cat([torch.from_numpy(x) for x in wavs]), starts, lens
    skip_context = True
    skips = []
    def forward(self, codes, skip_frames=0):
        # q3c_forward_skip: the samples of the first skip_frames frames are not produced (zeros here); the rest are the forward's
        DeviceStandIn.skips.append(skip_frames)
        wav = oracle.forward(cfg, w, codes)
        wav[..., : skip_frames * cfg.total_upsample] = 0.0
        return wav
out = decode_sharded(DeviceStandIn(), items, dst=None)
for a, b in zip(out[0], want):
    assert np.array_equal(a, b)
# one long utterance: its chunks are sharded round-robin and gathered in order
long_codes = q3.random_codes(cfg, 1, 23, seed=7)
want_long = oracle.chunked_decode(cfg, w, long_codes, 5, 2)[0, 0].numpy()
got_long = decode_long_sharded(DeviceStandIn(), long_codes[0], 5, 2, dst=None)
assert got_long.shape == want_long.shape == (cfg.chunked_output_length(23, 5, 2),)
assert np.abs(got_long - want_long).max() <= 1e-4      # fp32 batching noise of the CPU GEMMs is ~2e-5
# every chunk after the first was launched with its context as skip_frames (the launches a single GPU's chunked_decode makes)
assert sorted(set(DeviceStandIn.skips)) == ([0, 2] if rank == 0 else [2]), DeviceStandIn.skips

# streams over two ranks: placed on the least-loaded rank when they open, never migrating, no collective on the data path
from qwen3_tts_amd.parallel import ShardedStreams, StreamRouter
class CpuPool:                       # a StreamPool stand-in: a push = the un-chunked forward of all frames so far, new samples only
    def __init__(self, capacity): self.capacity, self.hist, self.done = capacity, {{}}, {{}}
    def open(self):
        sid = min(set(range(self.capacity)) - set(self.hist)); self.hist[sid] = None; self.done[sid] = 0; return sid
    def close_stream(self, sid): del self.hist[sid], self.done[sid]
    def push(self, ids, codes):
        out = []
        for b, sid in enumerate(ids):
            self.hist[sid] = codes[b:b + 1] if self.hist[sid] is None else torch.cat([self.hist[sid], codes[b:b + 1]], dim=2)
            wav = oracle.forward(cfg, w, self.hist[sid])[0, 0]
            out.append(wav[self.done[sid]:].clone()); self.done[sid] = wav.numel()
        return out
pool = CpuPool(3)
ss = ShardedStreams(pool)
gids = [ss.open() for _ in range(5)]
assert [ss.router.rank_of(g) for g in gids] == [0, 1, 0, 1, 0] and len(pool.hist) == (3 if rank == 0 else 2)
sc = q3.random_codes(cfg, 5, 6, seed=33)
got = [[] for _ in gids]
for t in range(0, 6, 2):
    pieces = ss.push_gather(gids, sc[:, :, t:t + 2], dst=None)
    for k, p_ in enumerate(pieces):
        got[k].append(p_)
for k in range(5):
    want_k = oracle.forward(cfg, w, sc[k:k + 1])[0, 0].numpy()
    assert np.abs(np.concatenate(got[k]) - want_k).max() <= 1e-4, k
ss.close(gids[1])                    # rank 1 now holds one stream, rank 0 three: the next two streams open on rank 1
g5, g6 = ss.open(), ss.open()
assert ss.router.rank_of(g5) == 1 and ss.router.rank_of(g6) == 1 and ss.router.load == [3, 3]
try:
    ss.open(); raise SystemExit("a full node must refuse a stream")
except RuntimeError as e:
    assert "full" in str(e)
only1 = ss.push_gather([g5], q3.random_codes(cfg, 1, 2, seed=5), dst=0)      # a tick whose streams all live on one rank
assert (only1 is not None and len(only1) == 1 and only1[0].shape == (2 * cfg.total_upsample - cfg.trim,)) if rank == 0 else only1 is None
r = StreamRouter(4, 2)
assert [r.rank_of(r.open()) for _ in range(8)] == [0, 1, 2, 3, 0, 1, 2, 3]
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


# ---- install hook (stub model: the real package is not installed here) ------------------------------------------
class _StubDecoderModule:
    def __init__(self, cfg, weights, prefix=""):
        import types
        # an upstream-style config: no `frontend` / `io_proj` / `rvq_out_dim` fields, and `codebook_dim` meaning the RVQ
        # in/out width (twice the code-book row width), which must not be mistaken for the row width
        d = {k: v for k, v in cfg.to_dict().items() if k not in ("frontend", "io_proj", "rvq_out_dim")}
        d["codebook_dim"] = 2 * cfg.codebook_dim
        self.config = types.SimpleNamespace(**d)
        self._sd = {prefix + k: v for k, v in weights.items()}

    def state_dict(self):
        return self._sd


class _StubTokenizer:
    def __init__(self, decoder):
        self.decoder = decoder

    def decode(self, encoded):
        return "reference", 24000


class _StubInner(torch.nn.Module):
    def __init__(self, tok):
        super().__init__()
        self.talker = torch.nn.Linear(2, 2)
        self.speech_tokenizer = tok


class _StubTTS:
    def __init__(self, tok):
        self.model = _StubInner(tok)
        self.name = "stub"


class _FakeDecoder:
    def __init__(self, cfg, weights, device, precision, **kw):
        self.cfg, self.weights, self.device, self.precision, self.kw = cfg, weights, device, precision, kw

    def decode(self, encoded):
        return "mi355x", self.cfg.sample_rate


@pytest.mark.parametrize("frontend,prefix", [("split_rvq", ""), ("mean_embedding", "model.code2wav.")])
def test_install_hook_finds_tokenizer_and_derives_config(frontend, prefix):
    from qwen3_tts_amd import install
    cfg = q3.tiny_config(frontend)
    w = q3.synthetic_weights(cfg, seed=0)
    extra = dict(w)
    extra["unrelated.buffer"] = torch.zeros(3)
    tok = _StubTokenizer(_StubDecoderModule(cfg, extra, prefix))
    model = _StubTTS(tok)
    new = install.install_decoder(model, precision="f32", device="cuda:0", decoder_factory=_FakeDecoder, max_positions=99)
    import dataclasses
    want = cfg if frontend == "split_rvq" else dataclasses.replace(     # the RVQ widths do not exist for a mean-embedding front end
        cfg, codebook_dim=new.cfg.codebook_dim, rvq_out_dim=new.cfg.rvq_out_dim)
    assert new.cfg == want and new.precision == "f32" and new.kw == {"max_positions": 99}
    assert set(new.weights) == set(w) and all(torch.equal(new.weights[k], w[k]) for k in w)
    assert model.model.speech_tokenizer.decode([0]) == ("mi355x", 24000)
    assert model.model.speech_tokenizer.reference_decode([0]) == ("reference", 24000)
    install.uninstall_decoder(model)
    assert model.model.speech_tokenizer.decode([0]) == ("reference", 24000)


def test_install_hook_refuses_mismatches():
    from qwen3_tts_amd import install
    cfg = q3.tiny_config("split_rvq")
    w = q3.synthetic_weights(cfg, seed=0)
    with pytest.raises(LookupError):
        install.find_speech_tokenizer(_StubTTS(object()))
    bad = dict(w)
    bad["decoder.1.block.2.conv1.conv.weight"] = torch.zeros(3, 3, 7)
    with pytest.raises(ValueError, match="missing or mis-shaped"):
        install.install_decoder(_StubTTS(_StubTokenizer(_StubDecoderModule(cfg, bad))), device="cuda:0", decoder_factory=_FakeDecoder)
    mod = _StubDecoderModule(cfg, w)
    mod.config.head_dim = cfg.head_dim * 2                        # not a shape by itself, but q_proj's rows contradict it
    with pytest.raises(ValueError, match="head_dim"):
        install.install_decoder(_StubTTS(_StubTokenizer(mod)), device="cuda:0", decoder_factory=_FakeDecoder)
    mod = _StubDecoderModule(cfg, w)
    mod.config.decoder_dim = cfg.decoder_dim * 2                  # shape-determined sizes come from the tensors alone
    assert install.install_decoder(_StubTTS(_StubTokenizer(mod)), device="cuda:0", decoder_factory=_FakeDecoder).cfg == cfg
    renamed = {k.replace("quantizer.", "rvq."): v for k, v in w.items()}
    with pytest.raises(KeyError, match="rename"):
        install.install_decoder(_StubTTS(_StubTokenizer(_StubDecoderModule(cfg, renamed))), device="cuda:0", decoder_factory=_FakeDecoder)
    got = install.install_decoder(_StubTTS(_StubTokenizer(_StubDecoderModule(cfg, renamed))), device="cuda:0",
                                  decoder_factory=_FakeDecoder, rename={"rvq.": "quantizer."})
    assert got.cfg == cfg and set(got.weights) == set(w)
    cpu_model = _StubTTS(_StubTokenizer(_StubDecoderModule(cfg, w)))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        install.install_decoder(cpu_model, decoder_factory=_FakeDecoder)


# ---- model manager and local checkpoints (reference models.py:15-132, config.py:12-16,23-72) ---------------------------
def test_local_checkpoint_round_trip_and_refusals(tmp_path):
    """save_local / load_local: config.json + model.safetensors in a sub-directory of a model directory, a key prefix, a
    rename map, an upstream-style config that cannot be trusted for shape-determined sizes; hub names and missing paths are
    refused before anything is read."""
    from qwen3_tts_amd import checkpoint
    cfg = q3.tiny_config("split_rvq", sliding_window=40, rope_theta=5000.0)
    w = q3.synthetic_weights(cfg, seed=6)
    extra = dict(w)
    extra["encoder.unrelated"] = torch.zeros(4)
    root = tmp_path / "Qwen3-TTS-12Hz-0.6B-Base"
    checkpoint.save_local(str(root / "speech_tokenizer"), cfg, extra, prefix="decoder.")
    got_cfg, got_w = checkpoint.load_local(str(root))
    assert got_cfg == cfg                                          # sliding_window / rope_theta come from config.json, sizes from shapes
    assert set(got_w) == set(w) and all(torch.equal(got_w[k], w[k]) for k in w)
    one_file = checkpoint.load_local(str(root / "speech_tokenizer" / "model.safetensors"))
    assert one_file[0] == cfg
    # key names of another convention: refused until a rename map says how they translate
    other = tmp_path / "renamed"
    checkpoint.save_local(str(other), cfg, {k.replace("quantizer.", "rvq."): v for k, v in w.items()}, prefix="tok.decoder.")
    with pytest.raises(KeyError, match="rename"):
        checkpoint.load_local(str(other))
    assert checkpoint.load_local(str(other), rename={"rvq.": "quantizer."})[0] == cfg
    # a tensor of the wrong shape
    bad = dict(w)
    bad["decoder.1.block.2.conv1.conv.weight"] = torch.zeros(3, 3, 7)
    checkpoint.save_local(str(tmp_path / "bad"), cfg, bad)
    with pytest.raises(ValueError, match="missing or mis-shaped"):
        checkpoint.load_local(str(tmp_path / "bad"))
    for name in ("Qwen/Qwen3-TTS-12Hz-0.6B-Base", str(tmp_path / "absent")):
        with pytest.raises(FileNotFoundError, match="local director"):
            checkpoint.load_local(name)
    (tmp_path / "empty").mkdir()
    with pytest.raises(FileNotFoundError, match="safetensors"):
        checkpoint.load_local(str(tmp_path / "empty"))
    # bf16 storage, as a released checkpoint has it: values come back as the bf16-rounded fp32
    checkpoint.save_local(str(tmp_path / "bf16"), cfg, w, dtype=torch.bfloat16)
    _, w16 = checkpoint.load_local(str(tmp_path / "bf16"))
    k = "decoder.0.conv.weight"
    assert w16[k].dtype == torch.float32 and torch.equal(w16[k], w[k].to(torch.bfloat16).float())


def test_model_manager_follows_the_reference_and_loads_local_paths_only(tmp_path):
    """Cache by model type, dtype map, attention string, TTS_* flags, aliases, default model, warm-up - with a stub for
    Qwen3TTSModel.from_pretrained - and the one deliberate difference: a model path that is not a local directory raises
    instead of being handed to the factory as a hub name."""
    from qwen3_tts_amd import models
    cfg = q3.tiny_config("split_rvq")
    w = q3.synthetic_weights(cfg, seed=0)
    calls = []

    class Model(_StubTTS):
        def __init__(self):
            super().__init__(_StubTokenizer(_StubDecoderModule(cfg, w)))
            self.generated = []

        def generate_voice_clone(self, **kw):
            self.generated.append(("voice_clone", kw))

        def generate_custom_voice(self, **kw):
            self.generated.append(("custom_voice", kw))

    def factory(path, device_map, dtype, attn_implementation):
        calls.append((path, device_map, dtype, attn_implementation))
        return Model()

    class Warm(_FakeDecoder):
        warmed = 0

        def warmup(self, batch, frames):
            Warm.warmed += 1

    def installer(model, precision, device, **kw):
        from qwen3_tts_amd import install
        return install.install_decoder(model, precision=precision, device=device, decoder_factory=Warm, **kw)

    base = tmp_path / "models" / "Qwen3-TTS-12Hz-0.6B-Base"
    base.mkdir(parents=True)
    custom = tmp_path / "elsewhere"
    custom.mkdir()
    sample = tmp_path / "sample.mp3"
    sample.write_bytes(b"x")
    env = {"TTS_DEVICE": "cuda:3", "TTS_DTYPE": "float32", "TTS_USE_FLASH_ATTENTION": "true", "TTS_USE_WARMUP": "true",
           "MODEL_1_7B_CUSTOM_VOICE": str(custom), "TTS_WARMUP_AUDIO": str(sample)}
    mm = models.TTSModelManager(factory=factory, installer=installer, env=env, models_root=str(tmp_path / "models"),
                                decoder_kwargs={"max_positions": 77})
    assert (mm.device, mm.dtype, mm.attn_impl, mm.default_model) == ("cuda:3", torch.float32, "flash_attention_2", "base_0.6b")
    assert not mm.is_loaded("base_0.6b") and mm.get_loaded_models() == []
    mm.load_default_models()                                       # TTS_DEFAULT_MODEL unset: base_0.6b (config.py:72)
    assert calls == [(str(base), "cuda:3", torch.float32, "flash_attention_2")]
    m = mm.get_model("base_0.6b")
    assert mm.load_model("base_0.6b") is m and len(calls) == 1     # cached: the factory is not called again
    dec = mm.get_decoder("base_0.6b")
    assert (dec.precision, dec.device, dec.kw) == ("f32", "cuda:3", {"max_positions": 77}) and dec.cfg == cfg
    assert m.model.speech_tokenizer.decode([0]) == ("mi355x", 24000)        # the decode was swapped in
    assert Warm.warmed == 1 and m.generated[0][0] == "voice_clone" and m.generated[0][1]["ref_audio"] == str(sample)
    assert m.generated[0][1]["x_vector_only_mode"] is True and m.generated[0][1]["max_new_tokens"] == 256
    # an environment override, reached through the alias too; custom-voice models warm up with a speaker
    m2 = mm.get_model("custom_voice")
    assert calls[-1][0] == str(custom) and m2.generated[0] == ("custom_voice", dict(text=models.WARMUP_TEXT, language="Korean",
                                                                                     speaker="Sohee", max_new_tokens=256))
    assert mm.get_loaded_models() == ["base_0.6b", "custom_voice"]
    # unknown type: the reference's message; a directory that does not exist: refused, the factory never sees a hub name
    with pytest.raises(ValueError, match="Unknown model type: nope"):
        mm.load_model("nope")
    n = len(calls)
    with pytest.raises(FileNotFoundError, match="never resolves hub names"):
        mm.load_model("base_1.7b")
    hub = models.TTSModelManager(factory=factory, installer=installer, env={"MODEL_0_6B_BASE": "Qwen/Qwen3-TTS-12Hz-0.6B-Base"},
                                 models_root=str(tmp_path / "models"))
    with pytest.raises(FileNotFoundError, match="never resolves hub names"):
        hub.load_default_models()
    assert len(calls) == n
    # defaults of the reference: bfloat16 -> the bf16 decode, sdpa, no warm-up; an unknown dtype name falls back to bfloat16
    d = models.TTSModelManager(factory=factory, installer=installer, env={"TTS_DTYPE": "float8"}, models_root=str(tmp_path / "models"))
    assert (d.device, d.dtype, d.attn_impl, d.use_warmup) == ("cuda:0", torch.bfloat16, "sdpa", False)
    d.load_model("base_0.6b")
    assert d.get_decoder("base_0.6b").precision == "bf16" and Warm.warmed == 2       # no warm-up without TTS_USE_WARMUP
    # float16 models decode in fp32 (no fp16 kernel path; never fewer mantissa bits than the model asked for)
    h = models.TTSModelManager(factory=factory, installer=installer, env={"TTS_DTYPE": "float16"}, models_root=str(tmp_path / "models"))
    assert h.get_decoder("base_0.6b").precision == "f32"
    # without the qwen-tts package and without an injected factory the load says what is missing
    with pytest.raises(ImportError, match="qwen-tts"):
        models.TTSModelManager(env={}, models_root=str(tmp_path / "models")).load_model("base_0.6b")


# ---- torch custom ops: registration and shape functions (no GPU: the kernels themselves are CUDA-only) -------------
def test_torch_ops_are_registered_with_shape_functions():
    from torch._subclasses.fake_tensor import FakeTensorMode
    from qwen3_tts_amd import torch_ops
    import types
    cfg = q3.tiny_config("split_rvq")
    h = torch_ops.register_decoder(types.SimpleNamespace(cfg=cfg))
    for name in ("decode_forward", "chunked_decode", "rvq_dequant", "causal_conv1d", "causal_trans_conv1d", "snake_beta"):
        assert hasattr(torch.ops.q3tts, name)
    with FakeTensorMode():
        codes = torch.zeros((2, cfg.num_quantizers, 310), dtype=torch.int64)
        assert torch.ops.q3tts.decode_forward(codes, h).shape == (2, 1, cfg.output_length(310))
        assert torch.ops.q3tts.chunked_decode(codes, h, 300, 25).shape == (2, 1, cfg.chunked_output_length(310))
        assert torch.ops.q3tts.rvq_dequant(codes, h).shape == (2, 310, 2, cfg.codebook_dim)
        x = torch.zeros((2, 50, 16))
        assert torch.ops.q3tts.causal_conv1d(x, torch.zeros((24, 16, 7)), None, 3, "f32").shape == (2, 50, 24)
        assert torch.ops.q3tts.causal_trans_conv1d(x, torch.zeros((16, 8, 10)), None, 5, "f32").shape == (2, 245, 8)
        assert torch.ops.q3tts.snake_beta(x, torch.zeros(16), torch.zeros(16), 1e-9).shape == x.shape
    with pytest.raises(NotImplementedError):                      # real CPU tensors: there is no CPU kernel
        torch.ops.q3tts.rvq_dequant(torch.zeros((1, cfg.num_quantizers, 3), dtype=torch.int64), h)
    torch_ops.unregister_decoder(h)


# ---- cross-request batching (host logic only: a stand-in decoder) ----------------------------------------
class _CountingDecoder:
    """`Decoder.decode`'s contract without a GPU: waveform i = per-frame sum of utterance i's codes."""

    def __init__(self, cfg, expose_cfg=True):
        from qwen3_tts_amd.decoder import Decoder
        self._cfg = cfg
        import types                                 # without the code-book size the batcher cannot range-check at submit()
        self.cfg = cfg if expose_cfg else types.SimpleNamespace(num_quantizers=cfg.num_quantizers)
        self.calls = []
        self._norm = Decoder._normalize_items

    def _normalize_items(self, encoded):
        return self._norm(self, encoded)

    def decode(self, encoded):
        items = self._normalize_items(encoded)
        self.calls.append(len(items))
        for c in items:
            if int(c.max()) >= self._cfg.codebook_size:
                raise IndexError("code out of range")
        return [c.float().sum(0).numpy() for c in items], 24000


def test_batcher_groups_concurrent_requests_and_returns_each_its_own():
    import threading
    from qwen3_tts_amd.batcher import DecodeBatcher
    cfg = q3.tiny_config()
    dec = _CountingDecoder(cfg)
    reqs = [[q3.random_codes(cfg, 1, 3 + (7 * i + j) % 11, seed=100 * i + j)[0] for j in range(1 + i % 3)] for i in range(12)]
    results = [None] * len(reqs)
    with DecodeBatcher(dec, max_wait_ms=200.0) as bat:
        def work(i):
            results[i] = bat.decode(reqs[i])
        threads = [threading.Thread(target=work, args=(i,)) for i in range(len(reqs))]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert bat.requests == len(reqs) and bat.batches < len(reqs)          # requests shared launches
    assert sum(dec.calls) == sum(len(r) for r in reqs)
    for i, r in enumerate(reqs):
        wavs, sr = results[i]
        assert sr == 24000 and len(wavs) == len(r)
        for w, c in zip(wavs, r):
            assert np.array_equal(w, c.float().sum(0).numpy())


def test_batcher_errors_stay_with_their_request():
    from qwen3_tts_amd.batcher import DecodeBatcher
    cfg = q3.tiny_config()
    good = q3.random_codes(cfg, 1, 5, seed=1)[0]
    bad = good.clone()
    bad[0, 2] = cfg.codebook_size
    # codes on the host are range-checked in the caller's thread: the bad request never reaches a window
    dec = _CountingDecoder(cfg)
    with DecodeBatcher(dec, max_wait_ms=100.0) as bat:
        with pytest.raises(ValueError):                             # wrong code-book count: rejected in the caller's thread
            bat.submit(torch.zeros((cfg.num_quantizers + 1, cfg.num_quantizers + 5), dtype=torch.int64))
        with pytest.raises(IndexError):
            bat.submit(bad)
        neg = good.clone()
        neg[1, 0] = -1
        with pytest.raises(IndexError):
            bat.submit([good, neg])
        assert np.array_equal(bat.decode(good)[0][0], good.float().sum(0).numpy())
        assert dec.calls == [1]                                     # nothing was decoded twice
    # an error found only by the decode itself (as for codes that live on the device) stays with its request
    dec = _CountingDecoder(cfg, expose_cfg=False)
    with DecodeBatcher(dec, max_wait_ms=100.0) as bat:
        f1, f2, f3 = bat.submit(good), bat.submit(bad), bat.submit([good, good])
        assert np.array_equal(f1.result()[0][0], good.float().sum(0).numpy())
        with pytest.raises(IndexError):
            f2.result()
        assert len(f3.result()[0]) == 2
    with pytest.raises(RuntimeError):
        bat.submit(good)


class _FakePool:
    """What StreamBatcher needs from a StreamPool, on the CPU: a chunk's "samples" are its codes' column sums times the
    stream's running frame count, so a mix-up of streams or of chunk order shows."""

    def __init__(self, cfg, capacity):
        import types
        self.dec = types.SimpleNamespace(cfg=cfg, max_positions=50)
        self.capacity, self._frames, self.calls = capacity, {}, []

    def open(self):
        sid = min(set(range(self.capacity)) - set(self._frames))
        self._frames[sid] = 0
        return sid

    def close_stream(self, sid):
        del self._frames[sid]

    def frames_pushed(self, sid):
        return self._frames[sid]

    def push(self, ids, codes):
        self.calls.append((list(ids), tuple(codes.shape)))
        out = []
        for b, sid in enumerate(ids):
            out.append((codes[b].sum(0).float() + 1000.0 * self._frames[sid]))
            self._frames[sid] += codes.shape[2]
        return out


def test_stream_batcher_shares_pushes_between_concurrent_streams():
    import threading
    from qwen3_tts_amd.batcher import StreamBatcher
    cfg = q3.tiny_config()
    pool = _FakePool(cfg, 16)
    sb = StreamBatcher(pool, max_wait_ms=200.0)
    try:
        N, chunks = 12, 4
        sids = [sb.open() for _ in range(N)]
        codes = {s: q3.random_codes(cfg, chunks, 3, seed=s) for s in sids}     # [chunk, Q, 3] per stream
        got = {s: [] for s in sids}

        def run(s):
            for k in range(chunks):
                got[s].append(sb.advance(s, codes[s][k]))

        threads = [threading.Thread(target=run, args=(s,)) for s in sids]
        for t in threads:
            t.start()
        for t in threads:
            t.join(30)
        for s in sids:
            for k in range(chunks):
                want = (codes[s][k].sum(0).float() + 1000.0 * 3 * k).numpy()
                assert np.array_equal(got[s][k], want), (s, k)
        # every tick waited for all open streams (or 200 ms): 12 streams x 4 chunks in about 4 pushes, never one per chunk
        assert sb.chunks == N * chunks and sb.pushes <= 2 * chunks, (sb.pushes, pool.calls)
        assert all(len(set(ids)) == len(ids) for ids, _ in pool.calls)          # a stream never appears twice in a push
        # chunks of different length in one tick: one push per length
        before = sb.pushes
        f1, f2, f3 = sb.submit(sids[0], codes[sids[0]][0][:, :2]), sb.submit(sids[1], codes[sids[1]][0]), sb.submit(sids[2], codes[sids[2]][0][:, :2])
        for s in sids[3:]:
            sb.close_stream(s)
        assert f1.result(10).shape == (2,) and f2.result(10).shape == (3,) and f3.result(10).shape == (2,)
        assert sb.pushes - before <= 3
        # refusals happen in the caller's thread
        with pytest.raises(ValueError):
            sb.submit(sids[0], torch.zeros((cfg.num_quantizers, 40), dtype=torch.int64))        # past max_positions = 50
        with pytest.raises(IndexError):
            sb.submit(sids[0], torch.full((cfg.num_quantizers, 1), cfg.codebook_size, dtype=torch.int64))
        with pytest.raises(ValueError):
            sb.submit(sids[0], torch.zeros((cfg.num_quantizers + 1, 1), dtype=torch.int64))
    finally:
        sb.close()
    with pytest.raises(RuntimeError):
        sb.submit(0, torch.zeros((cfg.num_quantizers, 1), dtype=torch.int64))


def test_wav_container_and_base64_match_the_oracle():
    import base64
    import pcm16_oracle as po
    from qwen3_tts_amd import audio
    pcm = po.float_to_pcm16(np.cos(np.arange(777, dtype=np.float32) * 0.02))
    assert audio.wav_file_bytes(pcm, 24000) == po.wav_file_bytes(pcm, 24000)
    assert base64.b64decode(audio.audio_to_base64(pcm, 24000)) == po.wav_file_bytes(pcm, 24000)
    with pytest.raises(ValueError):
        audio.to_pcm16(torch.zeros(4))                                 # no host-side conversion path



def test_built_kernels_do_not_spill_to_scratch():
    """The contraction kernels keep LDS-DMA tiles and inline-asm residual loads in flight across compiler-visible code:
    a scratch spill (or a reload's vmcnt(0)) inside those regions would drain the queue or read a register before its
    load has landed.  The register report of the built objects must therefore show no scratch spill in any kernel."""
    import glob
    import importlib.util
    objs = sorted(glob.glob(os.path.join(ROOT, "qwen3-tts_amd", "build", "*.hip.o")))
    if not objs:
        pytest.skip("object files are not in the tree (library built elsewhere)")
    spec = importlib.util.spec_from_file_location("check_spills", os.path.join(ROOT, "tools", "check_spills.py"))
    cs = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cs)
    seen, bad = 0, []
    for o in objs:
        for k in cs.kernels(o):
            seen += 1
            if k["vgpr_spill_count"]:
                bad.append((os.path.basename(o), k["name"], k["vgpr_spill_count"]))
    assert seen > 40, f"only {seen} kernels found in {len(objs)} objects"
    assert not bad, f"kernels spilling to scratch: {bad}"


# ---- bench.py starts its own ranks ----------------------------------------------------------------------------------
def test_host_side_is_clean_under_address_and_ub_sanitizers():
    """tools/sanitize_host.py: csrc/decoder.cpp rebuilt with -fsanitize=address,undefined on the host pass and device memory
    replaced by host memory (nothing is launched), driven through configuration checks, weight staging, q3c_finalize's
    repacking, the workspace / stream-state layouts and the error paths, on three architectures.  Any out-of-bounds access,
    leak or undefined operation in the host code fails the run."""
    import shutil
    if not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        pytest.skip("hipcc not available")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "sanitize_host.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    assert r.stdout.count("0 failed expectation(s)") == 3, r.stdout


def test_bench_launcher_starts_one_rank_per_gpu_and_refuses_a_mislabelled_run():
    """`python bench.py --gpus N` without a launcher environment starts N ranks itself (before torch is imported); a run
    whose rank count differs from --gpus exits non-zero instead of printing a 1-GPU number under an N-GPU label."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    cmd = bench.spawn_command(["--gpus", "4", "--steps", "7"], {})
    assert cmd is not None and "--nproc-per-node=4" in cmd and cmd[-4:] == ["--gpus", "4", "--steps", "7"]
    assert "127.0.0.1" in cmd and "torch.distributed.run" in cmd
    assert bench.spawn_command(["--gpus", "1"], {}) is None
    assert bench.spawn_command(["--gpus", "4"], {"WORLD_SIZE": "4"}) is None          # already a rank
    with pytest.raises(SystemExit):
        bench.check_world(8, 1)
    bench.check_world(2, 2)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"], env=env, capture_output=True,
                       text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    ranks = sorted(json.loads(l)["rank"] for l in r.stdout.splitlines() if l.startswith("{"))
    assert ranks == [0, 1]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"], env=dict(env, WORLD_SIZE="1"),
                       capture_output=True, text=True, timeout=240)
    assert r.returncode != 0 and "refusing" in r.stderr


# ---- REST surface: request shaping and schemas (no GPU) -----------------------------------------------------------------
def test_server_schemas_and_request_shaping_follow_the_reference():
    """Field names, defaults and bounds of reference schemas.py:10-35; sentence splitting and the auto-split rule of
    reference server.py:74-91, 205-211; the SSE wire format of API_GUIDE.md:138-161."""
    import pydantic
    from qwen3_tts_amd import server
    p = server.GenerationParams()
    assert (p.max_new_tokens, p.temperature, p.top_k, p.top_p, p.repetition_penalty, p.do_sample) == (2048, 0.9, 50, 1.0, 1.05, True)
    for bad in ({"max_new_tokens": 0}, {"max_new_tokens": 4097}, {"temperature": 2.5}, {"top_k": 101}, {"top_p": 1.5}, {"repetition_penalty": 0.9}):
        with pytest.raises(pydantic.ValidationError):
            server.GenerationParams(**bad)
    r = server.VoiceCloneRequest(text="a", ref_audio="x.mp3", ref_text="t")
    assert r.language == "Auto" and r.x_vector_only_mode is True and r.split_sentences is None and r.seed is None
    assert server.VoiceCloneRequest(text=["a", "b"], ref_audio=["x", "y"], ref_text=["t", "u"]).text == ["a", "b"]
    with pytest.raises(pydantic.ValidationError):
        server.VoiceCloneRequest(text="a")
    assert server.get_generation_kwargs() == {"max_new_tokens": 2048, "temperature": 0.9, "top_k": 50, "top_p": 1.0,
                                              "repetition_penalty": 1.05, "do_sample": True}
    assert server.split_into_sentences("One. Two!  Three?? Four") == ["One", "Two", "Three", "Four"]
    assert server.split_into_sentences("no terminator") == ["no terminator"]
    assert server.split_into_sentences("v1.5 is out") == ["v1.5 is out"]              # a dot without white space does not split
    assert not server.should_split("short. text.", None) and server.should_split("a. b. c. d. e. f. g", None)
    assert server.should_split("x" * 501, None) and not server.should_split("x" * 501, False) and server.should_split("a", True)
    assert server.model_key_for("0.6b") == "base_0.6b" and server.model_key_for("1.7b") == "base_1.7b" and server.model_key_for("big") == "base"
    assert server.sse_event("meta", {"status": "generating", "text": "안녕"}) == 'event: meta\ndata: {"status": "generating", "text": "안녕"}\n\n'
    cfg = q3.tiny_config()
    src = server.hashed_code_source(cfg)
    a, b = src(text="hello", seed=1), src(text="hello", seed=1)
    assert torch.equal(a, b) and a.shape[0] == cfg.num_quantizers and not torch.equal(a, src(text="hello", seed=2))


def test_launch_ordered_samples_reach_their_places_on_the_host():
    """`to_host_by_launch`: a device tensor laid out launch by launch, pieces that belong at scattered host offsets.  On the
    CPU only the small-result path runs (no staging, no streams), which shares the placement logic with the staged path."""
    from qwen3_tts_amd.decoder import to_host_by_launch
    rng = np.random.default_rng(3)
    counts = rng.integers(1, 400, size=57)
    host_order = rng.permutation(len(counts))                      # piece k of the device layout is utterance piece host_order[k]
    host_starts = np.zeros(len(counts), dtype=np.int64)
    o = 0
    for k in np.argsort(host_order):                               # host layout: pieces in host order, back to back
        host_starts[k] = o
        o += counts[k]
    total = int(counts.sum())
    dev_starts = np.concatenate([[0], np.cumsum(counts)[:-1]]).astype(np.int64)
    want = rng.standard_normal(total).astype(np.float32)
    flat = torch.empty(total, dtype=torch.float32)
    for k in range(len(counts)):
        flat[dev_starts[k]: dev_starts[k] + counts[k]] = torch.from_numpy(want[host_starts[k]: host_starts[k] + counts[k]])
    got = to_host_by_launch(flat, total, [int(dev_starts[20]), total], [None, None], dev_starts, host_starts, counts.astype(np.int64))
    assert np.array_equal(got, want)
    assert to_host_by_launch(torch.empty(0), 0, [], [], np.zeros(0, np.int64), np.zeros(0, np.int64), np.zeros(0, np.int64)).shape == (0,)


def test_model_code_source_takes_the_codes_at_the_decode_boundary():
    """`install.ModelCodeSource`: the model's own generation runs with the tokenizer's decode replaced by a recorder; the codes
    come back as [Q, T], the generation gets silence of the right length, and the real decode is back in place afterwards -
    also when the generation raises."""
    from qwen3_tts_amd import install
    cfg = q3.tiny_config()
    tok = _StubTokenizer(_StubDecoderModule(cfg, q3.synthetic_weights(cfg, seed=0), ""))
    model = _StubTTS(tok)
    calls = []

    def generate_voice_clone(text, language="Auto", **kw):
        g = torch.Generator().manual_seed(len(text))
        codes = torch.randint(0, cfg.codebook_size, (len(text) + 3, cfg.num_quantizers), generator=g)      # [T, Q], as the wrapper takes them
        wavs, sr = model.model.speech_tokenizer.decode([{"audio_codes": codes}])
        calls.append((text, language, wavs[0].shape, sr, torch.initial_seed()))
        if text == "boom":
            raise RuntimeError("generation failed")
        return wavs, sr

    model.generate_voice_clone = generate_voice_clone
    real = tok.decode
    src = install.ModelCodeSource(model, cfg)
    codes = src(text="hello", seed=7, language="Korean")
    assert codes.shape == (cfg.num_quantizers, 8) and codes.dtype == torch.int64
    g = torch.Generator().manual_seed(5)
    assert torch.equal(codes, torch.randint(0, cfg.codebook_size, (8, cfg.num_quantizers), generator=g).t())
    assert calls[0][:2] == ("hello", "Korean") and calls[0][2] == (cfg.chunked_output_length(8),) and calls[0][3] == cfg.sample_rate
    assert calls[0][4] == 7                                        # the request's seed reached torch before the generation
    assert tok.decode == real
    with pytest.raises(RuntimeError, match="generation failed"):
        src(text="boom")
    assert tok.decode == real
    model.generate_voice_clone = lambda text, **kw: ([np.zeros(4, np.float32)], 24000)      # never reaches the tokenizer
    with pytest.raises(RuntimeError, match="without handing any codes"):
        src(text="x")


def test_server_launcher_resolves_through_the_alias_and_refuses_a_missing_checkpoint():
    """`python -m qwen3_tts_amd.server` (the reference's `python server.py`): runs the real module as __main__ through the
    import alias, and without a local model directory stops before anything is fetched."""
    import subprocess
    env = {**os.environ, "TTS_MODELS_DIR": "/nonexistent-models", "PYTHONPATH": ROOT}
    r = subprocess.run([sys.executable, "-m", "qwen3_tts_amd.server"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "FileNotFoundError" in r.stderr and "local directories only" in r.stderr


def test_model_code_source_precomputes_the_voice_clone_prompt_as_the_reference_does():
    """One `create_voice_clone_prompt` per (reference audio, transcript, mode), reused by later sentences and requests, passed
    as `voice_clone_prompt=` with `non_streaming_mode=True` (reference server.py:228-251); a failing prompt falls back to the
    direct ref_audio call."""
    import types
    from qwen3_tts_amd import install
    cfg = q3.tiny_config()
    tok = _StubTokenizer(_StubDecoderModule(cfg, q3.synthetic_weights(cfg, seed=0), ""))
    model = _StubTTS(tok)
    made, calls = [], []

    def create_voice_clone_prompt(ref_audio, ref_text, x_vector_only_mode):
        made.append((ref_audio, ref_text, x_vector_only_mode))
        if ref_audio == "broken.wav":
            raise OSError("cannot read")
        return [types.SimpleNamespace(ref_spk_embedding=torch.ones(4), x_vector_only_mode=x_vector_only_mode, icl_mode=False, ref_code=None)]

    def generate_voice_clone(text, **kw):
        calls.append(kw)
        return model.model.speech_tokenizer.decode([torch.zeros((5, cfg.num_quantizers), dtype=torch.int64)])

    model.create_voice_clone_prompt, model.generate_voice_clone = create_voice_clone_prompt, generate_voice_clone
    src = install.ModelCodeSource(model, cfg, prompt_cache=2)
    kw = dict(language="Korean", ref_audio="a.wav", ref_text="hi", x_vector_only_mode=True, max_new_tokens=64)
    for _ in range(3):
        assert src(text="one sentence", **kw).shape == (cfg.num_quantizers, 5)
    assert made == [("a.wav", "hi", True)]                         # computed once, shared by the three generations
    assert all("ref_audio" not in c and c["non_streaming_mode"] is True and c["voice_clone_prompt"][0].icl_mode is False
               and c["language"] == "Korean" and c["max_new_tokens"] == 64 for c in calls)
    src(text="x", **{**kw, "ref_audio": "broken.wav"})             # prompt fails: the direct call, as the reference falls back
    assert calls[-1]["ref_audio"] == "broken.wav" and "voice_clone_prompt" not in calls[-1]
    src(text="x", **{**kw, "ref_audio": "b.wav"})
    src(text="x", **{**kw, "ref_audio": "c.wav"})                  # cache of two: a.wav is the least recently used and goes
    src(text="x", **kw)
    assert [m[0] for m in made] == ["a.wav", "broken.wav", "b.wav", "c.wav", "a.wav"]
