"""AddressSanitizer + UBSan pass over the HOST side of libq3c (csrc/decoder.cpp, csrc/encoder_impl.h), on a machine without a GPU.

    python tools/sanitize_host.py

Builds csrc/decoder.cpp once more with -fsanitize=address,undefined on the host pass only and -DQ3C_HOST_DRYRUN (device memory
becomes host memory, nothing is launched), links it with the regular objects of the other sources into tools/bin/sanitize_host,
and runs that driver (tools/sanitize_host.cpp) on seeded weights of three architectures: the small split-RVQ decoder, the small
mean-embedding decoder with grouped KV heads, and the small decoder with the encode side configured.  The driver walks
configuration checks, weight staging, q3c_finalize's repacking, the workspace / stream-state layouts and the error paths that
need no launch; the sanitizers abort on the first out-of-bounds access, leak or undefined operation."""
import ctypes as C
import os
import shutil
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import importlib  # noqa: E402

import qwen3_tts_amd as q3  # noqa: E402
from qwen3_tts_amd import _lib  # noqa: E402

PKG = os.path.join(ROOT, "qwen3-tts_amd")
BIN = os.path.join(ROOT, "tools", "bin")


def build() -> str:
    sys.path.insert(0, PKG)
    import build as q3build                      # regular objects of the other translation units
    q3build.build(verbose=False)
    os.makedirs(BIN, exist_ok=True)
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    san = ["-Xarch_host", "-fsanitize=address", "-Xarch_host", "-fsanitize=undefined", "-Xarch_host", "-fno-sanitize-recover=undefined",
           "-Xarch_host", "-fno-omit-frame-pointer"]
    common = ["--offload-arch=gfx950", "-O1", "-g", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function"]
    obj = os.path.join(BIN, "decoder_dryrun.o")
    drv = os.path.join(BIN, "sanitize_host.o")
    subprocess.run([hipcc] + common + ["-x", "hip", "-DQ3C_HOST_DRYRUN"] + san + ["-c", os.path.join(PKG, "csrc", "decoder.cpp"), "-o", obj], check=True)
    subprocess.run([hipcc] + common + ["-x", "hip"] + san + ["-c", os.path.join(ROOT, "tools", "sanitize_host.cpp"), "-o", drv], check=True)
    others = [os.path.join(PKG, "build", s + ".o") for s in q3build.SOURCES if s != "decoder.cpp"]
    exe = os.path.join(BIN, "sanitize_host")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-fsanitize=address", "-fsanitize=undefined", "-fno-gpu-sanitize", "-o", exe, drv, obj] + others, check=True)
    return exe


def write_blob(path, weights):
    with open(path, "wb") as f:
        f.write(struct.pack("<i", len(weights)))
        for name, t in weights.items():
            t = t.detach().float().contiguous()
            nb = name.encode()
            f.write(struct.pack("<i", len(nb)) + nb + struct.pack("<i", t.dim()) + struct.pack(f"<{t.dim()}q", *t.shape))
            f.write(t.numpy().tobytes())


def main() -> int:
    from qwen3_tts_amd.decoder import rope_inv_freq
    E = importlib.import_module("qwen3-tts_amd.encoder")
    exe = build()
    import torch
    cases = []
    cfg = q3.tiny_config()
    cases.append(("split_rvq bf16", cfg, "bf16", None))
    cases.append(("mean_embedding f32, grouped kv", q3.tiny_config("mean_embedding", num_key_value_heads=2), "f32", None))
    cases.append(("split_rvq f32 + encoder", cfg, "f32", E.EncoderConfig(num_filters=16, hidden_size=32, num_attention_heads=2, head_dim=16,
                                                                          intermediate_size=64, num_hidden_layers=2, sliding_window=20)))
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:protect_shadow_gap=0", UBSAN_OPTIONS="print_stacktrace=1")
    rc = 0
    with tempfile.TemporaryDirectory() as d:
        for k, (label, cfg, precision, ecfg) in enumerate(cases):
            w = dict(q3.synthetic_weights(cfg, seed=k))
            w["rope.inv_freq"] = rope_inv_freq(cfg)
            args = [os.path.join(d, f"cfg{k}.bin"), os.path.join(d, f"w{k}.bin")]
            open(args[0], "wb").write(bytes(_lib.make_config(cfg, precision, 8192)))
            if ecfg is not None:
                w.update(E.synthetic_encoder_weights(ecfg, cfg.codebook_dim, seed=3))
                hd = ecfg.head_dim
                w["encoder_rope.inv_freq"] = 1.0 / (ecfg.rope_theta ** (torch.arange(0, hd, 2, dtype=torch.float) / hd))
                args.append(os.path.join(d, f"e{k}.bin"))
                open(args[2], "wb").write(bytes(_lib.make_encoder_config(ecfg, 16384)))
            write_blob(args[1], w)
            r = subprocess.run([exe] + args, env=env, capture_output=True, text=True)
            print(f"[{label}] exit {r.returncode}: {r.stdout.strip()}")
            if r.returncode:
                print(r.stderr[-6000:])
                rc = 1
    return rc


if __name__ == "__main__":
    sys.exit(main())
