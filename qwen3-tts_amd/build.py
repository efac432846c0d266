"""Build libq3c.so (the C-ABI library of gfx950 kernels) in-tree with hipcc.  No GPU is needed to build."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libq3c.so")
SOURCES = ["convgemm.hip", "convgemm_f32_f32.hip", "convgemm_bf16_bf16.hip", "convgemm_f32_bf16.hip", "convgemm_bf16_f32.hip",
           "kernels.hip", "decoder.cpp"]
# no -ffast-math: the RVQ dequantiser and the fp32 parity path rely on IEEE adds / divides
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-x", "hip", "-Wall", "-Wno-unused-function"]


def _newer(target: str, deps) -> bool:
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(d) <= t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "convgemm_impl.h"), os.path.join(CSRC, "encoder_impl.h"),
               os.path.join(HERE, "..", "include", "q3c.h")]
    jobs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src + ".o")
        if force or not _newer(o, [s] + headers):
            jobs.append([hipcc] + FLAGS + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    with ThreadPoolExecutor(max_workers=6) as ex:
        list(ex.map(run, jobs))
    objs = [os.path.join(objdir, s + ".o") for s in SOURCES]
    if jobs or force or not _newer(LIB, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
