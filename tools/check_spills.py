"""Register report of the contraction kernels in the built objects: VGPRs, scratch spills, SGPR spills per kernel.
A scratch spill inside a DMA loop costs far more than its own latency: the compiler waits for a reload with vmcnt(0),
which drains every tile in flight.  (It cannot break the kernels' explicit `s_waitcnt vmcnt(N)` counts: an extra
vector-memory operation only makes such a wait stricter.)  Exit status 1 if any kernel spills to scratch.
Usage: python tools/check_spills.py [object files...]   (default: qwen3-tts_amd/build/*.hip.o)"""
import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def kernels(obj):
    with tempfile.TemporaryDirectory() as d:
        tmp = os.path.join(d, os.path.basename(obj))
        shutil.copy(obj, tmp)
        subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", tmp], check=True, capture_output=True)      # extracts the code objects beside it
        cos = glob.glob(tmp + ".*gfx950*")
        if not cos:
            return
        out = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", cos[0]], capture_output=True, text=True, check=True).stdout
    cur = {}
    for line in out.splitlines():
        m = re.match(r"\s+-?\s*\.(\w+):\s+(.*)", line)
        if not m:
            continue
        k, v = m.groups()
        if k == "name":
            cur["name"] = re.sub(r".*convgemm_kernelI(\w+?)EEv.*", r"convgemm<\1>", v)
        elif k in ("vgpr_count", "sgpr_spill_count"):
            cur[k] = int(v)
        elif k == "vgpr_spill_count":
            cur[k] = int(v)
            if "name" in cur:
                yield dict(cur)
            cur = {}


if __name__ == "__main__":
    objs = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "qwen3-tts_amd", "build", "*.hip.o")))
    bad = 0
    for o in objs:
        for k in kernels(o):
            flag = ""
            if k["vgpr_spill_count"]:
                flag = "   <-- SPILLS TO SCRATCH"
                bad += 1
            print(f"{os.path.basename(o):28s} {k['name'][:64]:64s} vgpr {k.get('vgpr_count', 0):4d} scratch spills {k['vgpr_spill_count']:5d} "
                  f"sgpr spills {k.get('sgpr_spill_count', 0):3d}{flag}")
    sys.exit(1 if bad else 0)
