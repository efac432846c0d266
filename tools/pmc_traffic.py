"""HBM traffic per launch and kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs of the
same command).  Units and corrections as for gfx950: both counters are in KiB; FETCH_SIZE reports half the bytes of a
wide coalesced read stream (requests tallied at 64 B for 128-B fetches), so it is doubled; WRITE_SIZE is exact.
Usage: python tools/pmc_traffic.py <fetch_dir> <write_dir> <out.json>"""
import csv
import glob
import json
import re
import sys
from collections import defaultdict


def collect(root, counter):
    tot, n = defaultdict(float), defaultdict(int)
    for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] != counter:
                    continue
                k = row["Kernel_Name"]
                m = re.search(r"convgemm_kernelI(\w+?)EEv", k)
                name = m.group(1) if m else re.sub(r"\(.*", "", k)[:80]
                tot[name] += float(row["Counter_Value"])
                n[name] += 1
    return tot, n


fetch, nf = collect(sys.argv[1], "FETCH_SIZE")
write, nw = collect(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(set(fetch) | set(write)):
    launches = max(nf.get(k, 0), nw.get(k, 0))
    if not launches:
        continue
    out[k] = {"launches": launches,
              "fetch_bytes_per_launch": 2.0 * fetch.get(k, 0.0) * 1024 / max(nf.get(k, 1), 1),
              "write_bytes_per_launch": write.get(k, 0.0) * 1024 / max(nw.get(k, 1), 1)}
    out[k]["hbm_bytes_per_launch"] = out[k]["fetch_bytes_per_launch"] + out[k]["write_bytes_per_launch"]
json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes over `python bench.py --steps 1 --warmup 1 "
                     "--no-cpu-baseline`; FETCH_SIZE doubled (gfx950), KiB -> bytes", "kernels": out}, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1)[:1500])
