"""HBM traffic per launch, split by (kernel, grid, LDS size) so that layers sharing one tile shape are told apart.
Input: two rocprofv3 --pmc directories (FETCH_SIZE pass, WRITE_SIZE pass) of the same command; units and the
FETCH_SIZE doubling as in pmc_traffic.py.  Usage: python tools/pmc_by_layer.py <fetch_dir> <write_dir>"""
import csv
import glob
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
                name = m.group(1) if m else re.sub(r"\(.*", "", k)[:60]
                key = (name, row.get("Grid_Size", "?"), row.get("LDS_Block_Size", "?"))
                tot[key] += float(row["Counter_Value"])
                n[key] += 1
    return tot, n


fetch, nf = collect(sys.argv[1], "FETCH_SIZE")
write, nw = collect(sys.argv[2], "WRITE_SIZE")
rows = []
for k in set(fetch) | set(write):
    f = 2.0 * fetch.get(k, 0.0) * 1024 / max(nf.get(k, 1), 1)
    w = write.get(k, 0.0) * 1024 / max(nw.get(k, 1), 1)
    rows.append((f + w, f, w, max(nf.get(k, 0), nw.get(k, 0)), k))
print(f"{'fetch MB':>10} {'write MB':>10} {'launches':>8}  kernel, grid, lds")
for t, f, w, n, k in sorted(rows, reverse=True)[:40]:
    print(f"{f / 1e6:10.1f} {w / 1e6:10.1f} {n:8d}  {k[0]}  grid {k[1]}  lds {k[2]}")
