"""Per-dispatch view of a rocprofv3 --pmc run: counters of the dispatches whose kernel name and grid size match.
Usage: python tools/pmc_dispatch.py <dir> <kernel substring> <grid size in work-items>"""
import csv
import glob
import sys
from collections import defaultdict

root, sub, grid = sys.argv[1], sys.argv[2], int(sys.argv[3])
agg = defaultdict(float)
n = defaultdict(int)
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if sub in row["Kernel_Name"] and int(row["Grid_Size"]) == grid:
                agg[row["Counter_Name"]] += float(row["Counter_Value"])
                n[row["Counter_Name"]] += 1
wc = agg.get("SQ_WAVE_CYCLES", 0)
for k in sorted(agg):
    print(f"{k:32s} {agg[k] / max(n[k], 1):14.4g} per dispatch ({n[k]} dispatches)" + (f"  {100 * agg[k] / wc:6.1f}% of WAVE_CYCLES" if wc else ""))
