"""Aggregate a rocprofv3 --pmc counter_collection.csv by kernel: sum of each counter over dispatches."""
import csv
import glob
import re
import sys
from collections import defaultdict

root = sys.argv[1]
files = glob.glob(root + "/**/*counter_collection.csv", recursive=True)
agg = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(int)
for f in files:
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row["Kernel_Name"]
            m = re.search(r"convgemm_kernelI(\w+?)EEv", k)
            name = m.group(1) if m else re.sub(r"\(.*", "", k)[:60]
            agg[name][row["Counter_Name"]] += float(row["Counter_Value"])
            cnt[(name, row["Counter_Name"])] += 1
names = sorted(agg, key=lambda n: -agg[n].get("SQ_WAVE_CYCLES", agg[n].get("GRBM_GUI_ACTIVE", 0)))
for n in names[:14]:
    c = agg[n]
    disp = max(cnt[(n, k)] for k in c)
    print(f"{n}  (dispatches {disp})")
    wc = c.get("SQ_WAVE_CYCLES", 0)
    for k in sorted(c):
        extra = f"  ({100 * c[k] / wc:.1f}% of WAVE_CYCLES)" if wc and k.startswith("SQ_") and k != "SQ_WAVE_CYCLES" else ""
        print(f"    {k:32s} {c[k]:.4g}{extra}")
