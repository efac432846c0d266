"""MFMA utilisation per kernel from one rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE):
busy cycles summed over the chip's 1024 SIMDs / (active GPU cycles of the dispatch x 1024).  rocprofv3 reports
GRBM_GUI_ACTIVE summed over the 8 XCDs, so it is divided by 8 (check: the result equals MFMA instructions x 32 cycles
/ (kernel duration x clock x 1024), and the algorithmic-FLOP rate of bench.py over the 2.5 PFLOP/s peak plus tile padding).
Usage: python tools/pmc_mfma_util.py <dir>"""
import csv
import glob
import re
import sys
from collections import defaultdict

agg = defaultdict(lambda: defaultdict(float))
n = defaultdict(int)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row["Kernel_Name"]
            m = re.search(r"convgemm_kernelI(\w+?)EEv", k)
            name = ("convgemm<" + m.group(1) + ">") if m else re.sub(r"\(.*", "", k)[:60]
            agg[name][row["Counter_Name"]] += float(row["Counter_Value"])
            if row["Counter_Name"] == "GRBM_GUI_ACTIVE":
                n[name] += 1
print(f"{'kernel':70s} {'launches':>8s} {'GPU cycles':>12s} {'MFMA insts':>12s} {'MFMA busy / (cycles x 1024 SIMDs)':>34s}")
for name, c in sorted(agg.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0)):
    act = c.get("GRBM_GUI_ACTIVE", 0) / 8.0
    if act <= 0 or not c.get("SQ_INSTS_MFMA"):
        continue
    print(f"{name:70s} {n[name]:8d} {act:12.4g} {c.get('SQ_INSTS_MFMA', 0):12.4g} {c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (act * 1024):34.3f}")
