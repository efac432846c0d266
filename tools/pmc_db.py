"""Per-kernel counter sums of a rocprofv3 --pmc run stored as a rocpd SQLite database, as a share of SQ_WAVE_CYCLES.
Usage: python tools/pmc_db.py <results.db> [rows]"""
import re
import sqlite3
import sys
from collections import defaultdict

cur = sqlite3.connect(sys.argv[1]).cursor()
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 4
agg = defaultdict(lambda: defaultdict(float))
seen = defaultdict(set)
dur = defaultdict(float)
for name, did, cn, cv, d in cur.execute("select name, dispatch_id, counter_name, counter_value, duration from pmc_events"):
    m = re.search(r"convgemm_kernelI(\w+?)EEv", name)
    k = m.group(1) if m else name[:48]
    agg[k][cn] += cv
    if did not in seen[k]:
        seen[k].add(did)
        dur[k] += d
for k, c in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:rows]:
    wc = c.get("SQ_WAVE_CYCLES", 0)
    print(f"{k}: {len(seen[k])} dispatches, {dur[k] / 1e6:.2f} ms")
    for cn, v in sorted(c.items()):
        print(f"   {cn:26s} {v:12.4g}" + (f" {100 * v / wc:6.1f}% of wave cycles" if wc else ""))
