"""Per-kernel averages of every counter in a rocprofv3 --pmc rocpd database (value summed over the dispatch's dimensions).
usage: python tools/pmc_counters.py <db> [kernel-name substring]"""
import sqlite3
import sys
from collections import defaultdict

c = sqlite3.connect(sys.argv[1])
sub = sys.argv[2] if len(sys.argv) > 2 else ""
cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
key = "dispatch_id" if "dispatch_id" in cols else "id"
acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
for d, n, cn, v in c.execute(f"select {key}, kernel_name, counter_name, sum(value) from counters_collection group by {key}, kernel_name, counter_name"):
    if sub in n:
        a = acc[n.replace("(anonymous namespace)::", "").split("(")[0][-70:]][cn]
        a[0] += 1
        a[1] += v
for k, cs in acc.items():
    print(k)
    for cn, (n, v) in sorted(cs.items()):
        print(f"    {cn:32s} n={n:4d} avg {v / n:16.1f}")
