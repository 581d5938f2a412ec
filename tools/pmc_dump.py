"""Mean counter values per kernel family from a rocprofv3 --pmc rocpd database: python tools/pmc_dump.py <db> [substring]"""
import sqlite3
import sys
from collections import defaultdict

c = sqlite3.connect(sys.argv[1])
want = sys.argv[2] if len(sys.argv) > 2 else ""
acc = defaultdict(lambda: [0, 0.0])
for name, ctr, val in c.execute("select kernel_name, counter_name, value from counters_collection"):
    if want in name:
        k = (name.split("(")[0][-60:], ctr)
        acc[k][0] += 1
        acc[k][1] += val
for (name, ctr), (n, v) in sorted(acc.items()):
    print(f"{name:60s} {ctr:32s} n={n:3d} mean={v / n:.4g}")
