"""Per-dispatch list of one counter from a rocprofv3 --pmc rocpd database, in dispatch order.
usage: python tools/pmc_list.py <db> <counter> [kernel-name substring]"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
key = "dispatch_id" if "dispatch_id" in cols else "id"
sub = sys.argv[3] if len(sys.argv) > 3 else ""
rows = c.execute(f"select {key}, kernel_name, sum(value) from counters_collection where counter_name = ? group by {key}, kernel_name order by {key}",
                 (sys.argv[2],)).fetchall()
for d, n, v in rows:
    if sub in n:
        scale = 2048.0 if sys.argv[2] == "FETCH_SIZE" else 1024.0 if sys.argv[2] == "WRITE_SIZE" else 1.0
        print(d, n[:70], f"{v * scale / 1e6:.1f}" + (" MB" if scale > 1 else ""))
