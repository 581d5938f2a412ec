"""Per-dispatch list of one counter from a rocprofv3 --pmc rocpd database, in dispatch order; consecutive dispatches of the same
kernel with the same grid are folded into one line (count, mean).
usage: python tools/pmc_list.py <db> <counter> [kernel-name substring]"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
key = "dispatch_id" if "dispatch_id" in cols else "id"
grid = "grid_size" if "grid_size" in cols else ("grid_size_x" if "grid_size_x" in cols else "0")
sub = sys.argv[3] if len(sys.argv) > 3 else ""
rows = c.execute(f"select {key}, kernel_name, {grid}, sum(value) from counters_collection where counter_name = ? "
                 f"group by {key}, kernel_name order by {key}", (sys.argv[2],)).fetchall()
scale = 2048.0 if sys.argv[2] == "FETCH_SIZE" else 1024.0 if sys.argv[2] == "WRITE_SIZE" else 1.0
unit = " MB" if scale > 1 else ""
run = None
def flush():
    if run: print(f"{run[0]:6d} x{run[3]:3d} grid {run[2]:>8} {run[1][:90]}  {run[4] / run[3] * scale / 1e6:.1f}{unit}")
for d, n, g, v in rows:
    if sub not in n: continue
    if run and run[1] == n and run[2] == g:
        run[3] += 1; run[4] += v
    else:
        flush(); run = [d, n, g, 1, v]
flush()
