"""Summarise a rocprofv3 rocpd database (kernel trace): per-kernel calls / total / average, per step.
usage: python tools/kstats.py <results.db> <steps> [out.csv]"""
import re, sqlite3, sys
db, steps = sys.argv[1], float(sys.argv[2])
c = sqlite3.connect(db)
rows = c.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3 "
                 "from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
lines = ["kernel,calls,total_us,avg_us,min_us,max_us,pct,us_per_step"]
for n, cnt, t, a, mn, mx in rows:
    n = re.sub(r"\(anonymous namespace\)::", "", n).replace(",", ";")
    lines.append(f"\"{n[:160]}\",{cnt},{t:.1f},{a:.2f},{mn:.2f},{mx:.2f},{100 * t / tot:.2f},{t / steps:.1f}")
if len(sys.argv) > 3:
    open(sys.argv[3], "w").write("\n".join(lines) + "\n")
print(f"total kernel time {tot / 1e3:.2f} ms over {steps:g} steps = {tot / steps / 1e3:.3f} ms/step")
for l in lines[1:40]:
    f = l.rsplit(",", 7)
    print(f"{float(f[7]):9.1f} us/step {float(f[1]) / steps:7.1f} calls/step avg {float(f[3]):8.1f}  {f[0][:100]}")
