"""Kernel sequence of ONE optimizer step from a rocprofv3 rocpd kernel trace: the dispatches between the last two adam_kernel
launches, in start order, with the idle gap before each (start - latest end so far).  Also prints the summed gap and the largest gaps.
usage: python tools/kseq.py <results.db> [out.txt]"""
import re, sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end from kernels order by start").fetchall()
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r[0]]
lo, hi = adam[-2] + 1, adam[-1] + 1
out, gaps, busy_end = [], [], rows[lo - 1][2]
for n, s, e in rows[lo:hi]:
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    m = re.search(r"([a-z0-9_]+_kernel|copyBuffer|fillBuffer\w*)", n)
    short = (m.group(1) if m else n[:40]) + ("" if "Li" not in n else " " + "".join(re.findall(r"Li(\d+)E", n)[:4]))
    gap = (s - busy_end) / 1e3
    gaps.append((gap, short))
    out.append(f"{gap:8.1f} gap {(e - s) / 1e3:8.1f} us  {short}")
    busy_end = max(busy_end, e)
span = (rows[hi - 1][2] - rows[lo - 1][2]) / 1e3
ker = sum(e - s for _, s, e in rows[lo:hi]) / 1e3
pos = sum(g for g, _ in gaps if g > 0)
head = [f"step span {span:.1f} us, kernels {ker:.1f} us, idle gaps {pos:.1f} us over {hi - lo} dispatches"]
by = {}
for g, n in gaps:
    if g > 0: by[n] = by.get(n, 0.0) + g
head += ["idle before (summed by kernel): " + ", ".join(f"{n} {g:.0f}" for n, g in sorted(by.items(), key=lambda x: -x[1])[:14])]
text = "\n".join(head + out) + "\n"
if len(sys.argv) > 2: open(sys.argv[2], "w").write(text)
print("\n".join(head))
