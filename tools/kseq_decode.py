import re, sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end from kernels order by start").fetchall()
marks = [i for i, r in enumerate(rows) if "collapse_kernel" in r[0]]
lo, hi = marks[-2] + 1, marks[-1] + 1
prev_end = rows[lo - 1][2]
for n, s, e in rows[lo:hi]:
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    m = re.search(r"([a-z0-9_]+_kernel|copyBuffer|fillBuffer\w*)", n)
    short = (m.group(1) if m else n[:50])
    print(f"{(s - prev_end) / 1e3:7.1f} gap {(e - s) / 1e3:7.1f} us  {short}")
    prev_end = max(prev_end, e)
