import re, sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end, stream_id from kernels order by start").fetchall()
marks = [i for i, r in enumerate(rows) if "lm_step_coop" in r[0]]
lo, hi = marks[-6], marks[-4]
t0 = rows[lo][1]
for n, s, e, st in rows[lo:hi]:
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    m = re.search(r"([a-z0-9_]+_kernel|copyBuffer|fillBuffer\w*)", n)
    print(f"{(s - t0) / 1e3:8.1f} .. {(e - t0) / 1e3:8.1f} us  stream {st}  {(m.group(1) if m else n[:50])}")
