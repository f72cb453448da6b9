import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = list(c.execute("select name, start, end from kernels order by start"))
idx = [i for i, r in enumerate(rows) if 'ransac_init' in r[0]]
i0, i1 = idx[-2], idx[-1]
t0 = rows[i0][1]
for n, s, e in rows[i0:i1]:
    print(f"{(s - t0) / 1e3:8.1f} -> {(e - t0) / 1e3:8.1f}  ({(e - s) / 1e3:6.1f})  {n[:60]}")
