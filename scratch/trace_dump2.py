import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = list(c.execute("select name, start, end from kernels order by start"))
idx = [i for i, r in enumerate(rows) if 'ransac_init' in r[0]]
print('periods between the last inits (us):', [round((rows[idx[-k]][1] - rows[idx[-k - 1]][1]) / 1e3, 1) for k in range(1, 8)])
i0, i1 = idx[-3], idx[-1]
t0 = rows[i0][1]
for n, s, e in rows[i0:i1 + 1]:
    print(f"{(s - t0) / 1e3:8.1f} -> {(e - t0) / 1e3:8.1f}  ({(e - s) / 1e3:6.1f})  {n[:50]}")
