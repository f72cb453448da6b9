import sqlite3, sys
for db in sys.argv[1:]:
    c = sqlite3.connect(db)
    for name, counter, n, avg in c.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name"):
        if 'dr::' in name:
            print(f"{name.split('(')[0][-40:]:40s} {counter:22s} n={n} avg={avg:.4g}")
