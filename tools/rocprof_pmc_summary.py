#!/usr/bin/env python
"""Summarises rocprofv3 --pmc captures (rocpd sqlite) per kernel:  python tools/rocprof_pmc_summary.py OUT.md OUT.json DB [DB ...]
FETCH_SIZE / WRITE_SIZE are reported in KiB by rocprofv3; MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE
shows exactly half of the bytes of a 16-B/lane coalesced stream, other widths and WRITE_SIZE are uncalibrated."""
import json
import sqlite3
import sys


def main():
    out_md, out_json, dbs = sys.argv[1], sys.argv[2], sys.argv[3:]
    rows = {}
    for db in dbs:
        c = sqlite3.connect(db)
        for name, counter, n, avg, lo, hi in c.execute(
                "select kernel_name, counter_name, count(*), avg(value), min(value), max(value) from counters_collection "
                "group by kernel_name, counter_name"):
            if "dr::" not in name:
                continue
            short = name.split("(")[0].replace("void ", "")
            rows.setdefault(short, {})[counter] = dict(dispatches=n, avg=avg, min=lo, max=hi)
    with open(out_md, "w") as f:
        f.write("# rocprofv3 --pmc summary (separate passes per counter, --kernel-trace only)\n\n")
        f.write("| kernel | counter | dispatches | avg (KiB) | min | max | avg bytes |\n|---|---|---|---|---|---|---|\n")
        for k, v in rows.items():
            for cn, d in v.items():
                f.write(f"| `{k}` | {cn} | {d['dispatches']} | {d['avg']:.1f} | {d['min']:.1f} | {d['max']:.1f} | {d['avg'] * 1024:.4g} |\n")
    json.dump(rows, open(out_json, "w"), indent=1)
    print("wrote", out_md, out_json)


if __name__ == "__main__":
    main()
