#!/usr/bin/env python
"""Summarises rocprofv3 --pmc captures (rocpd sqlite) per kernel:
    python tools/rocprof_pmc_summary.py OUT.md OUT.json [--pairs P --points N --hyps B] DB [DB ...]
FETCH_SIZE / WRITE_SIZE are reported in KiB by rocprofv3; MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE
shows exactly half of the bytes of a 16-B/lane coalesced stream, other widths and WRITE_SIZE are uncalibrated."""
import hashlib
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the counters belong to the kernels as compiled from these files: bench.py re-hashes them and drops the traffic figure
# when they have changed since the capture
KERNEL_SOURCES = ["differentiable_ransac_amd/csrc/msac_score.hip", "differentiable_ransac_amd/csrc/dr_common.hpp"]


def main():
    argv = list(sys.argv[1:])
    shape = {}
    for key in ("--pairs", "--points", "--hyps"):          # the workload the capture was taken on (bench.py checks it)
        if key in argv:
            i = argv.index(key)
            shape[key[2:]] = int(argv[i + 1])
            del argv[i:i + 2]
    out_md, out_json, dbs = argv[0], argv[1], argv[2:]
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
    srcs = {rel: hashlib.sha256(open(os.path.join(ROOT, rel), "rb").read()).hexdigest() for rel in KERNEL_SOURCES}
    json.dump({"kernel_sources": srcs, "workload": shape, "kernels": rows}, open(out_json, "w"), indent=1)
    print("wrote", out_md, out_json)


if __name__ == "__main__":
    main()
