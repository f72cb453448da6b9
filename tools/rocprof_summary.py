#!/usr/bin/env python
"""Turns a rocprofv3 (rocpd sqlite) capture into the per-kernel summary kept under profiles/.

    rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o bench -- python bench.py ...
    python tools/rocprof_summary.py gpurun_out/prof/bench_results.db profiles/rNN_bench_kernel_stats.md "command line" [first N | skip N]

`first N` / `skip N` / `last N`: statistics over the first N dispatches of every kernel only (over everything after them /
over the last N) -- bench.py runs a time-based pre-conditioning and its warm-up before the timed segments; `last N` with
N = segments x steps is the timed region.
"""
import sqlite3
import sys


def main():
    db, out = sys.argv[1], sys.argv[2]
    cmd = sys.argv[3] if len(sys.argv) > 3 else ""
    mode, n = (sys.argv[4], int(sys.argv[5])) if len(sys.argv) > 5 else ("all", 0)
    c = sqlite3.connect(db)
    per = {}
    for name, dur, vg, sg, lds, gx, gy, gz, wx in c.execute(
            "select name, duration, vgpr_count, sgpr_count, lds_size, grid_x, grid_y, grid_z, workgroup_x from kernels "
            "order by start"):
        per.setdefault(name, []).append((dur, vg, sg, lds, gx, gy, gz, wx))
    rows = []
    for name, lst in per.items():
        sel = lst[:n] if mode == "first" else (lst[n:] if mode == "skip" else (lst[-n:] if mode == "last" else lst))
        if not sel:
            continue
        durs = [x[0] for x in sel]
        rows.append((name, len(durs), sum(durs) / 1e3, sum(durs) / len(durs) / 1e3, min(durs) / 1e3, max(durs) / 1e3, sel[0][1:]))
    total = sum(r[2] for r in rows) or 1.0
    rows.sort(key=lambda r: -r[2])
    with open(out, "w") as f:
        f.write(f"# rocprofv3 --kernel-trace --stats summary\n\ncommand: `{cmd}`\n\n")
        if mode != "all":
            which = {"first": "the first", "skip": "all after the first", "last": "the last"}[mode]
            f.write(f"dispatches: {which} {n} of every kernel"
                    + (" (the timed segments: the time-based pre-conditioning and the warm-up come before them)\n\n"
                       if mode == "last" else "\n\n"))
        f.write("| kernel | calls | total (us) | avg (us) | min (us) | max (us) | % | VGPR | SGPR | LDS (B) | grid | block |\n")
        f.write("|---|---|---|---|---|---|---|---|---|---|---|---|\n")
        for name, calls, tot, avg, lo, hi, (vg, sg, lds, gx, gy, gz, wx) in rows:
            short = name if len(name) < 110 else name[:107] + "..."
            f.write(f"| `{short}` | {calls} | {tot:.1f} | {avg:.2f} | {lo:.2f} | {hi:.2f} | {100 * tot / total:.2f} | {vg} | {sg} | "
                    f"{lds} | {gx}x{gy}x{gz} | {wx} |\n")
    print("wrote", out)


if __name__ == "__main__":
    main()
