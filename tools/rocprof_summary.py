#!/usr/bin/env python
"""Turns a rocprofv3 (rocpd sqlite) capture into the per-kernel summary kept under profiles/.

    rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o bench -- python bench.py ...
    python tools/rocprof_summary.py gpurun_out/prof/bench_results.db profiles/rNN_bench_kernel_stats.md "command line"
"""
import sqlite3
import sys


def main():
    db, out = sys.argv[1], sys.argv[2]
    cmd = sys.argv[3] if len(sys.argv) > 3 else ""
    c = sqlite3.connect(db)
    rows = list(c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    extra = {}
    for name, vg, sg, lds, gx, gy, gz, wx in c.execute(
            "select name, max(vgpr_count), max(sgpr_count), max(lds_size), max(grid_x), max(grid_y), max(grid_z), "
            "max(workgroup_x) from kernels group by name"):
        extra[name] = (vg, sg, lds, gx, gy, gz, wx)
    mn = {n: (a, b) for n, a, b in c.execute("select name, min(duration), max(duration) from kernels group by name")}
    with open(out, "w") as f:
        f.write(f"# rocprofv3 --kernel-trace --stats summary\n\ncommand: `{cmd}`\n\n")
        f.write("| kernel | calls | total (us) | avg (us) | min (us) | max (us) | % | VGPR | SGPR | LDS (B) | grid | block |\n")
        f.write("|---|---|---|---|---|---|---|---|---|---|---|---|\n")
        for name, calls, tot, avg, pct in rows:
            short = name if len(name) < 110 else name[:107] + "..."
            vg, sg, lds, gx, gy, gz, wx = extra.get(name, (0,) * 7)
            lo, hi = mn.get(name, (0, 0))
            f.write(f"| `{short}` | {calls} | {tot:.1f} | {avg:.2f} | {lo / 1e3:.2f} | {hi / 1e3:.2f} | {pct:.2f} | {vg} | {sg} | "
                    f"{lds} | {gx}x{gy}x{gz} | {wx} |\n")
    print("wrote", out)


if __name__ == "__main__":
    main()
