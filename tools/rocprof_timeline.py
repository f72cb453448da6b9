#!/usr/bin/env python
"""Timeline of ONE call out of a rocprofv3 (rocpd sqlite) kernel trace: every dispatch between two consecutive dispatches of an
anchor kernel (e.g. `ransac_init_kernel`: the first kernel of a test-mode call), start / end relative to the anchor's start.

    python tools/rocprof_timeline.py capture_results.db ransac_init_kernel [call index from the end, default 3] > timeline.md
"""
import sqlite3
import sys


def main():
    db, anchor = sys.argv[1], sys.argv[2]
    back = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    c = sqlite3.connect(db)
    cols = {r[1] for r in c.execute("pragma table_info(kernels)")}
    stream = "stream_id" if "stream_id" in cols else ("queue_id" if "queue_id" in cols else "0")
    rows = list(c.execute(f"select name, start, start + duration, grid_x, workgroup_x, {stream} from kernels order by start"))
    marks = [i for i, r in enumerate(rows) if anchor in r[0]]
    if len(marks) < back + 1:
        raise SystemExit("not enough anchor dispatches")
    lo, hi = marks[-back - 1], marks[-back]
    t0 = rows[lo][1]
    print("| # | kernel | stream | start (us) | end (us) | duration (us) | gap to previous end (us) | blocks |")
    print("|---|---|---|---|---|---|---|---|")
    prev_end = None
    for i, (name, s, e, gx, wx, st) in enumerate(rows[lo:hi]):
        short = name.split("(")[0].replace("void ", "")
        short = short if len(short) < 70 else short[:67] + "..."
        gap = "" if prev_end is None else f"{(s - prev_end) / 1e3:.2f}"
        print(f"| {i} | `{short}` | {st} | {(s - t0) / 1e3:.2f} | {(e - t0) / 1e3:.2f} | {(e - s) / 1e3:.2f} | {gap} | {gx // max(wx, 1)} |")
        prev_end = e if prev_end is None else max(prev_end, e)
    print(f"\ncall length (anchor start to the next anchor start): {(rows[hi][1] - t0) / 1e3:.2f} us")


if __name__ == "__main__":
    main()
