#!/bin/bash
mkdir -p gpurun_out; O=$PWD/gpurun_out; L=$O/r4_w.log; : > $L; R=$PWD
cd /tmp; export TMPDIR=/tmp
for scr in 1 0; do
  echo "== screen=$scr" >> $L
  SCR=$scr timeout 200 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU --kernel-trace -d $O/r4w_$scr -o p -- python $R/scratch/k1s_run.py > $O/r4w_$scr.log 2>&1 || echo failed >> $L
  python - >> $L <<PY
import sqlite3, glob
for db in sorted(glob.glob("$O/r4w_$scr/**/*results.db", recursive=True)):
    c = sqlite3.connect(db)
    for name, n, avg, mn in c.execute("select name, count(*), avg(duration), min(duration) from kernels group by name"):
        if 'gumbel' in name: print(f"DURATION {name.split('(')[0][-50:]:50s} n={n} avg_ns={avg:.0f} min_ns={mn:.0f}")
    for kn, cn, n, avg, mn in c.execute("select kernel_name, counter_name, count(*), avg(value), min(value) from counters_collection group by kernel_name, counter_name"):
        if 'stream' in kn: print(f"{cn:28s} n={n} avg={avg:.4g}")
PY
  rm -rf $O/r4w_$scr
done
