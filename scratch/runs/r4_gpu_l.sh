#!/bin/bash
# round 4, pass L: SQ counters of the rigid residual kernel at config 4
mkdir -p gpurun_out; O=$PWD/gpurun_out; L=$O/r4_l.log; : > $L; R=$PWD
cd /tmp; export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU" \
           "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --kernel-trace -d $O/r4l_p$i -o p$i -- python $R/scratch/k4r_run.py > $O/r4l_p$i.log 2>&1 || echo "set $i failed: $(tail -2 $O/r4l_p$i.log | head -c 300)" >> $L
  python - >> $L <<PY
import sqlite3, glob
for db in sorted(glob.glob("$O/r4l_p$i/**/*results.db", recursive=True)):
    c = sqlite3.connect(db)
    for name, n, avg, mn in c.execute("select name, count(*), avg(duration), min(duration) from kernels group by name"):
        if 'rigid_residual' in name: print(f"DURATION {name.split('(')[0][-44:]:44s} n={n} avg_ns={avg:.0f} min_ns={mn:.0f}")
    for kn, cn, n, avg, mn in c.execute("select kernel_name, counter_name, count(*), avg(value), min(value) from counters_collection group by kernel_name, counter_name"):
        if 'rigid_residual' in kn: print(f"{cn:28s} n={n} avg={avg:.4g} min={mn:.4g}")
PY
  rm -rf $O/r4l_p$i
done
