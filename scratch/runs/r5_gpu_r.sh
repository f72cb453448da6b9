#!/bin/bash
# round 5: SQ counters of the five-point kernels (lane pairs and two-phase, Nister and Stewenius) at 131 072 samples
mkdir -p gpurun_out/r5r
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5r
cd /tmp; export PYTHONPATH=$R
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --kernel-trace -d $O/p$i -o p$i -- python $R/scratch/k3_pmc_target.py > $O/p$i.log 2>&1 || echo "set $i failed: $(tail -2 $O/p$i.log | head -c 300)"
done
cd $R
python - > $O/r5_k3_counters_raw.txt <<PY
import sqlite3, glob
for db in sorted(glob.glob("$O/p*/**/*results.db", recursive=True)):
    c = sqlite3.connect(db)
    try:
        for name, n, avg, mn in c.execute("select name, count(*), avg(duration), min(duration) from kernels group by name"):
            if '5_pair' in name or '5_fb' in name:
                print(f"DURATION {name.split('(')[0][-44:]:44s} n={n} avg_ns={avg:.0f} min_ns={mn:.0f}")
        rows = list(c.execute("select kernel_name, counter_name, count(*), avg(value), min(value) from counters_collection group by kernel_name, counter_name"))
    except Exception as e:
        print(db, e); continue
    for name, counter, n, avg, mn in rows:
        if '5_pair' in name or '5_fb' in name:
            print(f"{name.split('(')[0][-44:]:44s} {counter:34s} n={n} avg={avg:.6g} min={mn:.6g}")
PY
rm -rf $O/p*/
cat $O/r5_k3_counters_raw.txt
