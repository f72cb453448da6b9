#!/bin/bash
# round 6: SQ counters of the register sampler, selection on wave masks (tree) vs the list selection (variant library)
mkdir -p gpurun_out/k1sel
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/k1sel
cd /tmp
for v in new old; do
  lib=""; [ $v = old ] && lib=$R/scratch/libdransac_k1_oldsel.so
  i=0
  for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
             "SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    DRANSAC_LIB=$lib timeout 200 rocprofv3 --pmc $set --kernel-trace -d $O/p_${v}_$i -o p -- python $R/scratch/k1_pmc_target.py > $O/p_${v}_$i.log 2>&1 || echo "set $v $i failed"
  done
done
python - > $O/k1_counters_raw.txt <<PY
import sqlite3, glob
for db in sorted(glob.glob("$O/p_*/**/*results.db", recursive=True)):
    c = sqlite3.connect(db)
    tag = db.split("/p_")[1].split("/")[0]
    for name, n, avg, mn in c.execute("select name, count(*), avg(duration), min(duration) from kernels group by name"):
        if 'topk_fast' in name: print(f"{tag} DURATION n={n} avg_ns={avg:.0f} min_ns={mn:.0f}")
    for name, counter, n, avg in c.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name"):
        if 'topk_fast' in name: print(f"{tag} {counter:24s} n={n} avg={avg:.6g}")
PY
rm -rf $O/p_new_* $O/p_old_*
cat $O/k1_counters_raw.txt
