#!/bin/bash
# round 6: SQ counters of the train step's kernels
mkdir -p gpurun_out/trainpmc
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/trainpmc
cd /tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace -d $O/p$i -o p -- python $R/bench.py --mode train --graph off --no-configs --no-cpu-baseline --no-extras --steps 10 --warmup 3 --segments 1 --prewarm-s 0.05 > $O/p$i.log 2>&1 || echo "set $i failed"
done
python - > $O/train_counters_raw.txt <<PY
import sqlite3, glob
for db in sorted(glob.glob("$O/p*/**/*results.db", recursive=True)):
    c = sqlite3.connect(db)
    for name, n, avg in c.execute("select name, count(*), avg(duration) from kernels group by name"):
        if 'dr::' in name: print(f"DURATION {name.split('(')[0][-46:]:46s} n={n} avg_ns={avg:.0f}")
    for name, counter, n, avg in c.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name"):
        if 'dr::' in name: print(f"{name.split('(')[0][-46:]:46s} {counter:22s} n={n} avg={avg:.6g}")
PY
rm -rf $O/p1 $O/p2
cat $O/train_counters_raw.txt | grep -i "episym_kernel\|nister5_pair\|gumbel_bwd\|topk_fast" | grep "DURATION\|SQ_INSTS_VALU\|SQ_ACTIVE_INST_VALU\|SQ_WAVE_CYCLES\|SQ_WAVES\|SQ_BUSY\|GRBM\|SQ_INSTS_LDS\|SQ_WAIT_INST_LDS"
