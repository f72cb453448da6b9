#!/bin/bash
# round 3, GPU pass G: rocprofv3 kernel stats of every BASELINE config + the train step (-> profiles/r3_kernel_stats_*.md),
# FETCH_SIZE / WRITE_SIZE of the default workload (-> profiles/r3_pmc_fetch_write.{md,json}), SQ / GRBM / TCC counters of the
# scoring kernel at the headline shape (-> profiles/r3_k4_counters_raw.txt)
mkdir -p gpurun_out/r3g
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3g
cd /tmp
run() {  # name  title  bench args...
  name=$1; title=$2; shift 2
  timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof_$name -o bench -- python $R/bench.py "$@" > $O/prof_$name.json 2> $O/prof_$name.err
  python $R/tools/rocprof_summary.py $(find $O/prof_$name -name "*results.db" | head -1) $O/r3_kernel_stats_$name.md "$title" all 0
  rm -rf $O/prof_$name
}
COMMON="--no-configs --no-cpu-baseline --no-extras --prewarm-s 0.05 --segments 5 --steps 20 --warmup 5"
run c2 "python bench.py $COMMON   (128 pairs per step: 8+ pre-conditioning, 5 warm-up, 5 x 20 timed steps)" $COMMON
run c2_p32 "python bench.py --pairs 32 $COMMON" --pairs 32 $COMMON
run c1 "python bench.py --workload c1 --graph off $COMMON" --workload c1 --graph off $COMMON
run c3 "python bench.py --workload c3 $COMMON" --workload c3 $COMMON
run c4 "python bench.py --workload c4 $COMMON" --workload c4 $COMMON
run train "python bench.py --mode train --graph off $COMMON   (32 pairs per step, eager so that the launches are visible one by one)" --mode train --graph off $COMMON
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $c --kernel-trace -d $O/pmc_$c -o bench -- python $R/bench.py --steps 12 --warmup 3 --segments 1 --prewarm-s 0.05 --no-configs --no-cpu-baseline --no-extras > /dev/null 2> $O/pmc_$c.err
done
python $R/tools/rocprof_pmc_summary.py $O/r3_pmc_fetch_write.md $O/r3_pmc_fetch_write.json --pairs 128 --points 2000 --hyps 1024 $(find $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE -name "*results.db")
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
rocprofv3 -L 2>/dev/null | grep -oE "(SQ|GRBM)_[A-Z0-9_]+" | sort -u | tr '\n' ' ' > $O/sq_counter_names.txt
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU" \
           "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM" \
           "GRBM_GUI_ACTIVE GRBM_COUNT TCC_EA0_WRREQ TCC_EA0_WRREQ_STALL TCC_TAG_STALL" \
           "SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --kernel-trace -d $O/p$i -o p$i -- python $R/scratch/k4_general_run.py > $O/p$i.log 2>&1 || echo "set $i failed: $(tail -2 $O/p$i.log | head -c 300)"
done
cd $R
python - > $O/r3_k4_counters_raw.txt <<PY
import sqlite3, glob
for db in sorted(glob.glob("$O/p*/**/*results.db", recursive=True)):
    c = sqlite3.connect(db)
    try:
        rows = list(c.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name"))
    except Exception as e:
        print(db, e); continue
    try:
        for name, n, avg, mn in c.execute("select name, count(*), avg(duration), min(duration) from kernels group by name"):
            if 'msac_score' in name or 'fill' in name.lower():
                print(f"DURATION {name.split('(')[0][-44:]:44s} n={n} avg_ns={avg:.0f} min_ns={mn:.0f}")
    except Exception as e:
        print("durations:", e)
    for name, counter, n, avg in rows:
        if 'msac_score' in name or 'FillFunctor' in name or 'fill' in name.lower():
            print(f"{name.split('(')[0][-44:]:44s} {counter:34s} n={n} avg={avg:.6g}")
PY
rm -rf $O/p*/
cat $O/r3_k4_counters_raw.txt | head -60
ls $O
