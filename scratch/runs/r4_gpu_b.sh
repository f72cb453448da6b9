#!/bin/bash
# round 4, pass B: prologue / tail variants of fast16 in the step; SQ counters of the LDS-streamed variant next to fast16;
# one-pair latency today; multi-round calls with and without the second stream
mkdir -p gpurun_out; O=$PWD/gpurun_out; L=$O/r4_b.log; : > $L; R=$PWD
for v in sw tr xc nb all4 trxc; do
  echo "== equal cur $v" >> $L
  timeout 120 python scratch/k4_equal.py cur $v 2>&1 | tail -1 >> $L
  DRANSAC_LIB=$PWD/scratch/libdransac_$v.so timeout 300 python -m pytest tests/test_gpu_msac.py tests/test_gpu_edge_cases.py -x -q 2>&1 | tail -1 >> $L
done
echo "== in-step A/B" >> $L
bash scratch/ab_step.sh cur sw tr xc nb all4 trxc >> $L 2>&1
echo "== one pair (eager / graph)" >> $L
for g in off on; do timeout 120 python bench.py --pairs 1 --steps 300 --warmup 30 --no-configs --no-cpu-baseline --no-extras --graph $g 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('P=1 graph=$g', round(d['ms_per_step'],4), 'ms', round(d['value']/1e6,2), 'M hyps/s')" >> $L; done
echo "== multiround" >> $L
timeout 200 python scratch/multiround.py >> $L 2>&1
echo "== counters" >> $L
cd /tmp; export TMPDIR=/tmp
i=0
for lib in cur L0sc; do
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU" \
           "SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_INSTS_SALU GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  l=""; [ "$lib" != "cur" ] && l=$R/scratch/libdransac_$lib.so
  DRANSAC_LIB=$l K4_PREWARM=40 timeout 200 rocprofv3 --pmc $set --kernel-trace -d $O/r4b_p$i -o p$i -- python $R/scratch/k4_general_run.py > $O/r4b_p$i.log 2>&1 || echo "set $i failed: $(tail -2 $O/r4b_p$i.log | head -c 300)" >> $L
  echo "--- $lib set $i" >> $L
  python - >> $L <<PY
import sqlite3, glob
for db in sorted(glob.glob("$O/r4b_p$i/**/*results.db", recursive=True)):
    c = sqlite3.connect(db)
    try:
        tabs=[r[0] for r in c.execute("select name from sqlite_master where type='view' or type='table'")]
        kt=[t for t in tabs if t=='kernels'][0]
        for name, n, avg, mn in c.execute("select name, count(*), avg(duration), min(duration) from kernels group by name"):
            if 'msac_score' in name: print(f"DURATION {name.split('(')[0][-44:]:44s} n={n} avg_ns={avg:.0f} min_ns={mn:.0f}")
        for kn, cn, n, avg, mn in c.execute("select kernel_name, counter_name, count(*), avg(value), min(value) from counters_collection group by kernel_name, counter_name"):
            if 'msac_score' in kn: print(f"{cn:28s} n={n} avg={avg:.4g} min={mn:.4g}")
    except Exception as e:
        print("db error", e)
PY
  rm -rf $O/r4b_p$i
done
done
