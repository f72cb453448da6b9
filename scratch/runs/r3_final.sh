#!/bin/bash
# round 3, end-of-round validation + the profiles committed under profiles/r3_*:
#   full GPU suite, smoke(), default bench line, the driver's command, train lines (graph / eager), plain two-rank launches (gloo,
#   one shared GPU: functional), rocprofv3 kernel stats of every config over the timed segments, FETCH_SIZE / WRITE_SIZE of the
#   default workload, SQ / GRBM / TCC counters of the scoring kernel at the headline shape
mkdir -p gpurun_out/r3z
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3z
cd $R
timeout 900 python -m pytest tests -m gpu -q --timeout 300 > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?"
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "driver-style rc $?"
timeout 300 python bench.py --mode train --steps 300 > $O/bench_train.json 2> $O/bench_train.err; echo "train rc $?"
timeout 300 python bench.py --mode train --graph off --steps 300 > $O/bench_train_eager.json 2> $O/bench_train_eager.err; echo "train eager rc $?"
timeout 300 python bench.py --gpus 2 --gpus-shared --backend gloo --mode train --steps 30 --warmup 5 --segments 3 > $O/bench_train_2rank.json 2> $O/bench_train_2rank.err; echo "plain 2-rank train rc $?"; head -c 300 $O/bench_train_2rank.json; echo
timeout 300 python bench.py --gpus 2 --gpus-shared --backend gloo --pairs 16 --steps 30 --warmup 5 --segments 3 --no-configs --no-cpu-baseline --no-extras > $O/bench_test_2rank.json 2> $O/bench_test_2rank.err; echo "plain 2-rank test rc $?"; head -c 300 $O/bench_test_2rank.json; echo
python - <<PY
import json
d=json.load(open("$O/bench_default.json"))
print("default", d["value"], d["ms_per_step"], d["segments"]["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d["roofline"]["traffic"], d.get("two_batches_in_flight"), d.get("with_final_refit"), d["sampler_topdown"] and d["sampler_topdown"]["value"])
for k,v in d["configs"].items(): print(k, round(v["ms_per_step"],4), v["issue"][:20], round(v["hypotheses_per_s"]/1e6,1), "eager", round(v["eager_ms_per_step"],4), "graph", v["graph_replay_ms_per_step"], v["launch_ms"], v.get("scoring_roofline",{}).get("frac"))
print("clnet", d["clnet_logits"]["ms_per_step"], d["clnet_logits"]["best_mask_agreement_with_geometric_inliers"], "cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
t=json.load(open("$O/bench_driver.json")); print("driver-style", t["value"], t["ms_per_step"], t["roofline"]["avg_launch_ms"], t["roofline"]["frac"], t["segments"]["ms_per_step"])
t=json.load(open("$O/bench_train.json")); print("train", t["value"], t["ms_per_step"])
t=json.load(open("$O/bench_train_eager.json")); print("train eager", t["value"], t["ms_per_step"])
PY
cd /tmp
run() {  # name  title  bench args...
  name=$1; title=$2; shift 2
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_$name -o bench -- python $R/bench.py "$@" > $O/prof_$name.json 2> $O/prof_$name.err
  python $R/tools/rocprof_summary.py $(find $O/prof_$name -name "*results.db" | head -1) $O/r3_kernel_stats_$name.md "$title" last 100
  rm -rf $O/prof_$name
}
COMMON="--no-configs --no-cpu-baseline --no-extras --steps 20 --warmup 5"
run c2 "python bench.py --steps 20 --warmup 5 --no-configs --no-cpu-baseline --no-extras   (the driver's command without the sub-records and the informational regions; 128 pairs per step)" $COMMON
run c2_p32 "python bench.py --pairs 32 $COMMON" --pairs 32 $COMMON
run c1 "python bench.py --workload c1 --graph off $COMMON" --workload c1 --graph off $COMMON
run c3 "python bench.py --workload c3 $COMMON" --workload c3 $COMMON
run c4 "python bench.py --workload c4 $COMMON" --workload c4 $COMMON
run train "python bench.py --mode train --graph off $COMMON   (32 pairs per step, eager so that the launches are visible one by one)" --mode train --graph off $COMMON
for n in c2 c2_p32 c3 c4; do python -c "
import json,sys; r=json.load(open('$O/prof_$n.json')); print('under rocprofv3: $n', round(r['value']/1e6,2), round(r['ms_per_step'],4), r['roofline']['kernel'], round(r['roofline']['avg_launch_ms'],4), round(r['roofline']['frac'],4))"; done
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $c --kernel-trace -d $O/pmc_$c -o bench -- python $R/bench.py --steps 12 --warmup 3 --segments 1 --prewarm-s 0.05 --no-configs --no-cpu-baseline --no-extras > /dev/null 2> $O/pmc_$c.err
done
python $R/tools/rocprof_pmc_summary.py $O/r3_pmc_fetch_write.md $O/r3_pmc_fetch_write.json --pairs 128 --points 2000 --hyps 1024 $(find $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE -name "*results.db")
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU" \
           "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_TRANS_F32" \
           "GRBM_GUI_ACTIVE GRBM_COUNT TCC_EA0_WRREQ TCC_EA0_WRREQ_STALL TCC_TAG_STALL" \
           "SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SALU"; do
  i=$((i+1))
  K4_PREWARM=40 timeout 200 rocprofv3 --pmc $set --kernel-trace -d $O/p$i -o p$i -- python $R/scratch/k4_general_run.py > $O/p$i.log 2>&1 || echo "set $i failed: $(tail -2 $O/p$i.log | head -c 300)"
done
cd $R
python - > $O/r3_k4_counters_raw.txt <<PY
import sqlite3, glob
for db in sorted(glob.glob("$O/p*/**/*results.db", recursive=True)):
    c = sqlite3.connect(db)
    try:
        for name, n, avg, mn in c.execute("select name, count(*), avg(duration), min(duration) from kernels group by name"):
            if 'msac_score' in name:
                print(f"DURATION {name.split('(')[0][-44:]:44s} n={n} avg_ns={avg:.0f} min_ns={mn:.0f}")
        rows = list(c.execute("select kernel_name, counter_name, count(*), avg(value), min(value) from counters_collection group by kernel_name, counter_name"))
    except Exception as e:
        print(db, e); continue
    for name, counter, n, avg, mn in rows:
        if 'msac_score' in name:
            print(f"{name.split('(')[0][-44:]:44s} {counter:34s} n={n} avg={avg:.6g} min={mn:.6g}")
PY
rm -rf $O/p*/
cat $O/r3_k4_counters_raw.txt
ls $O
