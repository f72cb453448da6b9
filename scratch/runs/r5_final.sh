#!/bin/bash
# round 5, end-of-round validation + the profiles committed under profiles/r5_*:
#   full GPU suite, smoke(), default bench line, the driver's command, train lines (graph / eager), plain two-rank launches (gloo,
#   one shared GPU: functional; the train one exercises the asynchronous gradient bucket), rocprofv3 kernel stats of every config
#   over the timed segments, FETCH_SIZE / WRITE_SIZE of the default workload
mkdir -p gpurun_out/r5z
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5z
cd $R
timeout 1500 python -m pytest tests -m gpu -q --timeout 300 > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
( time timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | grep real; echo "bench rc $?"
( time timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err ) 2>&1 | grep real; echo "driver-style rc $?"
timeout 300 python bench.py --mode train --steps 300 > $O/bench_train.json 2> $O/bench_train.err; echo "train rc $?"
timeout 300 python bench.py --mode train --graph off --steps 300 > $O/bench_train_eager.json 2> $O/bench_train_eager.err; echo "train eager rc $?"
timeout 300 python bench.py --gpus 2 --gpus-shared --backend gloo --mode train --steps 30 --warmup 5 --segments 3 > $O/bench_train_2rank.json 2> $O/bench_train_2rank.err; echo "plain 2-rank train rc $?"; head -c 600 $O/bench_train_2rank.json; echo; tail -3 $O/bench_train_2rank.err
timeout 300 python bench.py --gpus 2 --gpus-shared --backend gloo --pairs 16 --steps 30 --warmup 5 --segments 3 --no-configs --no-cpu-baseline --no-extras > $O/bench_test_2rank.json 2> $O/bench_test_2rank.err; echo "plain 2-rank test rc $?"; head -c 300 $O/bench_test_2rank.json; echo
python - <<PY
import json
d=json.load(open("$O/bench_default.json"))
print("default", d["value"], d["ms_per_step"], d["segments"]["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d["roofline"]["traffic"], d.get("two_batches_in_flight"), d.get("with_final_refit"), d["sampler_topdown"] and d["sampler_topdown"]["value"])
for k,v in d["configs"].items():
  if "ms_per_step" in v: print(k, round(v["ms_per_step"],4), v["issue"][:20], round(v["hypotheses_per_s"]/1e6,1), "eager", round(v["eager_ms_per_step"],4), "graph", v.get("graph_replay_ms_per_step"), v["launch_ms"], v.get("scoring_roofline",{}).get("frac"))
print("fused", d["fused_driver"]["ms_per_step"], d["fused_driver"]["hypotheses_per_s"], d["fused_driver"]["scoring_roofline"])
print("all_valid", d["k4_all_valid"])
print("clnet", d["clnet_logits"]["ms_per_step"], d["clnet_logits"]["best_mask_agreement_with_geometric_inliers"], "cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
print("dropin", d["configs"].get("dropin_layer_loop"))
for k,v in d["configs"].items():
    if "cpu_baseline" in v: print("cpu", k, round(v["cpu_baseline"]["value"],1), round(v["cpu_baseline"]["single_thread_value"],1), v["cpu_baseline"]["cores"])
print("cpu c2", d["cpu_baseline"]["value"], d["cpu_baseline"].get("single_thread_value"), d["cpu_baseline"]["cores"])
t=json.load(open("$O/bench_driver.json")); print("driver-style", t["value"], t["ms_per_step"], t["roofline"]["avg_launch_ms"], t["roofline"]["frac"], t["segments"]["ms_per_step"])
t=json.load(open("$O/bench_train.json")); print("train", t["value"], t["ms_per_step"])
t=json.load(open("$O/bench_train_eager.json")); print("train eager", t["value"], t["ms_per_step"])
PY
cd /tmp
run() {  # name  title  bench args...
  name=$1; title=$2; shift 2
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_$name -o bench -- python $R/bench.py "$@" > $O/prof_$name.json 2> $O/prof_$name.err
  python $R/tools/rocprof_summary.py $(find $O/prof_$name -name "*results.db" | head -1) $O/r5_kernel_stats_$name.md "$title" last 100
  rm -rf $O/prof_$name
}
COMMON="--no-configs --no-cpu-baseline --no-extras --steps 20 --warmup 5"
run c2 "python bench.py --steps 20 --warmup 5 --no-configs --no-cpu-baseline --no-extras   (the driver's command without the sub-records and the informational regions; 128 pairs per step)" $COMMON
run c2_p32 "python bench.py --pairs 32 $COMMON" --pairs 32 $COMMON
run c2_p1 "python bench.py --pairs 1 --graph off $COMMON   (one pair per call, eager so that the launches are visible one by one)" --pairs 1 --graph off $COMMON
run c1 "python bench.py --workload c1 --graph off $COMMON" --workload c1 --graph off $COMMON
run c3 "python bench.py --workload c3 $COMMON" --workload c3 $COMMON
run c4 "python bench.py --workload c4 $COMMON" --workload c4 $COMMON
run train "python bench.py --mode train --graph off $COMMON   (32 pairs per step, eager so that the launches are visible one by one)" --mode train --graph off $COMMON
for n in c2 c2_p32 c3 c4; do python -c "
import json,sys; r=json.load(open('$O/prof_$n.json')); print('under rocprofv3: $n', round(r['value']/1e6,2), round(r['ms_per_step'],4), r['roofline']['kernel'], round(r['roofline']['avg_launch_ms'],4), round(r['roofline']['frac'],4))"; done
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $c --kernel-trace -d $O/pmc_$c -o bench -- python $R/bench.py --steps 12 --warmup 3 --segments 1 --prewarm-s 0.05 --no-configs --no-cpu-baseline --no-extras > /dev/null 2> $O/pmc_$c.err
done
python $R/tools/rocprof_pmc_summary.py $O/r5_pmc_fetch_write.md $O/r5_pmc_fetch_write.json --pairs 128 --points 2000 --hyps 1024 $(find $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE -name "*results.db")
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
# the headline workload WITH the final refit (K7), per launch; the per-pair drop-in loop, per launch
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_refit -o refit -- python $R/scratch/refit_step.py > $O/prof_refit.log 2>&1; grep "refit=" $O/prof_refit.log
python $R/tools/rocprof_summary.py $(find $O/prof_refit -name "*results.db" | head -1) $O/r5_kernel_stats_with_refit.md "python scratch/refit_step.py (128 pairs x 2000 points x 1024 hypotheses per step, eager; the first 110 dispatches of a kernel run WITHOUT the refit, the last 110 with it)" last 100
rm -rf $O/prof_refit
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_dropin -o dropin -- python $R/scratch/dropin_loop.py > $O/prof_dropin.log 2>&1; grep -v amdgpu.ids $O/prof_dropin.log | tail -6
python $R/tools/rocprof_summary.py $(find $O/prof_dropin -name "*results.db" | head -1) $O/r5_kernel_stats_dropin.md "python scratch/dropin_loop.py (32 pairs one by one through layers.RANSACLayer.forward, test mode, graph replay per pair)" last 2000
rm -rf $O/prof_dropin
cd $R; python scratch/dropin_host.py 2>&1 | grep -v amdgpu.ids | tail -12
