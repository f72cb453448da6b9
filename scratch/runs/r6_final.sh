#!/bin/bash
# round 6, end-of-round validation + the profiles committed under profiles/r6_*:
#   full GPU suite, smoke(), rocprofv3 kernel stats of every config over the timed segments, FETCH_SIZE / WRITE_SIZE of the headline
#   workload and of configs 3 and 4 (separate --pmc passes, --kernel-trace only), SQ counters of the five-point kernels, the drop-in
#   loop at -rbs 1024 and -rbs 64 per launch with the timeline of one replayed call, 2- and 8-rank gloo lines on the one shared GPU
mkdir -p gpurun_out/r6z
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6z
cd $R
timeout 1500 python -m pytest tests -m gpu -q --timeout 300 > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 300 python bench.py --mode train --steps 300 > $O/bench_train.json 2> $O/bench_train.err; echo "train rc $?"
timeout 300 python bench.py --gpus 2 --gpus-shared --backend gloo --mode train --steps 30 --warmup 5 --segments 3 > $O/bench_train_2rank.json 2> $O/bench_train_2rank.err; echo "plain 2-rank train rc $?"
timeout 300 python bench.py --gpus 2 --gpus-shared --backend gloo --pairs 16 --steps 30 --warmup 5 --segments 3 --no-configs --no-cpu-baseline --no-extras > $O/bench_test_2rank.json 2> $O/bench_test_2rank.err; echo "plain 2-rank test rc $?"
timeout 600 python bench.py --gpus 8 --gpus-shared --backend gloo --mode train --steps 10 --warmup 3 --segments 3 > $O/bench_train_8rank.json 2> $O/bench_train_8rank.err; echo "plain 8-rank train rc $?"; grep "^{" $O/bench_train_8rank.json | head -c 900; echo; tail -2 $O/bench_train_8rank.err
cd /tmp
run() {  # name  title  bench args...
  name=$1; title=$2; shift 2
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_$name -o bench -- python $R/bench.py "$@" > $O/prof_$name.json 2> $O/prof_$name.err
  python $R/tools/rocprof_summary.py $(find $O/prof_$name -name "*results.db" | head -1) $O/r6_kernel_stats_$name.md "$title" last 100
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
import json,sys; r=json.loads([l for l in open('$O/prof_$n.json') if l.startswith('{')][-1]); print('under rocprofv3: $n', round(r['value']/1e6,2), round(r['ms_per_step'],4), r['roofline']['kernel'], round(r['roofline']['avg_launch_ms'],4), round(r['roofline']['frac'],4))"; done
# FETCH_SIZE / WRITE_SIZE: separate passes per counter
for w in c2 c3 c4; do
  wl=""; [ $w != c2 ] && wl="--workload $w"
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 200 rocprofv3 --pmc $c --kernel-trace -d $O/pmc_${w}_$c -o bench -- python $R/bench.py $wl --steps 12 --warmup 3 --segments 1 --prewarm-s 0.05 --no-configs --no-cpu-baseline --no-extras > /dev/null 2> $O/pmc_${w}_$c.err
  done
  if [ $w = c2 ]; then
    python $R/tools/rocprof_pmc_summary.py $O/r6_pmc_fetch_write.md $O/r6_pmc_fetch_write.json --pairs 128 --points 2000 --hyps 1024 $(find $O/pmc_${w}_FETCH_SIZE $O/pmc_${w}_WRITE_SIZE -name "*results.db")
  else
    python $R/tools/rocprof_pmc_summary.py $O/r6_pmc_fetch_write_$w.md $O/r6_pmc_fetch_write_$w.json $(find $O/pmc_${w}_FETCH_SIZE $O/pmc_${w}_WRITE_SIZE -name "*results.db")
  fi
  rm -rf $O/pmc_${w}_FETCH_SIZE $O/pmc_${w}_WRITE_SIZE
done
# SQ counters of the five-point kernels at 131 072 samples
export PYTHONPATH=$R
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --kernel-trace -d $O/p$i -o p$i -- python $R/scratch/k3_pmc_target.py > $O/p$i.log 2>&1 || echo "set $i failed: $(tail -2 $O/p$i.log | head -c 300)"
done
python - > $O/r6_k3_counters_raw.txt <<PY
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
rm -rf $O/p1 $O/p2
# the headline workload WITH the final refit (K7), per launch; the per-pair drop-in loop, per launch + the timeline of one call
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_refit -o refit -- python $R/scratch/refit_step.py > $O/prof_refit.log 2>&1; grep "refit=" $O/prof_refit.log
python $R/tools/rocprof_summary.py $(find $O/prof_refit -name "*results.db" | head -1) $O/r6_kernel_stats_with_refit.md "python scratch/refit_step.py (128 pairs x 2000 points x 1024 hypotheses per step, eager; the first 110 dispatches of a kernel run WITHOUT the refit, the last 110 with it)" last 100
rm -rf $O/prof_refit
for rbs in 1024 64; do
  DROPIN_RBS=$rbs timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_dropin_$rbs -o dropin -- python $R/scratch/dropin_loop.py > $O/prof_dropin_$rbs.log 2>&1; grep -v amdgpu.ids $O/prof_dropin_$rbs.log | tail -3
  db=$(find $O/prof_dropin_$rbs -name "*results.db" | head -1)
  python $R/tools/rocprof_summary.py $db $O/r6_kernel_stats_dropin_rbs$rbs.md "DROPIN_RBS=$rbs python scratch/dropin_loop.py (32 pairs one by one through layers.RANSACLayer.forward, test mode, -rbs $rbs, one replayed graph per pair)" last 2000
  python $R/tools/rocprof_timeline.py $db ransac_init_kernel 4 > $O/r6_dropin_timeline_rbs$rbs.md 2>&1
  rm -rf $O/prof_dropin_$rbs
done
cd $R
for rep in 1 2 3; do for rbs in 1024 64; do echo "dropin rbs=$rbs $(DROPIN_RBS=$rbs timeout 300 python scratch/dropin_loop.py 2>&1 | grep 'ms per pair')"; done; done
ls $O
