#!/bin/bash
# round 5: train-mode sampler + gather in one launch each way (A/B through DRANSAC_FUSED_SAMPLE_GATHER), screening words of the long-row
# sampler in one launch (A/B against scratch/libdransac_screen2.so = -DDR_K1_SCREEN_FUSED=0), 3-D update kernel with its points
# requested before the arg-min; full GPU suite first
mkdir -p gpurun_out/r5s
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5s
cd $R
timeout 1500 python -m pytest tests -m gpu -q --timeout 300 -x > $O/pytest.log 2>&1; tail -5 $O/pytest.log
for round in 1 2; do
  for f in 0 1; do
    DRANSAC_FUSED_SAMPLE_GATHER=$f timeout 200 python bench.py --mode train --steps 300 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('train fused=$f', round(d['ms_per_step'],4), 'ms', d['segments']['ms_per_step'])"
  done
done
AB_ARGS="--workload c4" bash scratch/ab_step.sh screen2 cur
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_c4 -o bench -- python $R/bench.py --workload c4 --no-configs --no-cpu-baseline --no-extras --steps 20 --warmup 5 > $O/prof_c4.json 2> $O/prof_c4.err
python $R/tools/rocprof_summary.py $(find $O/prof_c4 -name "*results.db" | head -1) $O/kernel_stats_c4.md "c4" last 100
rm -rf $O/prof_c4
cut -c1-150 $O/kernel_stats_c4.md | head -20
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_tr -o bench -- python $R/bench.py --mode train --graph off --no-configs --no-cpu-baseline --no-extras --steps 20 --warmup 5 > $O/prof_tr.json 2> $O/prof_tr.err
python $R/tools/rocprof_summary.py $(find $O/prof_tr -name "*results.db" | head -1) $O/kernel_stats_train.md "train" last 100
rm -rf $O/prof_tr
cut -c1-150 $O/kernel_stats_train.md | head -22
