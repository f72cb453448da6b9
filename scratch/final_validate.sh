#!/bin/bash
# round-end validation on the GPU box: full GPU test suite, smoke, profiles of the default bench command and of the train step
R=$GRAFT_REPO_ROOT
cd $R
timeout 400 python -m pytest tests -q -m gpu 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py 2>&1 | tail -1 > gpurun_out/bench_line_final.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_v9 -o bench -- python $R/bench.py --no-cpu-baseline > $R/gpurun_out/prof_v9_bench.log 2>&1
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_v9t -o train -- python $R/bench.py --mode train --steps 20 --warmup 3 > $R/gpurun_out/prof_v9_train.log 2>&1
cd $R
python tools/rocprof_summary.py gpurun_out/prof_v9/bench_results.db gpurun_out/r1_bench_kernel_stats_v10.md "rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline  (default workload: warm-up 5 + 50 timed steps on one stream)" first 55
python tools/rocprof_summary.py gpurun_out/prof_v9t/train_results.db gpurun_out/r1_train_kernel_stats_v10.md "rocprofv3 --kernel-trace --stats -- python bench.py --mode train --steps 20 --warmup 3  (train step: sample, solve, best-of-10 vs GT, MatchLoss, backward to the logits)"
tail -1 gpurun_out/prof_v9_bench.log | cut -c1-200
rm -rf gpurun_out/prof_v9 gpurun_out/prof_v9t
