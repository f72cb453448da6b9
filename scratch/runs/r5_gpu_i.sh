#!/bin/bash
mkdir -p gpurun_out/r5i
O=$PWD/gpurun_out/r5i
R=$PWD
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_train -o bench -- python $R/bench.py --mode train --graph off --no-configs --no-cpu-baseline --no-extras --steps 20 --warmup 5 > $O/prof_train.json 2> $O/prof_train.err
python $R/tools/rocprof_summary.py $(find $O/prof_train -name "*results.db" | head -1) $O/r5_kernel_stats_train.md "python bench.py --mode train --graph off --no-configs --no-cpu-baseline --no-extras --steps 20 --warmup 5   (32 pairs per step, eager so that the launches are visible one by one)" last 100
rm -rf $O/prof_train
head -30 $O/r5_kernel_stats_train.md | cut -c1-230
