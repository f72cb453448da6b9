#!/bin/bash
# round-2 final measurements: default bench line, rocprofv3 kernel stats + PMC of the default (128-pair) workload, train line
mkdir -p gpurun_out/r2g
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2g
cd $R
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?"; head -c 700 $O/bench_default.json; echo
timeout 300 python bench.py --mode train --steps 300 > $O/bench_train.json 2> $O/bench_train.err; head -c 400 $O/bench_train.json; echo
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_c2 -o bench -- python $R/bench.py --steps 100 --warmup 5 --no-configs --no-cpu-baseline > $O/prof_c2.json 2> $O/prof_c2.err
python $R/tools/rocprof_summary.py $(find $O/prof_c2 -name "*results.db" | head -1) $O/r2_kernel_stats_c2.md "python bench.py --steps 100 --warmup 5 --no-configs --no-cpu-baseline   (default workload: c2, 128 pairs per step)" first 105
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_train -o bench -- python $R/bench.py --mode train --steps 100 --warmup 5 > $O/prof_train.json 2> $O/prof_train.err
python $R/tools/rocprof_summary.py $(find $O/prof_train -name "*results.db" | head -1) $O/r2_kernel_stats_train.md "python bench.py --mode train --steps 100 --warmup 5   (32 pairs per step)"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace -d $O/pmc_$c -o bench -- python $R/bench.py --steps 12 --warmup 3 --no-configs --no-cpu-baseline > /dev/null 2> $O/pmc_$c.err
done
python $R/tools/rocprof_pmc_summary.py $O/r2_pmc_fetch_write.md $O/r2_pmc_fetch_write.json --pairs 128 --points 2000 --hyps 1024 $(find $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE -name "*results.db")
rm -rf $O/prof_c2 $O/prof_train $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
ls $O
