#!/bin/bash
mkdir -p gpurun_out/r4t; export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/r4t; L=$O/log.txt; : > $L
timeout 600 python -m pytest tests/test_gpu_round4.py tests/test_gpu_configs.py tests/test_gpu_sampler.py tests/test_gpu_graphs.py -q 2>&1 | tail -3 >> $L
timeout 200 python scratch/fuzz_r4.py 30 2>&1 | tail -2 >> $L
for r in 1 2; do timeout 200 python bench.py --workload c4 --steps 300 --warmup 30 --no-configs --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c4', round(d['ms_per_step'],4), 'ms  residual launch', round(d['roofline']['avg_launch_ms'],4), 'ms frac', round(d['roofline']['frac'],3))" >> $L; done
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_c4 -o bench -- python $R/bench.py --workload c4 --no-configs --no-cpu-baseline --no-extras --steps 20 --warmup 5 > $O/prof_c4.json 2> $O/prof_c4.err
python $R/tools/rocprof_summary.py $(find $O/prof_c4 -name "*results.db" | head -1) $O/r4_kernel_stats_c4.md "python bench.py --workload c4 --no-configs --no-cpu-baseline --no-extras --steps 20 --warmup 5" last 100 >> $L
rm -rf $O/prof_c4
cut -c1-60,105-190 $O/r4_kernel_stats_c4.md >> $L
