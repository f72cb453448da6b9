#!/bin/bash
# round 4, pass F: suite on the current tree, c4 with the two-kernel screen, PMC FETCH_SIZE / WRITE_SIZE of the headline step
mkdir -p gpurun_out/r4f; export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/r4f; L=$O/log.txt; : > $L
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -3 >> $L
timeout 200 python bench.py --workload c4 --steps 300 --warmup 30 --no-configs --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c4', round(d['ms_per_step'],4), 'ms  residual launch', round(d['roofline']['avg_launch_ms'],4), 'ms frac', round(d['roofline']['frac'],3))" >> $L
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_c4 -o bench -- python $R/bench.py --workload c4 --no-configs --no-cpu-baseline --no-extras --steps 20 --warmup 5 > $O/prof_c4.json 2> $O/prof_c4.err
python $R/tools/rocprof_summary.py $(find $O/prof_c4 -name "*results.db" | head -1) $O/r4_kernel_stats_c4.md "python bench.py --workload c4 --no-configs --no-cpu-baseline --no-extras --steps 20 --warmup 5" last 100 >> $L
rm -rf $O/prof_c4
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $c --kernel-trace -d $O/pmc_$c -o bench -- python $R/bench.py --steps 12 --warmup 3 --segments 1 --prewarm-s 0.05 --no-configs --no-cpu-baseline --no-extras > /dev/null 2> $O/pmc_$c.err
done
python $R/tools/rocprof_pmc_summary.py $O/r4_pmc_fetch_write.md $O/r4_pmc_fetch_write.json --pairs 128 --points 2000 --hyps 1024 $(find $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE -name "*results.db") >> $L 2>&1
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
cat $O/r4_kernel_stats_c4.md >> $L
