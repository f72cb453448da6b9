#!/bin/bash
mkdir -p gpurun_out/r3o
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3o
(timeout 600 python -m pytest tests -m gpu -q -x --timeout 300 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log)
tail -3 $O/pytest.log
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_c4 -o bench -- python $R/bench.py --workload c4 --no-configs --no-cpu-baseline --no-extras --steps 20 --warmup 5 > $O/prof_c4.json 2> $O/prof_c4.err
python $R/tools/rocprof_summary.py $(find $O/prof_c4 -name "*results.db" | head -1) $O/r3_kernel_stats_c4.md "python bench.py --workload c4 --no-configs --no-cpu-baseline --no-extras --steps 20 --warmup 5" last 100
rm -rf $O/prof_c4
sed -n 9,22p $O/r3_kernel_stats_c4.md | cut -c1-70,110-160
cd $R
timeout 200 python bench.py --workload c4 --steps 300 --warmup 5 --no-configs --no-cpu-baseline --no-extras > $O/bench_c4.json 2> $O/bench_c4.err
python -c "
import json; r=json.load(open('$O/bench_c4.json')); print('c4 eager', r['value'], r['ms_per_step'], r['roofline']['avg_launch_ms'], r['roofline']['frac'])"
timeout 200 python bench.py --workload c4 --graph on --steps 300 --warmup 5 --no-configs --no-cpu-baseline --no-extras > $O/bench_c4_graph.json 2> $O/bench_c4_graph.err
python -c "
import json; r=json.load(open('$O/bench_c4_graph.json')); print('c4 graph', r['value'], r['ms_per_step'])"
