#!/bin/bash
mkdir -p gpurun_out/r3l
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3l
(timeout 600 python -m pytest tests -m gpu -q -x --timeout 300 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log)
tail -3 $O/pytest.log
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_train -o bench -- python $R/bench.py --mode train --graph off --no-configs --no-cpu-baseline --no-extras --steps 20 --warmup 5 > $O/prof_train.json 2> $O/prof_train.err
python $R/tools/rocprof_summary.py $(find $O/prof_train -name "*results.db" | head -1) $O/r3_kernel_stats_train.md "python bench.py --mode train --graph off --no-configs --no-cpu-baseline --no-extras --steps 20 --warmup 5   (32 pairs per step, eager so that the launches are visible one by one)" last 100
rm -rf $O/prof_train
sed -n 9,16p $O/r3_kernel_stats_train.md | cut -c1-70,110-170
cd $R
timeout 200 python bench.py --mode train --steps 300 > $O/bench_train.json 2> $O/bench_train.err
python -c "
import json; r=json.load(open('$O/bench_train.json')); print('train graph', r['value'], r['ms_per_step'])"
