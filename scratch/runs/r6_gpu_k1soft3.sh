#!/bin/bash
# round 6: kernel times of the train step with the one-logarithm form on / off, then step A/B
mkdir -p gpurun_out/k1sel
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/k1sel
cd /tmp
for on in 1; do
  DRANSAC_K1_RACE_SOFT=$on timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_soft_$on -o bench -- python $R/bench.py --mode train --graph off --no-configs --no-cpu-baseline --no-extras --steps 20 --warmup 5 > /dev/null 2> $O/prof_soft_$on.err
  python $R/tools/rocprof_summary.py $(find $O/prof_soft_$on -name "*results.db" | head -1) $O/train_stats_soft_$on.md "train, race_soft=$on" last 100
  rm -rf $O/prof_soft_$on
  grep "race_weights\|topk_fast" $O/train_stats_soft_$on.md | cut -c1-150
done
cd $R
timeout 600 python -m pytest tests/test_gpu_round6.py tests/test_gpu_sampler.py -m gpu -q -x --timeout 300 2>&1 | tail -2
for rep in 1 2 3; do for on in 1 0; do
  DRANSAC_K1_RACE_SOFT=$on timeout 200 python bench.py --mode train --steps 300 --warmup 30 --no-configs --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('race_soft=$on', round(d['value']/1e6,2), 'M  step', round(d['ms_per_step'],4), 'ms')"
done; done
