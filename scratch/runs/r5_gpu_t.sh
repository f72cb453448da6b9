#!/bin/bash
# round 5: K6 update kernel with its correspondences requested before the arg-max -- suite + the driver's command under rocprofv3
mkdir -p gpurun_out/r5t
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5t
cd $R
timeout 600 python -m pytest tests/test_gpu_msac.py tests/test_gpu_drivers.py tests/test_gpu_round3.py -q --timeout 300 -x > $O/pytest.log 2>&1; tail -3 $O/pytest.log
cd /tmp
for n in c2 c2_p1; do
  extra=""; [ $n = c2_p1 ] && extra="--pairs 1 --graph off"
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_$n -o bench -- python $R/bench.py $extra --no-configs --no-cpu-baseline --no-extras --steps 20 --warmup 5 > $O/prof_$n.json 2> $O/prof_$n.err
  python $R/tools/rocprof_summary.py $(find $O/prof_$n -name "*results.db" | head -1) $O/kernel_stats_$n.md "$n" last 100
  rm -rf $O/prof_$n
  cut -c1-150 $O/kernel_stats_$n.md | sed -n 7,16p
done
cd $R
python bench.py --pairs 1 --steps 600 --no-configs --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('one pair replayed', d['ms_per_step'])"
