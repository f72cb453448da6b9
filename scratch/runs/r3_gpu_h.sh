#!/bin/bash
# round 3, GPU pass H: rocprofv3 kernel stats of every BASELINE config + the train step on the TIMED segments of bench.py
# (default pre-conditioning; the summary takes the last 100 dispatches of every kernel = 5 segments x 20 steps)
mkdir -p gpurun_out/r3h
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3h
cd /tmp
run() {  # name  title  bench args...
  name=$1; title=$2; shift 2
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_$name -o bench -- python $R/bench.py "$@" > $O/prof_$name.json 2> $O/prof_$name.err
  python $R/tools/rocprof_summary.py $(find $O/prof_$name -name "*results.db" | head -1) $O/r3_kernel_stats_$name.md "$title" last 100
  rm -rf $O/prof_$name
}
COMMON="--no-configs --no-cpu-baseline --no-extras --steps 20 --warmup 5"
run c2 "python bench.py --steps 20 --warmup 5 --no-configs --no-cpu-baseline --no-extras   (the driver's command without the sub-records; 128 pairs per step)" $COMMON
run c2_p32 "python bench.py --pairs 32 $COMMON" --pairs 32 $COMMON
run c1 "python bench.py --workload c1 --graph off $COMMON" --workload c1 --graph off $COMMON
run c3 "python bench.py --workload c3 $COMMON" --workload c3 $COMMON
run c4 "python bench.py --workload c4 $COMMON" --workload c4 $COMMON
run train "python bench.py --mode train --graph off $COMMON   (32 pairs per step, eager so that the launches are visible one by one)" --mode train --graph off $COMMON
for n in c2 c2_p32 c3 c4; do python -c "
import json,sys; r=json.load(open('$O/prof_$n.json')); print('$n', round(r['value']/1e6,2), round(r['ms_per_step'],4), r['roofline']['kernel'], round(r['roofline']['avg_launch_ms'],4), round(r['roofline']['frac'],4))"; done
head -12 $O/r3_kernel_stats_c2.md | cut -c1-200
