#!/bin/bash
# round 6, run x: same-box A/B of the 3x3 Jacobi rotation (rsq / rcp + Newton vs IEEE divisions and square roots) in config 4
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for lib in "" $PWD/scratch/libdransac_jac_ieee.so; do
  DRANSAC_LIB=$lib timeout 300 python bench.py --workload c4 --no-configs --no-cpu-baseline --no-extras --steps 300 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('lib=[$(basename "$lib")] c4', round(d['ms_per_step'],4))"
done; done
export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r6x/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --workload c4 --no-configs --no-cpu-baseline --no-extras --steps 20 --warmup 5 > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $(find $GRAFT_REPO_ROOT/gpurun_out/r6x/prof -name "*results.db" | head -1) $GRAFT_REPO_ROOT/gpurun_out/r6x/r6_kernel_stats_c4.md "python bench.py --workload c4 --no-configs --no-cpu-baseline --no-extras --steps 20 --warmup 5" last 100 > /dev/null
rm -rf $GRAFT_REPO_ROOT/gpurun_out/r6x/prof
sed -n 9,14p $GRAFT_REPO_ROOT/gpurun_out/r6x/r6_kernel_stats_c4.md | cut -c1-60,150-230
