#!/bin/bash
# round 5: rigid residual kernel with 8 points per lane (76 registers, six waves per SIMD) against 16 (138, three): tests with the
# variant library, config 4 in the step, the launch alone under rocprofv3
mkdir -p gpurun_out/r5ab
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5ab
cd $R
DRANSAC_LIB=$R/scratch/libdransac_k4r8.so timeout 600 python -m pytest tests/test_gpu_solvers.py tests/test_gpu_round3.py tests/test_gpu_round4.py tests/test_gpu_configs.py -q --timeout 300 -x -k "rigid or residual or 3d or c4 or config" 2>&1 | tail -3
AB_ARGS="--workload c4" bash scratch/ab_step.sh cur k4r8
cd /tmp
for lib in "" $R/scratch/libdransac_k4r8.so; do
  DRANSAC_LIB=$lib timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python $R/bench.py --workload c4 --no-configs --no-cpu-baseline --no-extras --steps 20 --warmup 5 > $O/prof.json 2> $O/prof.err
  python $R/tools/rocprof_summary.py $(find $O/prof -name "*results.db" | head -1) $O/ks.md "x" last 100 > /dev/null
  rm -rf $O/prof
  echo "lib=[$(basename "$lib")]"; grep rigid_residual $O/ks.md | cut -c1-60,100-200
done
