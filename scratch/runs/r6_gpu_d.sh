#!/bin/bash
# round 6, run d: drop-in loop per plan after the arg-max fix of dr_ransac_update; kernel timeline of one replayed call
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6d
for rbs in 1024 64; do for hy in "1024,4096" "1024,1024" "1024,2048" "1024,1024,4096"; do
  echo "rbs=$rbs hyps=$hy"; DROPIN_RBS=$rbs DROPIN_HYPS=$hy timeout 300 python scratch/dropin_loop.py 2>&1 | grep "ms"
done; done | tee gpurun_out/r6d/dropin.log
export TMPDIR=/tmp
for rbs in 1024 64; do
  DROPIN_RBS=$rbs DROPIN_HYPS=1024,2048 DROPIN_PASSES=3 timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/r6d/prof_$rbs -o dropin -- python scratch/dropin_loop.py > gpurun_out/r6d/prof_$rbs.log 2>&1
  db=$(find gpurun_out/r6d/prof_$rbs -name "*.db" | head -1)
  python tools/rocprof_timeline.py $db ransac_init_kernel 3 > gpurun_out/r6d/timeline_$rbs.md 2>&1
  python tools/rocprof_timeline.py $db ransac_init_kernel 5 > gpurun_out/r6d/timeline_${rbs}_b.md 2>&1
  python tools/rocprof_summary.py $db gpurun_out/r6d/stats_$rbs.md "dropin_loop rbs=$rbs" last 200 > /dev/null 2>&1
  sqlite3 $db ".schema kernels" > gpurun_out/r6d/schema.txt 2>&1 || python -c "
import sqlite3,sys; c=sqlite3.connect('$db'); print(list(c.execute('pragma table_info(kernels)')))" > gpurun_out/r6d/schema.txt 2>&1
  rm -rf gpurun_out/r6d/prof_$rbs
done
cat gpurun_out/r6d/timeline_1024.md
