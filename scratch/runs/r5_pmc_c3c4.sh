#!/bin/bash
# round 5: FETCH_SIZE / WRITE_SIZE of the other configs' dominant kernels (c3: scoring at 32 x 2000 x 4096; c4: rigid residuals at
# 50 000 x 2048), separate passes per counter, --kernel-trace only
mkdir -p gpurun_out/r5pmc
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5pmc
cd /tmp
for w in c3 c4; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 200 rocprofv3 --pmc $c --kernel-trace -d $O/pmc_${w}_$c -o bench -- python $R/bench.py --workload $w --steps 12 --warmup 3 --segments 1 --prewarm-s 0.05 --no-configs --no-cpu-baseline --no-extras > /dev/null 2> $O/pmc_${w}_$c.err
  done
  python $R/tools/rocprof_pmc_summary.py $O/r5_pmc_fetch_write_$w.md $O/r5_pmc_fetch_write_$w.json $(find $O/pmc_${w}_FETCH_SIZE $O/pmc_${w}_WRITE_SIZE -name "*results.db")
  rm -rf $O/pmc_${w}_FETCH_SIZE $O/pmc_${w}_WRITE_SIZE
  cat $O/r5_pmc_fetch_write_$w.md
done
