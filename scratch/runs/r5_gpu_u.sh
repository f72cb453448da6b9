#!/bin/bash
# round 5: what the f64 pow / log10 tail of the K6 update kernel costs (timing build -DDR_K6_NOPOW)
mkdir -p gpurun_out/r5u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5u
cd /tmp
for lib in "" $R/scratch/libdransac_k6nopow.so; do
  for extra in "--pairs 1 --graph off" ""; do
  DRANSAC_LIB=$lib timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python $R/bench.py $extra --no-configs --no-cpu-baseline --no-extras --steps 20 --warmup 5 > $O/prof.json 2> $O/prof.err
  python $R/tools/rocprof_summary.py $(find $O/prof -name "*results.db" | head -1) $O/ks.md "x" last 100 > /dev/null
  rm -rf $O/prof
  echo "lib=[$lib] extra=[$extra]"; grep ransac_update $O/ks.md | cut -c1-150
  done
done
