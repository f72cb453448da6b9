#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the default workload on the CURRENT sources (bench.py accepts the capture only when the sha256 of
# msac_score.hip / msac_filter.hip / dr_common.hpp recorded in it match the tree): re-run after any edit of those files
mkdir -p gpurun_out/r3pmc
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3pmc
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $c --kernel-trace -d $O/pmc_$c -o bench -- python $R/bench.py --steps 12 --warmup 3 --segments 1 --prewarm-s 0.05 --no-configs --no-cpu-baseline --no-extras > /dev/null 2> $O/pmc_$c.err
done
python $R/tools/rocprof_pmc_summary.py $O/r3_pmc_fetch_write.md $O/r3_pmc_fetch_write.json --pairs 128 --points 2000 --hyps 1024 $(find $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE -name "*results.db")
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
head -8 $O/r3_pmc_fetch_write.md
