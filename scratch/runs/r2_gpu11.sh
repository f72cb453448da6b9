#!/bin/bash
# PMC passes (FETCH_SIZE, WRITE_SIZE) of the default workload after a change to the scoring kernel's source
mkdir -p gpurun_out/r2p
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2p
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace -d $O/pmc_$c -o bench -- python $R/bench.py --steps 12 --warmup 3 --no-configs --no-cpu-baseline --no-extras > /dev/null 2> $O/pmc_$c.err
done
python $R/tools/rocprof_pmc_summary.py $O/r2_pmc_fetch_write.md $O/r2_pmc_fetch_write.json --pairs 128 --points 2000 --hyps 1024 $(find $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE -name "*results.db")
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
cat $O/r2_pmc_fetch_write.md | head -20
