#!/bin/bash
mkdir -p gpurun_out/r5yy
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5yy
cd /tmp
for sp in 0 1 2; do
echo "spacer=$sp"; DRANSAC_REFIT_SPACER=$sp python $R/scratch/refit_step.py 2>&1 | grep "refit=True"
done
for sp in 1 2; do
DRANSAC_REFIT_SPACER=$sp timeout 300 rocprofv3 --kernel-trace -d $O/prof_refit -o refit -- python $R/scratch/refit_step.py > $O/prof_refit.log 2>&1; echo "spacer=$sp"; grep "refit=" $O/prof_refit.log
python $R/scratch/trace_dump2.py $(find $O/prof_refit -name "*results.db" | head -1) 2>&1 | tail -12
rm -rf $O/prof_refit
done
