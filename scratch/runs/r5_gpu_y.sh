#!/bin/bash
# round 5: the step with the final refit, per launch (after the refit kernel's final stage became wave-cooperative)
mkdir -p gpurun_out/r5yy
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5yy
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_refit -o refit -- python $R/scratch/refit_step.py > $O/prof_refit.log 2>&1; grep "refit=" $O/prof_refit.log
python $R/tools/rocprof_summary.py $(find $O/prof_refit -name "*results.db" | head -1) $O/ks.md "x" last 100 > /dev/null
cut -c1-40,100-175 $O/ks.md | sed -n 9,17p
python $R/scratch/trace_dump.py $(find $O/prof_refit -name "*results.db" | head -1) 2>/dev/null | tail -30
rm -rf $O/prof_refit
