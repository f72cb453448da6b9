#!/bin/bash
# round 6: timeline of replayed / eager headline steps (gaps between launches) -- four consecutive steps
mkdir -p gpurun_out/tl
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/tl
cd /tmp
timeout 300 rocprofv3 --kernel-trace -d $O/prof -o bench -- python $R/bench.py --no-configs --no-cpu-baseline --no-extras --steps 20 --warmup 5 > /dev/null 2> $O/prof.err
db=$(find $O/prof -name "*results.db" | head -1)
for k in 9 8 7 6; do python $R/tools/rocprof_timeline.py $db ransac_init_kernel $k | cut -c1-130; done > $O/r6_headline_timeline.md
rm -rf $O/prof
cat $O/r6_headline_timeline.md
