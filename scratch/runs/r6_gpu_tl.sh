#!/bin/bash
# round 6: timeline of the replayed c4 step
mkdir -p gpurun_out/tl
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/tl
cd /tmp
timeout 300 rocprofv3 --kernel-trace -d $O/prof -o bench -- python $R/bench.py --workload c4 --graph on --no-configs --no-cpu-baseline --no-extras --steps 20 --warmup 5 > $O/c4.json 2> $O/prof.err
db=$(find $O/prof -name "*results.db" | head -1)
for k in 9 8; do python $R/tools/rocprof_timeline.py $db gumbel_screen_fused_kernel $k | cut -c1-130; done
rm -rf $O/prof
python -c "
import json; d=json.loads([l for l in open('$O/c4.json') if l.startswith('{')][-1]); print(d['ms_per_step'], d['config']['issue'])"
