#!/bin/bash
# round 4, pass M: K4r tile sizes around the one-round choice (grid 1525 of 1536 slots leaves no slack for an uneven dispatch)
mkdir -p gpurun_out; O=$PWD/gpurun_out; L=$O/r4_m.log; : > $L
for r in 1 2; do for n in cur rt30 rt36 rt38 rt40 rt44; do
  lib=""; [ "$n" != "cur" ] && lib=$PWD/scratch/libdransac_$n.so
  DRANSAC_LIB=$lib timeout 200 python bench.py --workload c4 --steps 300 --warmup 30 --no-configs --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$n', round(d['ms_per_step'],4), 'ms  residual launch', round(d['roofline']['avg_launch_ms'],4), 'ms frac', round(d['roofline']['frac'],3))" >> $L
done; done
