#!/bin/bash
# round 4, pass H: the 16-slot halves of K4 for small grids (runtime tile_slots): suite, one-pair calls, headline step
mkdir -p gpurun_out; O=$PWD/gpurun_out; L=$O/r4_h.log; : > $L
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -3 >> $L
for g in off on; do timeout 120 python bench.py --pairs 1 --steps 300 --warmup 30 --no-configs --no-cpu-baseline --no-extras --graph $g 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('P=1 graph=$g', round(d['ms_per_step'],4), 'ms', round(d['value']/1e6,2), 'M hyps/s  K4', round(d['roofline']['avg_launch_ms'],4))" >> $L; done
for p in 2 3 4 8; do timeout 120 python bench.py --pairs $p --steps 300 --warmup 30 --no-configs --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('P=$p', round(d['ms_per_step'],4), 'ms', round(d['value']/1e6,2), 'M hyps/s  K4', round(d['roofline']['avg_launch_ms'],4))" >> $L; done
bash scratch/ab_step.sh cur >> $L 2>&1
