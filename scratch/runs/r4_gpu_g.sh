#!/bin/bash
# round 4, pass G: K4r variants at c4; the one-pair call with 16-slot K4 tiles; MSAC tests on the small-grid kernel
mkdir -p gpurun_out; O=$PWD/gpurun_out; L=$O/r4_g.log; : > $L
timeout 600 python -m pytest tests/test_gpu_msac.py tests/test_gpu_edge_cases.py tests/test_gpu_drivers.py tests/test_gpu_round3.py tests/test_gpu_round4.py tests/test_gpu_configs.py -q 2>&1 | tail -3 >> $L
echo "== c4: cur / group 2 / group 8 / tile 18 / tile 64" >> $L
for r in 1 2; do for n in cur rg2 rg8 rt17 rt64; do
  lib=""; [ "$n" != "cur" ] && lib=$PWD/scratch/libdransac_$n.so
  DRANSAC_LIB=$lib timeout 200 python bench.py --workload c4 --steps 300 --warmup 30 --no-configs --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$n', round(d['ms_per_step'],4), 'ms  residual launch', round(d['roofline']['avg_launch_ms'],4), 'ms frac', round(d['roofline']['frac'],3))" >> $L
done; done
echo "== one pair" >> $L
for g in off on; do timeout 120 python bench.py --pairs 1 --steps 300 --warmup 30 --no-configs --no-cpu-baseline --no-extras --graph $g 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('P=1 graph=$g', round(d['ms_per_step'],4), 'ms', round(d['value']/1e6,2), 'M hyps/s  K4', round(d['roofline']['avg_launch_ms'],4))" >> $L; done
for p in 2 4 8; do timeout 120 python bench.py --pairs $p --steps 300 --warmup 30 --no-configs --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('P=$p', round(d['ms_per_step'],4), 'ms', round(d['value']/1e6,2), 'M hyps/s  K4', round(d['roofline']['avg_launch_ms'],4), d['roofline']['kernel'])" >> $L; done
