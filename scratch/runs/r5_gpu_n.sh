#!/bin/bash
mkdir -p gpurun_out/r5n
O=$PWD/gpurun_out/r5n
for sc in 1 0; do
DRANSAC_SCREEN_SHORT=$sc timeout 300 python bench.py --steps 200 --warmup 20 --no-configs --no-cpu-baseline --no-extras --profile-kernels > $O/bench_sc$sc.json 2> $O/bench_sc$sc.err
python -c "
import json; d=json.load(open('$O/bench_sc$sc.json')); print('screen=$sc', round(d['value']/1e6,2), 'M  step', round(d['ms_per_step'],4), d.get('kernel_breakdown_ms'))"
done
