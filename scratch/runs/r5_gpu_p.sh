#!/bin/bash
mkdir -p gpurun_out/r5p
O=$PWD/gpurun_out/r5p
timeout 600 python -m pytest tests/test_gpu_roots.py tests/test_gpu_solvers.py tests/test_gpu_round5.py -q --timeout 300 > $O/pytest.log 2>&1; tail -6 $O/pytest.log
for v in "" k3nofb; do
lib=""; [ -n "$v" ] && lib=$PWD/scratch/libdransac_$v.so
DRANSAC_LIB=$lib DRANSAC_SCREEN_SHORT=0 timeout 300 python bench.py --steps 200 --warmup 20 --no-configs --no-cpu-baseline --no-extras --profile-kernels > $O/bench_$v.json 2> $O/bench_$v.err
python -c "
import json; d=json.load(open('$O/bench_$v.json')); print('lib=${v:-tree}', round(d['value']/1e6,2), 'M  step', round(d['ms_per_step'],4), 'K3', d['kernel_breakdown_ms']['K3_solver'])"
done
K3_PAIRS=128 K3_PATH=2 python scratch/prof_stages.py 2>&1 | tail -1
