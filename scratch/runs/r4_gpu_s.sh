#!/bin/bash
mkdir -p gpurun_out; O=$PWD/gpurun_out; L=$O/r4_s.log; : > $L
timeout 600 python -m pytest tests/test_gpu_round4.py tests/test_gpu_configs.py tests/test_gpu_round2.py tests/test_gpu_round3.py tests/test_gpu_graphs.py tests/test_gpu_drivers.py -q 2>&1 | tail -3 >> $L
for r in 1 2; do timeout 200 python bench.py --workload c4 --steps 300 --warmup 30 --no-configs --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c4', round(d['ms_per_step'],4), 'ms  residual launch', round(d['roofline']['avg_launch_ms'],4), 'ms frac', round(d['roofline']['frac'],3))" >> $L; done
timeout 200 python bench.py --workload c4 --steps 300 --warmup 30 --no-configs --no-cpu-baseline --no-extras --graph on 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c4 graph', round(d['ms_per_step'],4))" >> $L
