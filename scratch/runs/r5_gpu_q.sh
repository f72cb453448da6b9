#!/bin/bash
mkdir -p gpurun_out/r5q
O=$PWD/gpurun_out/r5q
timeout 600 python -m pytest tests/test_gpu_roots.py tests/test_gpu_round4.py tests/test_gpu_round5.py -q --timeout 300 > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 200 python scratch/r5_k3_time.py 131072 1024 4096 2>&1 | grep -v amdgpu.ids | tee $O/k3_time.log
timeout 300 python bench.py --pairs 1 --steps 600 --warmup 20 --no-configs --no-cpu-baseline --no-extras > $O/bench_p1.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/bench_p1.json')); print('one pair per call:', d['ms_per_step'], d['config'])"
python scratch/dropin_loop.py 2>&1 | grep -v amdgpu.ids | tee $O/dropin_loop.log
