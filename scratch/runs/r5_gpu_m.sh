#!/bin/bash
mkdir -p gpurun_out/r5m
O=$PWD/gpurun_out/r5m
timeout 600 python -m pytest tests/test_gpu_round5.py tests/test_gpu_sampler.py tests/test_gpu_round2.py tests/test_gpu_round4.py -q --timeout 300 > $O/pytest.log 2>&1; tail -8 $O/pytest.log
python scratch/ab_k1_screen.py 2>&1 | grep -v amdgpu.ids | tee $O/k1_screen.log
bash scratch/ab_step.sh cur 2>&1 | tee $O/step.log
