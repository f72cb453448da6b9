#!/bin/bash
# round 5, K3 third pass: the tree build (two-phase kernels chosen automatically at >= 65 536 samples) -- solver tests + timing
mkdir -p gpurun_out/r5d
O=gpurun_out/r5d
timeout 900 python -m pytest tests/test_gpu_round5.py tests/test_gpu_roots.py tests/test_gpu_solvers.py tests/test_gpu_round4.py tests/test_gpu_round3.py tests/test_gpu_configs.py tests/test_gpu_drivers.py -q --timeout 300 > $O/pytest.log 2>&1; tail -4 $O/pytest.log
timeout 200 python scratch/r5_k3_time.py 131072 65536 32768 2>&1 | grep -v amdgpu.ids | tee $O/k3_time.log
bash scratch/ab_step.sh cur 2>&1 | tee $O/step.log
