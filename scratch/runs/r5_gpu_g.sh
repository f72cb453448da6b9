#!/bin/bash
mkdir -p gpurun_out/r5g
O=$PWD/gpurun_out/r5g
timeout 900 python -m pytest tests/test_gpu_round5.py tests/test_gpu_drivers.py tests/test_gpu_graphs.py tests/test_gpu_round2.py -q --timeout 300 > $O/pytest.log 2>&1; tail -5 $O/pytest.log
python scratch/dropin_loop.py 2>&1 | grep -v amdgpu.ids | tee $O/dropin_loop.log
python scratch/dropin_host.py 2>&1 | grep -v amdgpu.ids | tee $O/dropin_host.log
