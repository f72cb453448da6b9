#!/bin/bash
# round 3, GPU pass I: K4 parity tests on the tree library (v_bitop3 mask op + LDS quad-sum reduction), then in-step A/B:
# cur = both, bitop = the mask op only, base = neither
mkdir -p gpurun_out/r3i
(timeout 300 python -m pytest tests/test_gpu_msac.py tests/test_gpu_round2.py tests/test_gpu_edge_cases.py tests/test_gpu_drivers.py tests/test_gpu_configs.py -m gpu -q -x --timeout 300 > gpurun_out/r3i/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3i/pytest.log)
tail -4 gpurun_out/r3i/pytest.log
AB_ARGS="--segments 3 --prewarm-s 0.3" bash scratch/ab_step.sh cur base bitop 2>&1 | tee gpurun_out/r3i/ab_reduction.log
