#!/bin/bash
# round 3, GPU pass E: K4 parity on the tree library (asm prefetch awaited at the end of the iteration + real SGPR pairs), then
# in-step A/B against the round-2 form ("base")
mkdir -p gpurun_out/r3e
(timeout 300 python -m pytest tests/test_gpu_msac.py tests/test_gpu_round2.py tests/test_gpu_edge_cases.py tests/test_gpu_drivers.py -m gpu -q -x --timeout 300 > gpurun_out/r3e/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3e/pytest.log)
tail -4 gpurun_out/r3e/pytest.log
AB_ARGS="--segments 3 --prewarm-s 0.3" bash scratch/ab_step.sh cur base 2>&1 | tee gpurun_out/r3e/ab_model_fetch2.log
