#!/bin/bash
# round 3, GPU pass P: K4 A/B passes against the tree library ("cur"): first "nopre" (point loads behind the model check), later "barrier" (block barrier at the end instead of the per-half LDS rendezvous)
mkdir -p gpurun_out/r3p
(timeout 300 python -m pytest tests/test_gpu_msac.py tests/test_gpu_round2.py tests/test_gpu_edge_cases.py tests/test_gpu_drivers.py -m gpu -q -x --timeout 300 > gpurun_out/r3p/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3p/pytest.log)
tail -3 gpurun_out/r3p/pytest.log
AB_ARGS="--segments 3 --prewarm-s 0.3" bash scratch/ab_step.sh cur barrier 2>&1 | tee gpurun_out/r3p/ab_tailsync.log
