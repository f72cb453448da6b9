#!/bin/bash
# round 3, GPU pass W: K6 (ransac_update) with its loads requested up front (tree) against the old kernel ("updold"); driver tests
mkdir -p gpurun_out/r3w
timeout 600 python -m pytest tests/test_gpu_drivers.py tests/test_gpu_msac.py tests/test_gpu_graphs.py -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/r3w/tests.log
AB_ARGS="--segments 3 --prewarm-s 0.3" bash scratch/ab_step.sh updold cur 2>&1 | tee gpurun_out/r3w/ab.log
