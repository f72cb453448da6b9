#!/bin/bash
# round 3, GPU pass X: sampler backward in the exponential-race form (tree) against the general form ("bwdold"), train step
mkdir -p gpurun_out/r3x
timeout 600 python -m pytest tests/test_gpu_sampler.py tests/test_gpu_round2.py tests/test_gpu_round3.py tests/test_gpu_drivers.py -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/r3x/tests.log
AB_ARGS="--mode train --segments 3 --prewarm-s 0.3" bash scratch/ab_step.sh bwdold cur 2>&1 | tee gpurun_out/r3x/ab.log
