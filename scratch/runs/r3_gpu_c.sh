#!/bin/bash
# round 3, GPU pass C: full GPU test suite, then in-step A/B of the contiguous zero-run fill of K4 ("cur" = tree library,
# "norun" = -DDR_K4_ZERO_RUNS=0), two rounds each, then the headline at the driver's settings
mkdir -p gpurun_out/r3c
(timeout 600 python -m pytest tests -m gpu -q -x --timeout 300 > gpurun_out/r3c/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3c/pytest.log)
tail -4 gpurun_out/r3c/pytest.log
AB_ARGS="--segments 3 --prewarm-s 0.3" bash scratch/ab_step.sh cur norun 2>&1 | tee gpurun_out/r3c/ab_zero_runs.log
