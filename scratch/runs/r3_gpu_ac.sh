#!/bin/bash
# round 3, GPU pass AC: K1 with the screened pass (tree) against the full pass ("noscreen")
mkdir -p gpurun_out/r3ac
timeout 600 python -m pytest tests/test_gpu_round3.py tests/test_gpu_sampler.py tests/test_gpu_round2.py tests/test_gpu_graphs.py -m gpu -x -q 2>&1 | tail -6 | tee gpurun_out/r3ac/tests.log
AB_ARGS="--segments 3 --prewarm-s 0.3" timeout 600 bash scratch/ab_step.sh noscreen cur 2>&1 | tee gpurun_out/r3ac/ab.log
