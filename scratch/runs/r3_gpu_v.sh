#!/bin/bash
# round 3, GPU pass V: K1 register kernel with the LDS-staged candidate scan (tree) against the per-element ballots ("k1old")
mkdir -p gpurun_out/r3v
timeout 600 python -m pytest tests/test_gpu_sampler.py tests/test_gpu_round2.py -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/r3v/tests.log
AB_ARGS="--segments 3 --prewarm-s 0.3" bash scratch/ab_step.sh k1old cur 2>&1 | tee gpurun_out/r3v/ab.log
