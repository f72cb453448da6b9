#!/bin/bash
# round 3, GPU pass J: in-step A/B of the hand-placed model prefetch WITHOUT the SGPR-pair copies (pref1: request pinned at the
# top of the iteration; pref2: placement left to the scheduler) against the tree library ("cur": compiler-placed loads + wait)
mkdir -p gpurun_out/r3j
DRANSAC_LIB=$PWD/scratch/libdransac_pref1.so timeout 200 python -m pytest tests/test_gpu_msac.py -m gpu -q -x > gpurun_out/r3j/pytest_pref1.log 2>&1; tail -2 gpurun_out/r3j/pytest_pref1.log
AB_ARGS="--segments 3 --prewarm-s 0.3" bash scratch/ab_step.sh cur pref1 pref2 2>&1 | tee gpurun_out/r3j/ab_prefetch.log
