#!/bin/bash
# round 6: the sampler's selection on wave masks -- identity with the list selection, timing, then the sampler tests
mkdir -p gpurun_out/k1sel
cd $GRAFT_REPO_ROOT
timeout 900 python scratch/k1_select_check.py > gpurun_out/k1sel/check.txt 2>&1; echo "check rc $?"
grep -v amdgpu.ids gpurun_out/k1sel/check.txt | grep "us$\|ALL\|DIFFER"
rm -f gpurun_out/k1sel/*.pt
timeout 900 python -m pytest tests/test_gpu_sampler.py tests/test_gpu_round6.py tests/test_gpu_round2.py -m gpu -q -x --timeout 300 2>&1 | tail -3
