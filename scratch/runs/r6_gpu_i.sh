#!/bin/bash
# round 6, run i: K1 one-logarithm form -- index-set agreement and time; sampler / driver tests
cd $GRAFT_REPO_ROOT
python scratch/k1_race_check.py 2>&1 | grep -v amdgpu.ids
timeout 900 python -m pytest tests/test_gpu_sampler.py tests/test_gpu_round2.py tests/test_gpu_round3.py tests/test_gpu_round5.py tests/test_gpu_round6.py tests/test_gpu_drivers.py -m gpu -x -q 2>&1 | tail -5
