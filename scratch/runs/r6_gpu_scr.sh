#!/bin/bash
# round 6: the fused screening launch with sixteen loads in flight: c4 in-step A/B + the long-row sampler tests
cd $GRAFT_REPO_ROOT
AB_ARGS="--workload c4" bash scratch/ab_step.sh cur scr_old 2>&1 | grep -v amdgpu.ids
timeout 900 python -m pytest tests/test_gpu_round4.py tests/test_gpu_round5.py tests/test_gpu_configs.py tests/test_gpu_round3.py -m gpu -q --timeout 300 2>&1 | tail -2
