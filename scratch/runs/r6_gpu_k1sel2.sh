#!/bin/bash
# round 6: selection on wave masks -- in-step A/B (headline, c3) and the whole GPU suite
mkdir -p gpurun_out/k1sel
cd $GRAFT_REPO_ROOT
bash scratch/ab_step.sh cur k1_oldsel 2>&1 | grep -v amdgpu.ids
AB_ARGS="--workload c3" bash scratch/ab_step.sh cur k1_oldsel 2>&1 | grep -v amdgpu.ids
timeout 1500 python -m pytest tests -m gpu -q --timeout 300 > gpurun_out/k1sel/pytest.log 2>&1; tail -3 gpurun_out/k1sel/pytest.log
