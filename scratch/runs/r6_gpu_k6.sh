#!/bin/bash
# round 6: the update launch's arg-max scan with 8 / 16 scores in flight per thread (4 in the tree): config 3 and headline steps
cd $GRAFT_REPO_ROOT
AB_ARGS="--workload c3" bash scratch/ab_step.sh cur k6f8 k6f16 2>&1 | grep -v amdgpu.ids
bash scratch/ab_step.sh cur k6f8 k6f16 2>&1 | grep -v amdgpu.ids
