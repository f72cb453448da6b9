#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 300 python scratch/dropin_iters.py 2>&1 | grep -v amdgpu | tail -3
