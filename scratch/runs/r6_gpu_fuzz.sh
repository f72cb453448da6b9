#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python scratch/fuzz_r6.py 60 2>&1 | grep -v amdgpu | tail -25
timeout 900 python scratch/fuzz_r4.py 20 2>&1 | grep -v amdgpu | tail -3
timeout 900 python scratch/fuzz_r3.py 2>&1 | grep -v amdgpu | tail -3
