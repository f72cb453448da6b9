#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python scratch/fuzz_r6.py 40 2>&1 | grep -v amdgpu | tail -25
