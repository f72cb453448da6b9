#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python scratch/ab_k6.py 2>&1 | grep -v amdgpu | tail -12
