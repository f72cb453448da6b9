#!/bin/bash
# round 6, run l: standalone five-point root search at one vs two waves per SIMD (review item 1(b): would a separate launch pay?)
cd $GRAFT_REPO_ROOT
python scratch/roots_occupancy.py 2>&1 | grep -v amdgpu.ids
