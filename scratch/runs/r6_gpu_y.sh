#!/bin/bash
# round 6, run y: K3 A/B -- divisions / square roots by rcp / rsq + Newton (DR_K3_FAST_DIV) against the IEEE sequences
cd $GRAFT_REPO_ROOT
for rep in 1 2; do timeout 600 python scratch/ab_k3.py r6ieee r6fdiv 2>&1 | grep -v "amdgpu.ids"; done
