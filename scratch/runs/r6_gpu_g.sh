#!/bin/bash
# round 6, run g: K3 A/B -- hand-scheduled Sturm evaluation (asm), Estrin refine tasks, both; two passes for box noise
cd $GRAFT_REPO_ROOT
for rep in 1 2; do timeout 600 python scratch/ab_k3.py r6base r6asm r6estrin r6both 2>&1 | grep -v "amdgpu.ids"; done
