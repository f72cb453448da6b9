#!/bin/bash
cd $GRAFT_REPO_ROOT
for lead in 0 1; do echo "lead=$lead"; DRANSAC_REFIT_LEAD=$lead python scratch/refit_step.py 2>&1 | grep "refit="; done
DRANSAC_REFIT_LEAD=1 bash scratch/runs/r5_gpu_y.sh
