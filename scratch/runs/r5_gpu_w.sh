#!/bin/bash
# round 5: the essential refit on a high-priority side stream, issued up front / right before the first scoring launch
cd $GRAFT_REPO_ROOT
for late in 0 1; do for prio in 0 -1; do
  echo "late=$late prio=$prio"; DRANSAC_REFIT_LATE=$late DRANSAC_REFIT_PRIO=$prio python scratch/refit_step.py 2>&1 | grep "refit="
done; done
