#!/bin/bash
# round 6, run o: same-box A/B of the folded set-up (seeds out of dr_ransac_init, packed one-pair state) in the drop-in loop
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for fold in 1 0; do for rbs in 1024 64; do
  echo "fold=$fold rbs=$rbs $(DRANSAC_FOLD_SETUP=$fold DROPIN_RBS=$rbs timeout 300 python scratch/dropin_loop.py 2>&1 | grep 'ms per pair')"
done; done; done
