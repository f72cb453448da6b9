#!/bin/bash
# round 6, run f: one-pair refit, light (one-sample final stage) vs wave-cooperative final stage, in the drop-in loop with the (2048, 4096) plan
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for pm in 16 1; do for rbs in 1024 64; do
  echo "pair_min=$pm rbs=$rbs $(DRANSAC_REFIT_PAIR_MIN=$pm DROPIN_RBS=$rbs timeout 300 python scratch/dropin_loop.py 2>&1 | grep 'ms per pair')"
done; done; done
