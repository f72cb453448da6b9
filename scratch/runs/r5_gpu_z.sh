#!/bin/bash
# round 5: per-pair drop-in loop, refit kernel with the one-sample final stage (scratch/libdransac_refitold.so) vs the wave-cooperative one
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do for lib in "" $PWD/scratch/libdransac_refitold.so; do
  echo "lib=[$(basename "$lib")]"; DRANSAC_LIB=$lib python scratch/dropin_loop.py 2>&1 | grep "ms per pair"
done; done
