#!/bin/bash
# round 6: device-round plans of the replayed drop-in call (default 2048,4096)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for h in "2048,4096" "5120" "3072,2048" "2048,3072" "4096,1024" "2048,1024,2048"; do
  echo "hyps=$h rbs=1024 $(DROPIN_HYPS=$h DROPIN_RBS=1024 timeout 300 python scratch/dropin_loop.py 2>&1 | grep 'ms per pair')"
done; done
