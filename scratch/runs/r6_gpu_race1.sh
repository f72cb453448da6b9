#!/bin/bash
# round 6: does the one-logarithm form (weights out of the set-up launch) pay for one-pair calls now that its selection is cheaper?
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for mn in "32768,32" "1,1"; do
    for rbs in 1024 64; do
      echo "race_min=$mn rbs=$rbs $(DRANSAC_K1_RACE_MIN=$mn DROPIN_RBS=$rbs timeout 300 python scratch/dropin_loop.py 2>&1 | grep 'ms per pair')"
    done
  done
done
for mn in "32768,32" "1,1" "32768,32" "1,1"; do
  for pairs in 1 8 32; do
    DRANSAC_K1_RACE_MIN=$mn timeout 200 python bench.py --pairs $pairs --steps 300 --warmup 30 --no-configs --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('race_min=$mn pairs=$pairs', round(d['value']/1e6,2), 'M  step', round(d['ms_per_step'],4), 'ms')"
  done
done
