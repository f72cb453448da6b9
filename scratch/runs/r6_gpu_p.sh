#!/bin/bash
# round 6, run p: weights of the one-logarithm sampler out of dr_ransac_init; update kernel without the second barrier: tests, headline A/B
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
for rep in 1 2 3; do for f in 1 0; do
  echo "fold=$f $(DRANSAC_FOLD_SETUP=$f timeout 300 python bench.py --no-configs --no-cpu-baseline --no-extras --steps 300 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(round(d['value']/1e6,2), round(d['ms_per_step'],4), round(d['roofline']['avg_launch_ms'],4))")"
done; done
for rbs in 1024 64; do echo "dropin rbs=$rbs $(DROPIN_RBS=$rbs timeout 300 python scratch/dropin_loop.py 2>&1 | grep 'ms per pair')"; done
