#!/bin/bash
# round 6: the sampler backward's buffer cleared by the forward launch: tests + train-step A/B
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 300 2>&1 | tail -3
for rep in 1 2 3; do for on in 1 0; do
  DRANSAC_ZERO_GRAD_IN_FORWARD=$on timeout 200 python bench.py --mode train --steps 300 --warmup 30 --no-configs --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('zero_in_forward=$on', round(d['value']/1e6,2), 'M  step', round(d['ms_per_step'],4), 'ms')"
done; done
