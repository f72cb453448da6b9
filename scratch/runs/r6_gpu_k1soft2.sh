#!/bin/bash
# round 6: the one-logarithm form in train mode: tests, then train-step A/B (DRANSAC_K1_RACE_SOFT=0/1)
mkdir -p gpurun_out/k1sel
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 300 > gpurun_out/k1sel/pytest.log 2>&1; tail -15 gpurun_out/k1sel/pytest.log
for rep in 1 2; do for on in 1 0; do
  DRANSAC_K1_RACE_SOFT=$on timeout 200 python bench.py --mode train --steps 300 --warmup 30 --no-configs --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('race_soft=$on', round(d['value']/1e6,2), 'M  step', round(d['ms_per_step'],4), 'ms')"
done; done
