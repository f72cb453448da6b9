#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_round6.py -m gpu -x -q -k batched_forward 2>&1 | grep -v "^$" | tail -30
for rep in 1 2 3; do for r in 0 1; do
  echo "race=$r $(DRANSAC_K1_RACE=$r timeout 300 python bench.py --no-configs --no-cpu-baseline --no-extras --steps 300 --profile-kernels 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print(round(d['value']/1e6,2), round(d['ms_per_step'],4), {k:round(v,4) for k,v in (d.get('kernel_ms') or d.get('profile_kernels') or {}).items()} if isinstance(d.get('kernel_ms') or d.get('profile_kernels'),dict) else '')
")"
done; done
