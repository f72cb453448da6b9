#!/bin/bash
# round 6, run w: 3x3 Jacobi rotations by rsq / rcp + Newton (rigid solver, pose-error SVD, rigid backward): tests + config 4
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
for rep in 1 2 3; do timeout 300 python bench.py --workload c4 --no-configs --no-cpu-baseline --no-extras --steps 300 --profile-kernels 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('c4', round(d['ms_per_step'],4), d.get('launch_ms') or d.get('kernel_ms') or '')"; done
