#!/bin/bash
# round 6, run z: full GPU suite with DR_K3_FAST_DIV (rcp / rsq + Newton in the five-point stages), solver fuzz, headline step
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -4
timeout 600 python scratch/fuzz_r3.py 2>&1 | tail -4
for rep in 1 2; do timeout 300 python bench.py --no-configs --no-cpu-baseline --no-extras --steps 300 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(round(d['value']/1e6,2), round(d['ms_per_step'],4))"; done
