#!/bin/bash
# (the DRANSAC_CPU_PIN knob existed in bench.py for this measurement only: pinning was worse and was removed again -- docs/LOG.md, round-6 log item 23)
# round 6: does pinning the CPU legs' threads make the c2 CPU baseline repeat?  alternating runs, one box
cd $GRAFT_REPO_ROOT
for rep in 1 2 3 4; do
  for pin in 1 0; do
    DRANSAC_CPU_PIN=$pin timeout 300 python bench.py --steps 5 --warmup 2 --no-configs --no-extras 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); c = d['cpu_baseline']
print('pin=$pin', round(c['value']), c['threads'], c['spread'], c['by_threads'], c.get('fastest_unit_by_threads'))"
  done
done
