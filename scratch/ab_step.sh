#!/bin/bash
# in-step A/B of library builds: bench.py with DRANSAC_LIB pointing at scratch/libdransac_<name>.so ("cur" = the tree's library), two rounds
for round in 1 2; do
  for n in "$@"; do
    lib=""; [ "$n" != "cur" ] && lib=$PWD/scratch/libdransac_$n.so
    DRANSAC_LIB=$lib timeout 200 python bench.py --steps 300 --warmup 30 --no-configs --no-cpu-baseline --no-extras ${AB_ARGS} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$n', round(d['value']/1e6,2), 'M  step', round(d['ms_per_step'],4), 'ms  scoring launch', round((d.get('roofline') or {}).get('avg_launch_ms',0),4), 'ms')"
  done
done
