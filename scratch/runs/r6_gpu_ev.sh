#!/bin/bash
# round 6: the headline line with the measurement's own event records thinned out (every 4th step) -- before / after on one box
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for ref in HEAD cur; do
    if [ $ref = HEAD ]; then f=/tmp/bench_head.py; cp scratch/bench_head.py $f; cp $f ./bench_head_tmp.py; b=bench_head_tmp.py; else b=bench.py; fi
    timeout 300 python $b --steps 20 --warmup 5 --no-configs --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$ref', round(d['value']/1e6,2), 'M  step', round(d['ms_per_step'],4), 'ms  scoring launch', round(d['roofline']['avg_launch_ms'],4), d['roofline'].get('event_sampling','')[:60])"
  done
done
rm -f bench_head_tmp.py
