#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 300 python bench.py --steps 20 --warmup 5 --no-configs --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value']/1e6,2), round(d['ms_per_step'],4), r['avg_launch_ms'], r['frac'], r['empty_event_pair_ms'], r['frac_net_of_event_pair'])"
