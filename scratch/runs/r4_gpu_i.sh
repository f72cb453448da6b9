#!/bin/bash
# round 4, pass I: sampler prefetch (informational sub-record) next to the headline and two batches in flight
mkdir -p gpurun_out; O=$PWD/gpurun_out; L=$O/r4_i.log; : > $L
timeout 300 python -m pytest tests/test_gpu_round4.py -x -q 2>&1 | tail -5 >> $L
for r in 1 2; do
timeout 200 python bench.py --steps 300 --warmup 30 --no-configs --no-cpu-baseline 2>$O/r4_i.err | python -c "
import sys,json; d=json.loads(sys.stdin.read())
print('headline', round(d['value']/1e6,2), round(d['ms_per_step'],4), '| two batches', round(d['two_batches_in_flight']['value']/1e6,2), round(d['two_batches_in_flight']['ms_per_step'],4), '| prefetch', round(d['sampler_prefetch']['value']/1e6,2), round(d['sampler_prefetch']['ms_per_step'],4), '| topdown', round(d['sampler_topdown']['value']/1e6,2))" >> $L 2>&1
done
tail -3 $O/r4_i.err >> $L
