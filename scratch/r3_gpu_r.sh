#!/bin/bash
# round 3, GPU pass R: sampler backward with 4096 / 1024 blocks (16 / 64 rows each) against the tree's 2048 x 32, in the train step
mkdir -p gpurun_out/r3r
for round in 1 2; do
  for n in cur bwd4k bwd1k; do
    lib=""; [ "$n" != "cur" ] && lib=$PWD/scratch/libdransac_$n.so
    DRANSAC_LIB=$lib timeout 200 python bench.py --mode train --graph off --steps 300 --warmup 10 --segments 3 --prewarm-s 0.3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$n', round(d['value']/1e6,2), 'M  step', round(d['ms_per_step'],4), 'ms')"
  done
done 2>&1 | tee gpurun_out/r3r/ab_bwd_blocks.log
