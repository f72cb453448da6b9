#!/bin/bash
# round 4, pass A: suite on the pruned tree; K4 with the points streamed from LDS (lds16<RES>) against fast16, in the step
mkdir -p gpurun_out; L=gpurun_out/r4_a.log; : > $L
echo "== suite" >> $L
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 >> $L
for v in L8 L4 L0 L0h8 L0sc L0m; do
  echo "== equal cur $v" >> $L
  timeout 120 python scratch/k4_equal.py cur $v >> $L 2>&1
  DRANSAC_LIB=$PWD/scratch/libdransac_$v.so timeout 300 python -m pytest tests/test_gpu_msac.py -x -q 2>&1 | tail -2 >> $L
done
echo "== in-step A/B" >> $L
bash scratch/ab_step.sh cur L8 L4 L0 L0h8 L0sc L0m >> $L 2>&1
