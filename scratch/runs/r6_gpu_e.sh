#!/bin/bash
# round 6, run e: update kernel with all sub-batch loads in flight; drop-in loop per plan (first device round of 2048 hypotheses and more)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6e
timeout 600 python -m pytest tests/test_gpu_round6.py tests/test_gpu_round5.py -m gpu -q -x 2>&1 | tail -3
for rep in 1 2; do for rbs in 1024 64; do for hy in "1024,1024" "1024,1024,4096" "2048,4096" "2048,1024,4096" "3072,2048" "5120"; do
  echo "rbs=$rbs hyps=$hy $(DROPIN_RBS=$rbs DROPIN_HYPS=$hy timeout 300 python scratch/dropin_loop.py 2>&1 | grep 'ms per pair')"
done; done; done | tee gpurun_out/r6e/dropin.log
