#!/bin/bash
# round 6, run c: the gate test again, round-6 tests, which rounds the drop-in loop's pairs run, the loop per plan
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6c
timeout 600 python -m pytest tests/test_gpu_round5.py tests/test_gpu_round6.py tests/test_plugin_contract.py -m gpu -q 2>&1 | tail -30
python scratch/dropin_iters.py 2>&1 | tail -3
for rbs in 1024 64; do for hy in "1024,4096" "1024,1024" "1024,2048"; do
  echo "rbs=$rbs hyps=$hy"; DROPIN_RBS=$rbs DROPIN_HYPS=$hy timeout 300 python scratch/dropin_loop.py 2>&1 | grep "ms"
done; done | tee gpurun_out/r6c/dropin.log
