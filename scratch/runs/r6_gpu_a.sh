#!/bin/bash
# round 6, run a: full GPU suite on the round's first changes (gate fix, super-rounds), drop-in loop at -rbs 1024 / 64, default bench
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6a
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r6a/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r6a/pytest.log
tail -5 gpurun_out/r6a/pytest.log
for rbs in 1024 64; do for hy in "1024,4096" "1024,1024"; do
  echo "rbs=$rbs hyps=$hy"; DROPIN_RBS=$rbs DROPIN_HYPS=$hy timeout 300 python scratch/dropin_loop.py 2>&1 | grep "ms"
done; done | tee gpurun_out/r6a/dropin.log
timeout 600 python bench.py > gpurun_out/r6a/bench.json 2> gpurun_out/r6a/bench.err; tail -c 600 gpurun_out/r6a/bench.json
