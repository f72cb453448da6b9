#!/bin/bash
# round 6, run k: full GPU suite + smoke after the race-form sampler, the prune and the bench changes
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6k
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/r6k/pytest.log 2>&1; tail -6 gpurun_out/r6k/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
