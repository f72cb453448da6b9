#!/bin/bash
# round 6, run m: full GPU suite + smoke after the entry-point collapse (97 -> 78 exported names)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6m
timeout 1200 python -m pytest tests -m gpu -q -x > gpurun_out/r6m/pytest.log 2>&1; tail -6 gpurun_out/r6m/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 300 python bench.py --steps 20 --warmup 5 --no-configs --no-cpu-baseline > gpurun_out/r6m/bench_short.json 2>gpurun_out/r6m/bench_short.err; head -c 400 gpurun_out/r6m/bench_short.json
