#!/bin/bash
# round-2 GPU call 1: filter-kernel check + timing, the scoring tests, a short bench
mkdir -p gpurun_out/r2a
export TMPDIR=/tmp
timeout 600 python scratch/k4f_check.py > gpurun_out/r2a/k4f_check.log 2>&1; echo "k4f_check rc $?" >> gpurun_out/r2a/k4f_check.log
tail -60 gpurun_out/r2a/k4f_check.log
timeout 600 python -m pytest tests/test_gpu_msac.py tests/test_gpu_drivers.py tests/test_gpu_configs.py -x -q > gpurun_out/r2a/pytest.log 2>&1; tail -15 gpurun_out/r2a/pytest.log
timeout 300 python bench.py --steps 200 --warmup 20 > gpurun_out/r2a/bench.json 2> gpurun_out/r2a/bench.err; tail -3 gpurun_out/r2a/bench.json
