#!/bin/bash
# round 3, GPU pass T: non-minimal five-point backward (new tests) + solver / driver suites on the K3 pre-check default
mkdir -p gpurun_out/r3t
timeout 900 python -m pytest tests/test_gpu_round3.py -m gpu -x -q -k "nonminimal or eight_point" 2>&1 | tail -25 | tee gpurun_out/r3t/new.log
timeout 900 python -m pytest tests/test_gpu_solvers.py tests/test_gpu_drivers.py tests/test_gpu_configs.py -m gpu -x -q 2>&1 | tail -5 | tee gpurun_out/r3t/suites.log
