#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_round6.py tests/test_gpu_drivers.py tests/test_gpu_round5.py -m gpu -q --timeout 300 2>&1 | tail -4
for rep in 1 2 3; do for rbs in 1024 64; do echo "dropin rbs=$rbs $(DROPIN_RBS=$rbs timeout 300 python scratch/dropin_loop.py 2>&1 | grep 'ms per pair')"; done; done
