#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_round6.py -m gpu -q -x 2>&1 | tail -15
