#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_round6.py -m gpu -q --timeout 300 2>&1 | tail -8
