#!/bin/bash
mkdir -p gpurun_out/r2e
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r2e/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r2e/pytest.log; tail -12 gpurun_out/r2e/pytest.log
