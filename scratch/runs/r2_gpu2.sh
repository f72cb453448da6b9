#!/bin/bash
mkdir -p gpurun_out/r2b
export TMPDIR=/tmp
timeout 600 python scratch/k4f_check.py > gpurun_out/r2b/k4f_check.log 2>&1; echo "k4f_check rc $?" >> gpurun_out/r2b/k4f_check.log
grep -v "^      p " gpurun_out/r2b/k4f_check.log | tail -60
