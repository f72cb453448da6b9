#!/bin/bash
# round 3, GPU pass R: K3 with the residual pre-check before the first Gauss-Newton step ("k3pre") against the tree ("cur")
mkdir -p gpurun_out/r3r
V=$PWD/scratch/libdransac_k3pre.so
python scratch/k3_ab.py gpurun_out/r3r/cur.npz 2>&1 | tee gpurun_out/r3r/k3_cur.log
DRANSAC_LIB=$V python scratch/k3_ab.py gpurun_out/r3r/pre.npz 2>&1 | tee gpurun_out/r3r/k3_pre.log
python scratch/k3_ab.py cmp gpurun_out/r3r/cur.npz gpurun_out/r3r/pre.npz 2>&1 | tee gpurun_out/r3r/cmp.log
rm -f gpurun_out/r3r/*.npz
DRANSAC_LIB=$V timeout 600 python -m pytest tests/test_gpu_solvers.py tests/test_gpu_configs.py tests/test_gpu_drivers.py -m gpu -x -q 2>&1 | tail -5 | tee gpurun_out/r3r/tests.log
AB_ARGS="--segments 3 --prewarm-s 0.3" bash scratch/ab_step.sh cur k3pre 2>&1 | tee gpurun_out/r3r/ab.log
