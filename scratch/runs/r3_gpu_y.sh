#!/bin/bash
# round 3, GPU pass Y: K3 with Sturm-sequence root isolation (tree) against the derivative chain ("k3nosturm")
mkdir -p gpurun_out/r3y
DRANSAC_LIB=$PWD/scratch/libdransac_k3nosturm.so timeout 120 python scratch/k3_ab.py gpurun_out/r3y/old.npz 2>&1 | grep K3 | tee gpurun_out/r3y/k3.log
timeout 120 python scratch/k3_ab.py gpurun_out/r3y/new.npz 2>&1 | grep -E "K3|Error|error" | tee -a gpurun_out/r3y/k3.log
python scratch/k3_ab.py cmp gpurun_out/r3y/old.npz gpurun_out/r3y/new.npz 2>&1 | tee -a gpurun_out/r3y/k3.log
rm -f gpurun_out/r3y/*.npz
timeout 600 python -m pytest tests/test_gpu_solvers.py tests/test_gpu_configs.py tests/test_gpu_edge_cases.py -m gpu -x -q 2>&1 | tail -5 | tee gpurun_out/r3y/tests.log
AB_ARGS="--segments 3 --prewarm-s 0.3" timeout 600 bash scratch/ab_step.sh k3nosturm cur 2>&1 | tee gpurun_out/r3y/ab.log
