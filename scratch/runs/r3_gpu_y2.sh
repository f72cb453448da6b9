#!/bin/bash
# round 3, GPU pass Y2: K3 with the staged (coalesced) f32 output (tree) against the scattered stores ("k3nostage"), both with Sturm isolation
mkdir -p gpurun_out/r3y2
DRANSAC_LIB=$PWD/scratch/libdransac_k3nostage.so timeout 120 python scratch/k3_ab.py gpurun_out/r3y2/old.npz 2>&1 | grep K3 | tee gpurun_out/r3y2/k3.log
timeout 120 python scratch/k3_ab.py gpurun_out/r3y2/new.npz 2>&1 | grep -E "K3|Error|error" | tee -a gpurun_out/r3y2/k3.log
python scratch/k3_ab.py cmp gpurun_out/r3y2/old.npz gpurun_out/r3y2/new.npz 2>&1 | grep -v "^a ms" | tee -a gpurun_out/r3y2/k3.log
rm -f gpurun_out/r3y2/*.npz
timeout 900 python -m pytest tests/test_gpu_solvers.py tests/test_gpu_configs.py tests/test_gpu_edge_cases.py tests/test_gpu_drivers.py tests/test_gpu_round3.py -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/r3y2/tests.log
AB_ARGS="--segments 3 --prewarm-s 0.3" timeout 600 bash scratch/ab_step.sh k3nostage cur 2>&1 | tee gpurun_out/r3y2/ab.log
