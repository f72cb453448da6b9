#!/bin/bash
# round 5, first pass: two-phase five-point kernels (correctness + timing against the lane pairs, Sturm drain A/B, stage cycles),
# K4 with the empty rows issued inside the model loop (bit-identity + in-step A/B)
mkdir -p gpurun_out/r5a
O=gpurun_out/r5a
timeout 900 python -m pytest tests/test_gpu_round5.py tests/test_gpu_roots.py tests/test_gpu_solvers.py tests/test_gpu_round4.py tests/test_gpu_configs.py -q -x --timeout 300 > $O/pytest.log 2>&1; tail -5 $O/pytest.log
timeout 300 python scratch/r5_k3_time.py 131072 65536 32768 > $O/k3_time.log 2>&1; cat $O/k3_time.log
DRANSAC_LIB=$PWD/scratch/libdransac_k3nodrain.so timeout 300 python scratch/r5_k3_time.py 131072 32768 > $O/k3_time_nodrain.log 2>&1; cat $O/k3_time_nodrain.log
timeout 200 python scratch/prof_stages.py > $O/k3_stages_pair.log 2>&1; tail -2 $O/k3_stages_pair.log
K3_PAIRS=128 K3_PATH=2 timeout 200 python scratch/prof_stages.py > $O/k3_stages_fb.log 2>&1; tail -2 $O/k3_stages_fb.log
K3_PAIRS=128 K3_PATH=1 timeout 200 python scratch/prof_stages.py > $O/k3_stages_pair128.log 2>&1; tail -2 $O/k3_stages_pair128.log
timeout 200 python scratch/k4_equal.py cur k4z1 > $O/k4_equal.log 2>&1; timeout 200 python scratch/k4_equal.py cur k4z2 >> $O/k4_equal.log 2>&1; cat $O/k4_equal.log
bash scratch/ab_step.sh cur k4z1 k4z2 > $O/k4_ab.log 2>&1; cat $O/k4_ab.log
