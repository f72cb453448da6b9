#!/bin/bash
# round 5, K3 second pass: two-phase kernels with the hand-over in accumulation registers; flat isolation step, f32 polish tolerance,
# symmetric G -- each against the tree build on one box
mkdir -p gpurun_out/r5c
O=gpurun_out/r5c
timeout 900 python -m pytest tests/test_gpu_round5.py tests/test_gpu_roots.py tests/test_gpu_solvers.py tests/test_gpu_round4.py tests/test_gpu_round3.py tests/test_gpu_configs.py -q -x --timeout 300 > $O/pytest.log 2>&1; tail -4 $O/pytest.log
for v in "" k3isobranch k3tol17 k3nosymg; do
  lib=""; [ -n "$v" ] && lib=$PWD/scratch/libdransac_$v.so
  echo "== ${v:-tree}"; DRANSAC_LIB=$lib timeout 200 python scratch/r5_k3_time.py 131072 32768 2>&1 | grep -v amdgpu.ids
done > $O/k3_variants.log 2>&1; cat $O/k3_variants.log
timeout 100 python scratch/r5_k3_time.py 65536 1024 2>&1 | grep -v amdgpu.ids | tee $O/k3_sizes.log
K3_PAIRS=128 K3_PATH=1 timeout 200 python scratch/prof_stages.py 2>&1 | tail -1 > $O/k3_stages_pair128.log; cat $O/k3_stages_pair128.log
K3_PAIRS=128 K3_PATH=2 timeout 200 python scratch/prof_stages.py 2>&1 | tail -1 > $O/k3_stages_fb128.log; cat $O/k3_stages_fb128.log
