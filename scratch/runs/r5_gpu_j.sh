#!/bin/bash
mkdir -p gpurun_out/r5j
O=$PWD/gpurun_out/r5j
timeout 900 python -m pytest tests/test_gpu_roots.py tests/test_gpu_solvers.py tests/test_gpu_round5.py -q --timeout 300 -x > $O/pytest.log 2>&1; tail -12 $O/pytest.log
for v in "" k3nofb; do
  lib=""; [ -n "$v" ] && lib=$PWD/scratch/libdransac_$v.so
  echo "== ${v:-tree}"; DRANSAC_LIB=$lib timeout 200 python scratch/r5_k3_time.py 131072 32768 2>&1 | grep -v amdgpu.ids
done > $O/k3_fallback_cost.log 2>&1; cat $O/k3_fallback_cost.log
