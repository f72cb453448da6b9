#!/bin/bash
mkdir -p gpurun_out/r5l
O=$PWD/gpurun_out/r5l
timeout 600 python -m pytest tests/test_gpu_round5.py -q --timeout 300 -k "screened or two_phase" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for v in "" k3rebasis "" k3rebasis; do
  lib=""; [ -n "$v" ] && lib=$PWD/scratch/libdransac_$v.so
  echo "== ${v:-tree}"; DRANSAC_LIB=$lib timeout 200 python scratch/r5_k3_time.py 131072 65536 2>&1 | grep -v amdgpu.ids | grep nister
done > $O/k3_rebasis.log 2>&1; cat $O/k3_rebasis.log
