#!/bin/bash
# round 5, K3: where does the lane-pair kernel spend its time?  timing of cut-down builds + finer stage cycles
mkdir -p gpurun_out/r5b
O=gpurun_out/r5b
for v in "" k3skipfinal k3stageout k3nostore; do
  lib=""; [ -n "$v" ] && lib=$PWD/scratch/libdransac_$v.so
  echo "== ${v:-tree}"; DRANSAC_LIB=$lib K3_ONLY_PAIRS=1 timeout 200 python scratch/r5_k3_time.py 131072 32768 2>&1 | grep -v amdgpu.ids
done > $O/k3_cuts.log 2>&1; cat $O/k3_cuts.log
K3_PAIRS=128 K3_PATH=1 timeout 200 python scratch/prof_stages.py 2>&1 | tail -1 > $O/k3_stages_pair128.log; cat $O/k3_stages_pair128.log
K3_PAIRS=32 K3_PATH=1 timeout 200 python scratch/prof_stages.py 2>&1 | tail -1 > $O/k3_stages_pair32.log; cat $O/k3_stages_pair32.log
K3_PAIRS=128 K3_PATH=2 timeout 200 python scratch/prof_stages.py 2>&1 | tail -1 > $O/k3_stages_fb128.log; cat $O/k3_stages_fb128.log
