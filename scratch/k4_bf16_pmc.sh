#!/bin/bash
# SQ counters of the bf16-matrix-core scoring prototype (run on the GPU box through gpurun):
#   gpurun --timeout 300 -- 'bash scratch/k4_bf16_pmc.sh pipe'
# Two counter passes (rocprofv3 --pmc with --kernel-trace only), summaries under gpurun_out/.
V=${1:-base}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace -d $R/gpurun_out/pmc_q1 -o q1 -- \
  python $R/scratch/k4_bf16.py $V > $R/gpurun_out/pmc_q1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_WAVES --kernel-trace -d $R/gpurun_out/pmc_q2 -o q2 -- \
  python $R/scratch/k4_bf16.py $V > $R/gpurun_out/pmc_q2.log 2>&1
cd $R
python scratch/pmc_dump.py gpurun_out/pmc_q1/q1_results.db gpurun_out/pmc_q2/q2_results.db | tee gpurun_out/k4_bf16_pmc_$V.txt
grep -E "prototype|production" gpurun_out/pmc_q1.log | tail -3
rm -rf gpurun_out/pmc_q1 gpurun_out/pmc_q2
