#!/bin/bash
mkdir -p gpurun_out/r2f
O=$GRAFT_REPO_ROOT/gpurun_out/r2f
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd $R
timeout 600 python -m pytest tests/test_gpu_round2.py tests/test_gpu_pose.py -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 300 python scratch/ab_k1.py > $O/ab_k1.log 2>&1; cat $O/ab_k1.log
# N > 1 code paths on one GPU (functional check: gloo, both ranks on cuda:0)
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --backend gloo --gpus-shared --mode train --steps 50 --warmup 5 > $O/bench_train_2ranks_gloo.json 2> $O/bench_train_2ranks_gloo.err; head -c 900 $O/bench_train_2ranks_gloo.json; echo
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 2 --backend gloo --gpus-shared --split hypotheses --pairs 1 --steps 50 --warmup 5 --no-configs --no-cpu-baseline > $O/bench_split_2ranks_gloo.json 2> $O/bench_split_2ranks_gloo.err; head -c 700 $O/bench_split_2ranks_gloo.json; echo
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 2 --backend gloo --gpus-shared --steps 50 --warmup 5 --no-configs --no-cpu-baseline > $O/bench_test_2ranks_gloo.json 2> $O/bench_test_2ranks_gloo.err; head -c 400 $O/bench_test_2ranks_gloo.json; echo
tail -3 $O/bench_train_2ranks_gloo.err
cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace -d $O/pmc_q1 -o q1 -- python $R/scratch/k4_paths_run.py > $O/pmc_q1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_WAVES --kernel-trace -d $O/pmc_q2 -o q2 -- python $R/scratch/k4_paths_run.py > $O/pmc_q2.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_BUSY_CYCLES --kernel-trace -d $O/pmc_q3 -o q3 -- python $R/scratch/k4_paths_run.py > $O/pmc_q3.log 2>&1
cd $R
python scratch/pmc_dump.py $(find $O/pmc_q1 $O/pmc_q2 $O/pmc_q3 -name "*results.db") > $O/k4_paths_sq_counters.txt 2>&1; cat $O/k4_paths_sq_counters.txt | head -40
rm -rf $O/pmc_q1 $O/pmc_q2 $O/pmc_q3
