#!/bin/bash
# round-2 GPU call: full -m gpu suite, default bench, rocprofv3 kernel stats per BASELINE config, PMC passes
mkdir -p gpurun_out/r2c
export TMPDIR=/tmp
O=gpurun_out/r2c
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -8 $O/pytest.log
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?"; head -c 1500 $O/bench_default.json; echo
cd /tmp
for wl in c2 c1 c3 c4; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_$wl -o bench -- python $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 100 --warmup 5 --no-configs --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/prof_$wl.json 2> $GRAFT_REPO_ROOT/$O/prof_$wl.err
  DB=$(find $GRAFT_REPO_ROOT/$O/prof_$wl -name "*results.db" | head -1)
  python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $DB $GRAFT_REPO_ROOT/$O/r2_kernel_stats_$wl.md "python bench.py --workload $wl --steps 100 --warmup 5 --no-configs --no-cpu-baseline" first 105
done
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace -d $GRAFT_REPO_ROOT/$O/pmc_$c -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 3 --no-configs --no-cpu-baseline > /dev/null 2> $GRAFT_REPO_ROOT/$O/pmc_$c.err
done
python $GRAFT_REPO_ROOT/tools/rocprof_pmc_summary.py $GRAFT_REPO_ROOT/$O/r2_pmc_fetch_write.md $GRAFT_REPO_ROOT/$O/r2_pmc_fetch_write.json $(find $GRAFT_REPO_ROOT/$O/pmc_FETCH_SIZE $GRAFT_REPO_ROOT/$O/pmc_WRITE_SIZE -name "*results.db")
cd $GRAFT_REPO_ROOT
timeout 300 python bench.py --mode train --steps 200 > $O/bench_train.json 2> $O/bench_train.err; head -c 600 $O/bench_train.json; echo
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 100 --no-configs --no-cpu-baseline > $O/bench_torchrun1.json 2> $O/bench_torchrun1.err; head -c 300 $O/bench_torchrun1.json; echo
rm -rf $O/prof_c*/ $O/pmc_*_SIZE 2>/dev/null
ls $O
