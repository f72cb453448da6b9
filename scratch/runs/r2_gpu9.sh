#!/bin/bash
# end-of-round validation: full GPU suite, smoke(), default bench, torchrun N=1 bench, train bench (graph + eager), two-rank gloo functional runs
mkdir -p gpurun_out/r2m
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2m
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?"
timeout 300 python bench.py --mode train --steps 300 > $O/bench_train.json 2> $O/bench_train.err; echo "train rc $?"
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 200 --warmup 10 --no-configs --no-cpu-baseline > $O/bench_torchrun1.json 2> $O/bench_torchrun1.err; echo "torchrun rc $?"; head -c 250 $O/bench_torchrun1.json; echo
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --gpus-shared --backend gloo --mode train --steps 50 --warmup 5 > $O/bench_train_2rank.json 2> $O/bench_train_2rank.err; echo "2-rank train rc $?"; head -c 400 $O/bench_train_2rank.json; echo
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 2 --gpus-shared --backend gloo --split hypotheses --pairs 4 --steps 50 --warmup 5 --no-configs --no-cpu-baseline > $O/bench_split_2rank.json 2> $O/bench_split_2rank.err; echo "2-rank split rc $?"; head -c 300 $O/bench_split_2rank.json; echo
python - <<PY
import json
d=json.load(open("$O/bench_default.json"))
print("default", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic"], d["two_batches_in_flight"], d["with_final_refit"], d["sampler_topdown"] and d["sampler_topdown"]["value"])
for k,v in d["configs"].items(): print(k, round(v["ms_per_step"],4), v["issue"][:20], round(v["hypotheses_per_s"]/1e6,1), "eager", round(v["eager_ms_per_step"],4), "graph", v["graph_replay_ms_per_step"])
print("clnet", d["clnet_logits"]["ms_per_step"], "cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
t=json.load(open("$O/bench_train.json")); print("train", t["value"], t["ms_per_step"])
PY
