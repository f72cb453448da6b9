#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_drivers.py tests/test_gpu_round2.py -m gpu -q -x 2>&1 | tail -3
python - <<'PY'
import sys, types, time, torch
sys.argv = ["bench.py"]
import bench
r = bench.dropin_layer_loop_record(torch.device("cuda:0"))
print({k: (round(v["ms_per_pair"], 4) if isinstance(v, dict) else v) for k, v in r.items() if k != "workload"})
PY
