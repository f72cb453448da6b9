#!/bin/bash
# round 3, GPU pass F: full GPU suite, then the c4 workload (split-row sampler, per-launch tile of the rigid residual kernel), the
# train step, and the headline at the driver's settings
mkdir -p gpurun_out/r3f
(timeout 600 python -m pytest tests -m gpu -q -x --timeout 300 > gpurun_out/r3f/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3f/pytest.log)
tail -4 gpurun_out/r3f/pytest.log
timeout 200 python bench.py --workload c4 --steps 200 --warmup 5 --segments 3 --no-configs --no-cpu-baseline --no-extras --profile-kernels > gpurun_out/r3f/c4.json 2> gpurun_out/r3f/c4.err
python - <<'PY'
import json
r=json.load(open("gpurun_out/r3f/c4.json")); print("c4", r["value"], r["ms_per_step"], r["roofline"]["avg_launch_ms"], r["roofline"]["frac"])
PY
timeout 200 python bench.py --mode train --steps 200 --warmup 5 --segments 3 > gpurun_out/r3f/train.json 2> gpurun_out/r3f/train.err
python - <<'PY'
import json
r=json.load(open("gpurun_out/r3f/train.json")); print("train", r["value"], r["ms_per_step"], r["segments"]["ms_per_step"])
PY
timeout 280 python bench.py --steps 20 --warmup 5 > gpurun_out/r3f/bench_driver.json 2> gpurun_out/r3f/bench_driver.err
python - <<'PY'
import json
r=json.load(open("gpurun_out/r3f/bench_driver.json")); print("c2", r["value"], r["ms_per_step"], r["roofline"]["avg_launch_ms"], r["roofline"]["frac"])
for k,v in r["configs"].items(): print(k, v["ms_per_step"], v["launch_ms"], v.get("scoring_roofline",{}).get("frac"))
PY
