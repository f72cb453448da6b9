#!/bin/bash
mkdir -p gpurun_out/r3m
O=gpurun_out/r3m
(timeout 600 python -m pytest tests -m gpu -q --timeout 300 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log)
tail -3 $O/pytest.log
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?"
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "driver-style rc $?"
python - <<PY
import json
for f in ("bench_default","bench_driver"):
    d=json.load(open("$O/%s.json"%f)); r=d["roofline"]
    print(f, d["value"], d["ms_per_step"], r["avg_launch_ms"], r["frac"], r["traffic"], r["traffic_source"][:60])
PY
