#!/bin/bash
# round 6, run h: K1 upper bound of the one-logarithm form (timing builds with wrong noise: one / no logarithm) at 128 pairs; full GPU suite; bench
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6h
for rep in 1 2; do AB_K1_PAIRS=128 timeout 300 python scratch/ab_k1.py base onelog nolog 2>&1 | grep "test mode"; done | tee gpurun_out/r6h/k1.log
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r6h/pytest.log 2>&1; tail -4 gpurun_out/r6h/pytest.log
timeout 900 python bench.py > gpurun_out/r6h/bench.json 2> gpurun_out/r6h/bench.err; tail -c 300 gpurun_out/r6h/bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r6h/bench.json"))
print("headline", round(d["value"] / 1e6, 2), d["ms_per_step"], d["roofline"]["frac"], "cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["threads"], d["cpu_baseline"]["by_threads"], d["cpu_baseline"]["spread"])
c = d["configs"]
for k in ("c1", "c3", "c4", "c2_p1", "c5_train_p32"):
    print(k, c[k]["ms_per_step"], {kk: (round(v, 1) if isinstance(v, float) else v) for kk, v in c[k].get("cpu_baseline", {}).items() if kk in ("value", "threads", "by_threads", "spread", "seconds")})
print("c4 roofline", c["c4"]["scoring_roofline"])
print("dropin", {k: (v.get("ms_per_pair") if isinstance(v, dict) else v) for k, v in c["dropin_layer_loop"].items() if k != "workload"})
print("refit", d["with_final_refit"]["ms_per_step"], "run_s", d.get("run_seconds"))
PY
