#!/bin/bash
# round 6: the two committed bench lines (default command, the driver's command) taken AFTER profiles/r6_pmc_fetch_write.json of
# the same sources is in place, so that roofline.traffic is filled in from it
mkdir -p gpurun_out/r6y
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6y
cd $R
( time timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | grep real
( time timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err ) 2>&1 | grep real
python - <<PY
import json
for n in ("bench_default", "bench_driver"):
    d = json.loads([l for l in open("$O/" + n + ".json") if l.startswith("{")][-1])
    print(n, round(d["value"] / 1e6, 2), round(d["ms_per_step"], 4), d["roofline"]["avg_launch_ms"], round(d["roofline"]["frac"], 4), d["roofline"]["traffic"], d["cpu_baseline"]["value"], d["cpu_baseline"]["threads"], d["cpu_baseline"]["by_threads"])
    if "configs" in d:
        c = d["configs"]
        print(" c1", c["c1"]["ms_per_step"], "c3", c["c3"]["ms_per_step"], "c4", c["c4"]["ms_per_step"], "p1", c["c2_p1"]["ms_per_step"], "train", c["c5_train_p32"]["ms_per_step"], "refit", d["with_final_refit"]["ms_per_step"])
        print(" dropin", {k: (round(v["ms_per_pair"], 4) if isinstance(v, dict) else v) for k, v in c["dropin_layer_loop"].items() if k != "workload"})
        print(" c4 roofline", c["c4"]["scoring_roofline"]["frac"], c["c4"]["scoring_roofline"]["back_to_back"])
        for k in ("c1", "c3", "c4", "c5_train_p32"):
            b = c[k]["cpu_baseline"]; print(" cpu", k, round(b["value"], 1), b["threads"], b["by_threads"], b["spread"])
PY
