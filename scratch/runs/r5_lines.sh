#!/bin/bash
# round 5: the two committed bench lines (default command, the driver's command) taken AFTER profiles/r5_pmc_fetch_write.json of
# the same sources is in place, so that roofline.traffic is filled in from it
mkdir -p gpurun_out/r5y
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5y
cd $R
( time timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | grep real
( time timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err ) 2>&1 | grep real
python - <<PY
import json
for n in ("bench_default", "bench_driver"):
    d = json.load(open("$O/" + n + ".json"))
    print(n, round(d["value"] / 1e6, 2), round(d["ms_per_step"], 4), d["roofline"]["avg_launch_ms"], round(d["roofline"]["frac"], 4), d["roofline"]["traffic"], d["cpu_baseline"]["value"])
    if "configs" in d:
        c = d["configs"]
        print(" c1", c["c1"]["ms_per_step"], "c3", c["c3"]["ms_per_step"], "c4", c["c4"]["ms_per_step"], "p1", c["c2_p1"]["ms_per_step"], "train", c["c5_train_p32"]["ms_per_step"], "dropin", c["dropin_layer_loop"]["test_mode"]["ms_per_pair"], "refit", d["with_final_refit"]["ms_per_step"])
PY
