#!/bin/bash
mkdir -p gpurun_out/r2d
O=gpurun_out/r2d
timeout 600 python -m pytest tests/test_gpu_round2.py -x -q > $O/pytest_round2.log 2>&1; tail -5 $O/pytest_round2.log
for P in 32 64 128 256; do
  timeout 300 python bench.py --pairs $P --steps 300 --no-configs --no-cpu-baseline > $O/bench_p$P.json 2> $O/bench_p$P.err
  python - <<PY
import json
d=json.load(open("$O/bench_p$P.json"))
print("pairs", $P, "Mhyps/s", round(d["value"]/1e6,2), "ms/step", round(d["ms_per_step"],4), "K4 ms", round(d["roofline"]["avg_launch_ms"],4), "frac", round(d["roofline"]["frac"],4), "two streams", round(d["two_batches_in_flight"]["value"]/1e6,2))
PY
done
timeout 300 python bench.py --pairs 128 --steps 100 --no-configs --no-cpu-baseline --profile-kernels > $O/bench_p128_kernels.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/bench_p128_kernels.json')); print(d['kernel_breakdown_ms'])"
timeout 300 python bench.py --logits-fixture --steps 300 --no-configs --no-cpu-baseline > $O/bench_fixture.json 2> $O/bench_fixture.err; python -c "
import json; d=json.load(open('$O/bench_fixture.json')); print('fixture', d['value']/1e6, d['check'])"
