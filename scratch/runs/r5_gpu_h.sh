#!/bin/bash
# round 5: MatchLoss value + gradient in one pass (tests, train lines fused / two-pass)
mkdir -p gpurun_out/r5h
O=$PWD/gpurun_out/r5h
timeout 900 python -m pytest tests/test_gpu_round5.py tests/test_gpu_drivers.py tests/test_gpu_round3.py tests/test_gpu_graphs.py tests/test_gpu_round2.py -q --timeout 300 > $O/pytest.log 2>&1; tail -6 $O/pytest.log
for i in 1 2; do
timeout 300 python bench.py --mode train --steps 300 > $O/bench_train.json 2> $O/bench_train.err; python -c "
import json; t=json.load(open('$O/bench_train.json')); print('train graph', round(t['value']/1e6,2), round(t['ms_per_step'],4))"
timeout 300 python bench.py --mode train --graph off --steps 300 > $O/bench_train_eager.json 2> $O/bench_train_eager.err; python -c "
import json; t=json.load(open('$O/bench_train_eager.json')); print('train eager', round(t['value']/1e6,2), round(t['ms_per_step'],4))"
done
