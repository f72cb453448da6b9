#!/bin/bash
# round 5: device-side termination + the drop-in RANSAC call as a replayed graph (tests, the dropin_layer_loop record)
mkdir -p gpurun_out/r5e
O=gpurun_out/r5e
timeout 900 python -m pytest tests/test_gpu_round5.py tests/test_gpu_drivers.py tests/test_gpu_graphs.py tests/test_gpu_round4.py tests/test_gpu_msac.py tests/test_gpu_sampler.py -q -x --timeout 300 > $O/pytest.log 2>&1; tail -15 $O/pytest.log
timeout 300 python -c "
import json, torch, bench
print(json.dumps(bench.dropin_layer_loop_record(torch.device('cuda:0')), indent=1))" > $O/dropin.json 2> $O/dropin.err; cat $O/dropin.json; tail -5 $O/dropin.err
