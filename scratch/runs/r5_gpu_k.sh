#!/bin/bash
mkdir -p gpurun_out/r5k
O=$PWD/gpurun_out/r5k
timeout 900 python -m pytest tests/test_gpu_round5.py tests/test_gpu_round4.py tests/test_gpu_edge_cases.py tests/test_gpu_solvers.py -q --timeout 300 > $O/pytest.log 2>&1; tail -12 $O/pytest.log
timeout 300 python tools/run_path.py -nf 2000 -bs 4 -rbs 256 -fmat 0 -sam 2 -tr 1 -w2 1 -t 0.75 -pr 2 > $O/run_path_f64.json 2> $O/run_path_f64.err; tail -c 700 $O/run_path_f64.json; tail -3 $O/run_path_f64.err
timeout 300 python scratch/k3_stats.py 2>&1 | grep -v amdgpu.ids | tee $O/k3_stats.log
