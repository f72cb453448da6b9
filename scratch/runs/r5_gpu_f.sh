#!/bin/bash
mkdir -p gpurun_out/r5f
O=$PWD/gpurun_out/r5f
R=$PWD
timeout 900 python -m pytest tests/test_gpu_round5.py tests/test_gpu_drivers.py tests/test_gpu_graphs.py tests/test_gpu_msac.py tests/test_gpu_sampler.py -q --timeout 300 > $O/pytest.log 2>&1; tail -5 $O/pytest.log
python scratch/dropin_loop.py 2>&1 | grep -v amdgpu.ids | tee $O/dropin_loop.log
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_dropin -o dropin -- python $R/scratch/dropin_loop.py > $O/prof_dropin.log 2>&1
python $R/tools/rocprof_summary.py $(find $O/prof_dropin -name "*results.db" | head -1) $O/r5_kernel_stats_dropin.md "python scratch/dropin_loop.py (32 pairs one by one through layers.RANSACLayer.forward, test mode, graph replay per pair)" last 2000
head -40 $O/r5_kernel_stats_dropin.md
rm -rf $O/prof_dropin
