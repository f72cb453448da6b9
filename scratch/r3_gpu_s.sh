#!/bin/bash
# round 3, GPU pass S: K1 register kernel with the candidates collected by the lanes that own them ("cur") against the ballot
# compaction ("ballot"): sampler parity tests on the tree library, then the c2 step
mkdir -p gpurun_out/r3s
(timeout 300 python -m pytest tests/test_gpu_sampler.py tests/test_gpu_round2.py tests/test_gpu_graphs.py tests/test_gpu_edge_cases.py tests/test_gpu_drivers.py -m gpu -q -x --timeout 300 > gpurun_out/r3s/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3s/pytest.log)
tail -3 gpurun_out/r3s/pytest.log
for round in 1 2; do
  for n in cur ballot; do
    lib=""; [ "$n" != "cur" ] && lib=$PWD/scratch/libdransac_$n.so
    DRANSAC_LIB=$lib timeout 200 python bench.py --steps 300 --warmup 30 --segments 3 --prewarm-s 0.3 --no-configs --no-cpu-baseline --no-extras --profile-kernels 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$n', round(d['value']/1e6,2), 'M  step', round(d['ms_per_step'],4), 'ms  K1', round(d['kernel_breakdown_ms']['K1_gumbel_topk'],4))"
  done
done 2>&1 | tee gpurun_out/r3s/ab_passb.log
