#!/bin/bash
# round 4, pass J: five-point kernels compiled with floating-point reassociation allowed (ILP of the dependent f64 chains)
mkdir -p gpurun_out; O=$PWD/gpurun_out; L=$O/r4_j.log; : > $L
DRANSAC_LIB=$PWD/scratch/libdransac_k3ra.so timeout 600 python -m pytest tests/test_gpu_solvers.py tests/test_gpu_roots.py tests/test_gpu_round3.py tests/test_gpu_drivers.py tests/test_gpu_configs.py -q 2>&1 | tail -4 >> $L
AB_ARGS="--profile-kernels" bash scratch/ab_step.sh cur k3ra >> $L 2>&1
for n in cur k3ra cur k3ra; do
  lib=""; [ "$n" != "cur" ] && lib=$PWD/scratch/libdransac_$n.so
  DRANSAC_LIB=$lib timeout 200 python - >> $L 2>&1 <<PY
import torch, sys
sys.path.insert(0, '.')
from differentiable_ransac_amd import ops, synth
dev='cuda'; P,N,B=128,2000,1024
d=synth.batch_two_view(P,N)
r=ops.gumbel_topk(d['logits'].to(dev),B,5,1.0,None,seed=1,soft=False)
smp=ops.gather(d['matches'].to(dev), r['idx'])
for _ in range(10): ops.solve_nister5(smp)
torch.cuda.synchronize()
a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(50): m,v=ops.solve_nister5(smp)
b.record(); torch.cuda.synchronize()
t1=a.elapsed_time(b)/50
a.record()
for _ in range(50): ms,vs=ops.solve_stewenius5(smp)
b.record(); torch.cuda.synchronize()
print('$n nister', round(t1*1e3,1), 'us  valid', int(v.sum()), ' stewenius', round(a.elapsed_time(b)/50*1e3,1), 'us valid', int(vs.sum()))
PY
done
