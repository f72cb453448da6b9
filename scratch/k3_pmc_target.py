"""target of the rocprofv3 --pmc passes over the five-point kernels at 131 072 samples: both paths of both solvers, 6 launches each"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from differentiable_ransac_amd import ops, synth
dev = 'cuda'
d = synth.batch_two_view(128, 2000)
r = ops.gumbel_topk(d['logits'].to(dev), 1024, 5, 1.0, None, seed=1, soft=False)
smp = ops.gather(d['matches'].to(dev), r['idx']).reshape(-1, 5, 4).contiguous()
for _ in range(6):
    for fn in (ops.solve_nister5, ops.solve_stewenius5):
        for path in (1, 2):
            fn(smp, path=path)
torch.cuda.synchronize()
