"""Runs the two K4 kernel families (general kernel = path 1, matrix-core filter kernel = path 2) on the benchmark shape a few
times: the target of the rocprofv3 --pmc passes of scratch/r2_k4_pmc.sh."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from differentiable_ransac_amd import ops, synth
dev = 'cuda'
P, N, B = 32, 2000, 1024
data = synth.batch_two_view(P, N)
r = ops.gumbel_topk(data['logits'].to(dev), B, 5, 1.0, None, seed=1)
smp = ops.gather(data['matches'].to(dev), r['idx'])
models, valid = ops.solve_nister5(smp)
flat = models.reshape(P, -1, 3, 3).contiguous()
v = valid.reshape(P, -1).contiguous()
mt = data['matches'].to(dev).contiguous()
thr = torch.full((P,), 7.5e-4, device=dev)
for _ in range(6):
    ops.msac_score(mt, flat, thr, True, v, path=1)
    ops.msac_score(mt, flat, thr, True, v, path=2)
torch.cuda.synchronize()
print('done')
