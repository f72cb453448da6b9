"""The general scoring kernel at K4_PAIRS pairs (default 128), a few launches: target of rocprofv3 --pmc passes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from differentiable_ransac_amd import ops, synth
dev = 'cuda'
P, N, B = int(os.environ.get('K4_PAIRS', '128')), 2000, 1024
data = synth.batch_two_view(P, N)
r = ops.gumbel_topk(data['logits'].to(dev), B, 5, 1.0, None, seed=1)
smp = ops.gather(data['matches'].to(dev), r['idx'])
models, valid = ops.solve_nister5(smp)
flat = models.reshape(P, -1, 3, 3).contiguous()
v = valid.reshape(P, -1).contiguous()
mt = data['matches'].to(dev).contiguous()
thr = torch.full((P,), 7.5e-4, device=dev)
for _ in range(int(os.environ.get('K4_PREWARM', '0')) + 6):   # K4_PREWARM: launches before the ones worth reading (clock ramp)
    ops.msac_score(mt, flat, thr, True, v, path=1)
torch.cuda.synchronize()
x = torch.empty(P * 10240 * N, dtype=torch.uint8, device=dev)
for _ in range(6):
    x.zero_()
torch.cuda.synchronize()
print('done')
