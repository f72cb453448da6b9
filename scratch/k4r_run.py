"""The rigid residual kernel at BASELINE configs[3] (one pair, 50 000 points, 2048 models), a few launches: target of rocprofv3 --pmc passes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from differentiable_ransac_amd import ops, synth
dev = 'cuda'
it = synth.rigid_pair(0, 50000)
pts = it['matches'][None].to(dev)
idx = ops.gumbel_topk(it['logits'][None].to(dev), 2048, 3, 1.0, None, 1, soft=False)["idx"]
smp = ops.gather(pts, idx)
model = ops.solve_rigid(smp.reshape(2048, 3, 6), None, False)[0].reshape(1, 2048, 4, 4)
for _ in range(int(os.environ.get('K4_PREWARM', '40')) + 6):
    ops.rigid_residual(pts, model, 0.03, True)
torch.cuda.synchronize()
print('done')
