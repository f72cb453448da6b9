"""The screened long-row sampler at BASELINE configs[3] (50 000 points x 2048 rows, k = 3): target of rocprofv3 --pmc passes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from differentiable_ransac_amd import ops, synth
dev = 'cuda'
lg = synth.rigid_pair(0, 50000)['logits'][None].to(dev)
scr = os.environ.get('SCR', '1') == '1'
for i in range(46):
    ops.gumbel_topk(lg, 2048, 3, 1.0, None, i, soft=False, screen=scr)
torch.cuda.synchronize()
print('done')
