"""target of the rocprofv3 --pmc passes over the register sampler at 128 x 1024 x 2000, k = 5 (race form): 6 launches"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from differentiable_ransac_amd import ops, synth
dev = 'cuda'
d = synth.batch_two_view(128, 2000)
m, lg = d['matches'].to(dev), d['logits'].to(dev)
for _ in range(6):
    ops.gumbel_topk_gather(m, lg, 1024, 5, 1.0, 7, race=True)
torch.cuda.synchronize()
