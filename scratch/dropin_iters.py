"""iteration counts of the 32 pairs of scratch/dropin_loop.py (which device rounds a replayed drop-in call really runs)"""
import os, sys, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from differentiable_ransac_amd import layers, synth
dev = torch.device('cuda:0')
pairs, N = 32, 2000
d = synth.batch_two_view(pairs, N)
m, lg, K1, K2 = (d[k].to(dev) for k in ("matches", "logits", "K1", "K2"))
for B in (1024, 64):
    opt = types.SimpleNamespace(fmat=False, sampler=2, ransac_batch_size=B, tr=False, weighted=0, threshold=0.75, precision=1, device=str(dev))
    layer = layers.RANSACLayer(opt)
    its = []
    for p in range(pairs):
        _, mask, score, it = layer.estimator(m[p], lg[p], K1[p], K2[p], None)
        its.append((int(it), int(mask.sum())))
    print("rbs", B, "iterations / inliers:", its)
