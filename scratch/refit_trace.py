import torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from differentiable_ransac_amd import synth
from differentiable_ransac_amd.ransac import BatchedRANSAC
dev='cuda'; P,N,B=32,2000,1024
d=synth.batch_two_view(P,N); args=(d['matches'].to(dev),d['logits'].to(dev),d['K1'].to(dev),d['K2'].to(dev))
rn=BatchedRANSAC('nister',ransac_batch_size=B,max_iterations=B,refit=True)
for _ in range(6): out=rn(*args)
torch.cuda.synchronize()
