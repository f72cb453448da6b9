import torch, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from differentiable_ransac_amd import ops, synth
dev='cuda'; P,N,B=32,2000,1024
d=synth.batch_two_view(P,N); m=d['matches'].to(dev); lg=d['logits'].to(dev)
for _ in range(4):
    r=ops.gumbel_topk(lg,B,5,1.0,None,seed=1)
    smp=ops.gather(m,r['idx'],r['y_sel'])
    models,valid=ops.solve_nister5(smp)
    thr=torch.full((P,),7.5e-4,device=dev)
    sc,mk=ops.msac_score(m,models.reshape(P,-1,3,3),thr,True,valid.reshape(P,-1))
torch.cuda.synchronize()
