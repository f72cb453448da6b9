import torch, time, sys
sys.path.insert(0,'.')
from differentiable_ransac_amd import ops, synth
from differentiable_ransac_amd.ransac import BatchedRANSAC
dev='cuda'; P,N,B=32,2000,1024
d=synth.batch_two_view(P,N); m=d['matches'].to(dev)
mask=(torch.rand(P,N,device=dev)>0.5)
def t(fn,reps=20):
    fn(); torch.cuda.synchronize()
    a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b)/reps*1e3
print('refit_essential  [32 pairs x 2000 pts]: %.1f us'%t(lambda: ops.refit_essential(m)))
print('refit_fundamental[32 pairs x ~1000 inl]: %.1f us'%t(lambda: ops.refit_fundamental(m,mask)))
thr=torch.full((P,),7.5e-4,device=dev)
cand,cv=ops.refit_essential(m)
print('msac on 10 candidates: %.1f us'%t(lambda: ops.msac_score(m,cand,thr,want_masks=False)))
cs,_=ops.msac_score(m,cand,thr,want_masks=False)
print('select_best on 10 candidates: %.1f us'%t(lambda: ops.select_best(m,cand,cs,thr,cv)))
for solver in ('nister','f8'):
    dd=synth.batch_two_view(P,N,pixel=(solver=='f8'))
    args=(dd['matches'].to(dev),dd['logits'].to(dev)) + ((dd['K1'].to(dev),dd['K2'].to(dev)) if solver=='nister' else ())
    for refit in (False,True):
        rn=BatchedRANSAC(solver,ransac_batch_size=B,max_iterations=B,refit=refit)
        print(solver,'refit',refit,'%.1f us'%t(lambda: rn(*args)))
