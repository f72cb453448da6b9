import torch, time, sys
sys.path.insert(0,'.')
from differentiable_ransac_amd import ops, synth
from differentiable_ransac_amd.ransac import BatchedRANSAC
dev='cuda'
P,N,B=32,2000,1024
data=synth.batch_two_view(P,N)
m=data['matches'].to(dev); lg=data['logits'].to(dev); K1=data['K1'].to(dev); K2=data['K2'].to(dev)
def t(fn,reps=10):
    fn(); torch.cuda.synchronize(); a=time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-a)/reps*1e3
print('nister nonminimal f64 all points [P,N]: %.3f ms' % t(lambda: ops.solve_nister5(m.double())))
mask=(torch.rand(P,N,device=dev)>0.5)
print('f8 LSQ on 1000 pts x 32 pairs (one call): %.3f ms' % t(lambda: ops.solve_f8(m[:, :1000].contiguous())))
for refit in (False, True):
    rn=BatchedRANSAC('nister',ransac_batch_size=B,max_iterations=B,refit=refit)
    print('BatchedRANSAC refit',refit,'%.3f ms'%t(lambda: rn(m,lg,K1,K2)))
rnf=BatchedRANSAC('f8',ransac_batch_size=B,max_iterations=B,refit=True)
dF=synth.batch_two_view(P,N,pixel=True)
print('BatchedRANSAC f8 refit %.3f ms'%t(lambda: rnf(dF['matches'].to(dev),dF['logits'].to(dev))))
