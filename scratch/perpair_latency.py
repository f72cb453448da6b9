import torch, time, sys
sys.path.insert(0,'.')
from differentiable_ransac_amd import synth
from differentiable_ransac_amd.estimators import EssentialMatrixEstimatorNister
from differentiable_ransac_amd.ransac import RANSAC, BatchedRANSAC
from differentiable_ransac_amd.samplers import GumbelSoftmaxSampler
from differentiable_ransac_amd.scorings import MSACScore
dev='cuda'
pair=synth.two_view_pair(0,2000)
m=pair['matches'].to(dev); lg=pair['logits'].to(dev); K1=pair['K1'].to(dev); K2=pair['K2'].to(dev)
def t(fn,reps=5):
    fn(); torch.cuda.synchronize(); a=time.perf_counter()
    for _ in range(reps): out=fn()
    torch.cuda.synchronize(); return (time.perf_counter()-a)/reps*1e3, out
for rbs,maxit in ((1024,1024),(64,5000),(1024,5000)):
    r=RANSAC(EssentialMatrixEstimatorNister('cuda'),GumbelSoftmaxSampler(rbs,5,device='cuda'),MSACScore('cuda'),train=False,
             ransac_batch_size=rbs,sampler_id=2,threshold=0.75,max_iterations=maxit)
    ms,out=t(lambda: r(m,lg,K1,K2,None))
    print(f'reference-API RANSAC test mode rbs={rbs} max_it={maxit}: {ms:.2f} ms/pair, iterations {out[3]}, score {float(out[2]):.1f}')
    b=BatchedRANSAC('nister',ransac_batch_size=rbs,max_iterations=maxit,refit=True)
    ms,out=t(lambda: b(m[None],lg[None],K1[None],K2[None]))
    print(f'BatchedRANSAC P=1 rbs={rbs} max_it={maxit}: {ms:.2f} ms/pair, iterations {int(out["iterations"][0])}, score {float(out["score"][0]):.1f}')
