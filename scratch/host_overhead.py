import torch, time, sys
sys.path.insert(0,'.')
from differentiable_ransac_amd import synth
from differentiable_ransac_amd.ransac import BatchedRANSAC
dev='cuda'; P,N,B=32,2000,1024
d=synth.batch_two_view(P,N); m=d['matches'].to(dev); lg=d['logits'].to(dev); K1=d['K1'].to(dev); K2=d['K2'].to(dev)
for samp in ('gumbel','topdown'):
    rn=BatchedRANSAC('nister',ransac_batch_size=B,threshold=0.75,max_iterations=B,keep_masks=True,refit=False,sampling=samp)
    st=[torch.cuda.Stream() for _ in range(2)]
    keep=[None,None]
    for i in range(6):
        with torch.cuda.stream(st[i%2]): keep[i%2]=rn(m,lg,K1,K2)
    torch.cuda.synchronize()
    t0=time.perf_counter()
    for i in range(100):
        with torch.cuda.stream(st[i%2]): keep[i%2]=rn(m,lg,K1,K2)
    t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
    print(samp,'host issue %.3f ms/step, total %.3f ms/step'%((t1-t0)*10,(t2-t0)*10))
