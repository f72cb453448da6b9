import torch, time, sys
sys.path.insert(0,'.')
from differentiable_ransac_amd import synth
from differentiable_ransac_amd.ransac import BatchedRANSAC
dev='cuda'; P,N,B=32,2000,1024
for rho in (0.5, 0.25):
    d=synth.batch_two_view(P,N,inlier_ratio=rho); m=d['matches'].to(dev); lg=torch.zeros_like(d['logits']).to(dev); K1=d['K1'].to(dev); K2=d['K2'].to(dev)
    for pipe in (False, True):
        rn=BatchedRANSAC('nister',ransac_batch_size=B,threshold=0.75,max_iterations=5000,refit=True)
        rn.pipeline=pipe
        for _ in range(3): out=rn(m,lg,K1,K2)
        torch.cuda.synchronize(); t0=time.perf_counter()
        for _ in range(10): out=rn(m,lg,K1,K2)
        torch.cuda.synchronize(); ms=(time.perf_counter()-t0)*100
        print(f'inlier ratio {rho} pipeline={pipe}: {ms:.3f} ms/call, iterations min/max {int(out["iterations"].min())}/{int(out["iterations"].max())}, inliers {float(out["inliers"].float().mean()):.0f}')
