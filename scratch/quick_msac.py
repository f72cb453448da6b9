import torch, time
from differentiable_ransac_amd import ops, synth
dev='cuda'
P,N,M=32,2000,10240
b=synth.batch_two_view(P,N)
models=(b['gt_E'][:,None]+0.05*torch.randn(P,M,3,3)).to(dev)
mt=b['matches'].to(dev)
for wm in (True,False):
    for _ in range(3): ops.msac_score(mt,models,7.5e-4,wm)
    torch.cuda.synchronize(); t=time.time()
    for _ in range(20): ops.msac_score(mt,models,7.5e-4,wm)
    torch.cuda.synchronize(); dt=(time.time()-t)/20
    by=P*(16*N+40*M+(M*N if wm else 0))
    print('masks',wm,'ms',dt*1e3,'GB/s',by/dt/1e9,'TFLOP/s',39*P*M*N/dt/1e12)
