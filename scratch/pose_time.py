import torch, time, sys
sys.path.insert(0,'.')
from differentiable_ransac_amd import ops, synth
dev='cuda'; P,N,B=32,2000,1024
d=synth.batch_two_view(P,N); m=d['matches'].to(dev)
r=ops.gumbel_topk(d['logits'].to(dev),B,5,1.0,None,seed=1)
models,valid=ops.solve_nister5(ops.gather(m,r['idx'],r['y_sel']))
gt=d['gt_E'].to(dev)
chosen,which=ops.select_closest_autograd(models,valid,gt)
chosen=chosen.detach().clone().requires_grad_(True)
R,t=d['R'].to(dev),d['t'].to(dev)
def f():
    eq,et,_,_=ops.pose_error(m,chosen,R,t); return eq,et
f(); torch.cuda.synchronize(); a=time.perf_counter()
for _ in range(10): eq,et=f()
torch.cuda.synchronize(); ms=(time.perf_counter()-a)/10*1e3
print('pose_error fwd P=32 M=1024 N=2000: %.3f ms  (%.1f G model x point /s)'%(ms,P*B*N/ms/1e6))
for i in range(4):
    eq,et=f(); l=((eq+et)/2).mean(); torch.cuda.synchronize(); a=time.perf_counter(); l.backward(); torch.cuda.synchronize()
    print('bwd %.3f ms'%((time.perf_counter()-a)*1e3), float(l), bool(torch.isfinite(chosen.grad).all()))
