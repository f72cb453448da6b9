import torch, time, sys
sys.path.insert(0,'.')
from differentiable_ransac_amd import ops, synth
dev='cuda'
def t(fn,reps=10):
    fn(); torch.cuda.synchronize(); a=time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-a)/reps*1e3
# config 4: rigid, N=50000, B=2048, P=1
rp=synth.rigid_pair(1,50000); m=rp['matches'][None].to(dev); lg=rp['logits'][None].to(dev)
def c4():
    r=ops.gumbel_topk(lg,2048,3,1.0,None,seed=1)
    s=ops.gather(m,r['idx'],r['y_sel'])
    mod,R,tt,sc,v=ops.solve_rigid(s[0],flag=False)
    res,mk=ops.rigid_residual(m,mod[None],0.03,True)
    return res
ms=t(c4); print('config4 rigid N=50000 B=2048: %.3f ms -> %.2f M hyps/s'%(ms,2048/ms/1e3))
print('  K1 %.3f  K3r %.3f  K4r %.3f'%(t(lambda: ops.gumbel_topk(lg,2048,3,1.0,None,seed=1)), t(lambda: ops.solve_rigid(torch.rand(2048,3,6,device=dev),flag=False)), t(lambda: ops.rigid_residual(m,torch.eye(4,device=dev).repeat(1,2048,1,1),0.03,True))))
# config 1: 8-pt F, N=128, B=64, uniform, P=1 and P=256
for P in (1,256):
    d=synth.batch_two_view(P,128,pixel=True); mm=d['matches'].to(dev)
    def c1():
        idx=ops.uniform_sample(P,64,8,128,seed=3,device=dev)
        s=ops.gather(mm,idx,None)
        F,v=ops.solve_f8(s)
        sc,mk=ops.msac_score(mm,F,0.75,True,v)
        return sc
    ms=t(c1); print('config1 f8 N=128 B=64 P=%d: %.3f ms -> %.2f M hyps/s'%(P,ms,P*64/ms/1e3))
