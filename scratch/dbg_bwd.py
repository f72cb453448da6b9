import torch, sys
sys.path.insert(0, '.')
from oracle import cpu_ref as O
from differentiable_ransac_amd import ops, synth
dev='cuda'
pair = synth.two_view_pair(90, 200, inlier_ratio=1.0, noise=2e-3, dtype=torch.float64)
B = 12
smp = pair["matches"][: 5 * B].reshape(B, 5, 4).contiguous()
b = 3
x = smp[b].float().double()
I = torch.eye(3, dtype=torch.float64)
def skew(a): return torch.tensor([[0,-a[2],a[1]],[a[2],0,-a[0]],[-a[1],a[0],0]],dtype=torch.float64)
def analytic(Em, g):
    x1=torch.cat((x[:,:2],torch.ones(5,1,dtype=torch.float64)),1); x2=torch.cat((x[:,2:],torch.ones(5,1,dtype=torch.float64)),1)
    J=[skew(I[i])@Em for i in range(3)]+[Em@skew(I[i]) for i in range(3)]
    AJ=torch.stack([torch.stack([x2[k]@J[c]@x1[k] for c in range(6)]) for k in range(5)])
    Jg=torch.stack([(J[c]*g).sum() for c in range(6)])
    lam=torch.linalg.solve(AJ@AJ.T,AJ@Jg)
    gx=torch.zeros(5,4,dtype=torch.float64)
    for k in range(5):
        gx[k,:2]=-lam[k]*(Em.T@x2[k])[:2]; gx[k,2:]=-lam[k]*(Em@x1[k])[:2]
    return gx
Eo, ok, real = O.nister_5pt(x[None])
so = O.canonical(Eo[0][real[0]])
for slot in range(10):
    s32 = smp.float().to(dev).requires_grad_(True)
    E, valid = ops.solve_essential(s32, None, "nister")
    if not bool(valid[b, slot]): continue
    W = torch.zeros(B, 10, 3, 3); g = torch.randn(3, 3, generator=torch.Generator().manual_seed(slot)); W[b, slot] = g
    (E * W.to(dev)).sum().backward()
    gg = s32.grad[b].cpu().double()
    Em = E[b, slot].detach().cpu().double()
    ga = analytic(Em, g.double())
    d = (O.canonical(Em)[None] - so).abs().amax((-1,-2)).min()
    print('slot', slot, 'dist to oracle sol', float(d), 'gpu-vs-analytic rel', float((gg-ga).abs().max()/ga.abs().max()), 'gmax', float(ga.abs().max()))
