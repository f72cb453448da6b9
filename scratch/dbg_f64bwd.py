import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from differentiable_ransac_amd import ops, synth
dev='cuda'
pair = synth.two_view_pair(41, 400, inlier_ratio=1.0, noise=0.0, dtype=torch.float64)
smp0 = pair["matches"][:40].reshape(8, 5, 4).to(dev)
wgt = torch.randn(3, 3, generator=torch.Generator().manual_seed(4), dtype=torch.float64).to(dev)
gt = pair["gt_E"].to(dev)
def picked(smp):
    E, valid = ops.solve_essential(smp, None, "nister")
    d = torch.minimum(((E - gt) ** 2).sum((-1, -2)), ((E + gt) ** 2).sum((-1, -2)))
    d = torch.where(valid, d, torch.full_like(d, 1e9))
    j = d.argmin(-1)
    Ej = E[torch.arange(E.shape[0], device=dev), j]
    sgn = torch.sign((Ej * gt).sum((-1, -2)))
    return Ej * sgn[:, None, None], d.min(-1).values, j
smp = smp0.clone().requires_grad_(True)
Ej, dist, j = picked(smp)
print("dist", dist.tolist(), "slot", j.tolist())
(Ej * wgt).sum().backward()
g = smp.grad
print("grad scale per sample", g.abs().amax((1, 2)).tolist())
# f32-I/O kernel on the same
s32 = smp0.float().clone().requires_grad_(True)
E32, v32 = ops.solve_essential(s32, None, "nister")
d = torch.minimum(((E32 - gt.float()) ** 2).sum((-1, -2)), ((E32 + gt.float()) ** 2).sum((-1, -2)))
d = torch.where(v32, d, torch.full_like(d, 1e9)); j32 = d.argmin(-1)
Ej32 = E32[torch.arange(8, device=dev), j32]; sg = torch.sign((Ej32 * gt.float()).sum((-1, -2)))
((Ej32 * sg[:, None, None]) * wgt.float()).sum().backward()
print("f64 vs f32-I/O kernel: max rel diff per sample", ((g - s32.grad.double()).abs().amax((1, 2)) / g.abs().amax((1, 2))).tolist())
for h in (1e-4, 1e-6, 1e-8):
    for (b, k, c) in ((0, 0, 0), (1, 2, 3), (3, 4, 1), (5, 1, 2), (7, 3, 0)):
        sp, sm = smp0.clone(), smp0.clone(); sp[b, k, c] += h; sm[b, k, c] -= h
        fd = float(((picked(sp)[0] - picked(sm)[0]) * wgt).sum()) / (2 * h)
        print(h, (b, k, c), "fd", fd, "g", float(g[b, k, c]))
