"""K3 A/B: times dr_solve_nister5_f32 on synthetic two-view samples (C2 x 32 pairs shape: 32768 samples) with whichever
library DRANSAC_LIB names, and dumps the models/valid so two runs can be compared.
    DRANSAC_LIB=... python scratch/k3_ab.py out.npz ; python scratch/k3_ab.py cmp a.npz b.npz"""
import sys
import numpy as np

if sys.argv[1] == "cmp":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    va, vb = a["valid"], b["valid"]
    print("valid a", va.sum(), "valid b", vb.sum(), "differ", (va != vb).sum())
    both = va & vb
    d = np.abs(a["models"] - b["models"]).reshape(*va.shape, 9).max(-1)
    print("max |dE| where both valid", d[both].max(), " >1e-6:", (d[both] > 1e-6).sum(), " bit-identical:", (d[both] == 0).mean())
    for k in ("a", "b"):
        z = (a if k == "a" else b)
        print(k, "ms", z["ms"], " best-solution error vs E_gt: median", np.median(z["err"]), " p99", np.percentile(z["err"], 99),
              " frac<1e-4", (z["err"] < 1e-4).mean())
    sys.exit(0)

import torch
sys.path.insert(0, ".")
from differentiable_ransac_amd import ops

torch.manual_seed(3)
dev = "cuda:0"
B = 32768
g = torch.Generator(device="cpu").manual_seed(11)
# exact two-view geometry: random rotation (small), translation, 5 points in front of both cameras
def rot(w):
    th = w.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    k = w / th
    K = torch.zeros(w.shape[0], 3, 3, dtype=w.dtype)
    K[:, 0, 1], K[:, 0, 2], K[:, 1, 0], K[:, 1, 2], K[:, 2, 0], K[:, 2, 1] = -k[:, 2], k[:, 1], k[:, 2], -k[:, 0], -k[:, 1], k[:, 0]
    th = th[..., None]
    return torch.eye(3, dtype=w.dtype) + torch.sin(th) * K + (1 - torch.cos(th)) * (K @ K)
w = torch.randn(B, 3, generator=g, dtype=torch.float64) * 0.3
R = rot(w)
t = torch.randn(B, 3, generator=g, dtype=torch.float64)
t = t / t.norm(dim=-1, keepdim=True)
X = torch.rand(B, 5, 3, generator=g, dtype=torch.float64) * 2 - 1
X[..., 2] += 4.0
x1 = X[..., :2] / X[..., 2:]
X2 = X @ R.transpose(1, 2) + t[:, None]
x2 = X2[..., :2] / X2[..., 2:]
tx = torch.zeros(B, 3, 3, dtype=torch.float64)
tx[:, 0, 1], tx[:, 0, 2], tx[:, 1, 0], tx[:, 1, 2], tx[:, 2, 0], tx[:, 2, 1] = -t[:, 2], t[:, 1], t[:, 2], -t[:, 0], -t[:, 1], t[:, 0]
Egt = tx @ R
Egt = Egt / Egt.flatten(1).norm(dim=-1)[:, None, None]
s = torch.cat([x1, x2], -1).float().to(dev)
for _ in range(20):
    models, valid = ops.solve_nister5(s)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ts = []
for _ in range(5):
    e0.record()
    for _ in range(50):
        models, valid = ops.solve_nister5(s)
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 50)
ms = sorted(ts)[2]
m = models.double().cpu()
m = m / m.flatten(2).norm(dim=-1).clamp_min(1e-30)[..., None, None]
d = torch.minimum((m - Egt[:, None]).flatten(2).norm(dim=-1), (m + Egt[:, None]).flatten(2).norm(dim=-1))
d = torch.where(valid.cpu(), d, torch.full_like(d, 9.0))
err = d.min(dim=1).values.numpy()
print("K3 ms", ms, "valid", int(valid.sum()), "median err", float(np.median(err)), "frac<1e-4", float((err < 1e-4).mean()))
np.savez(sys.argv[1], models=models.cpu().numpy(), valid=valid.cpu().numpy(), ms=ms, err=err)
