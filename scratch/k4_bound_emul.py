"""How selective is the Cauchy-Schwarz bound J <= |A|_F^2 |x2h|^2 + |B|_F^2 |x1h|^2 as an outlier pre-test for K4?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from differentiable_ransac_amd import synth
from oracle import cpu_ref as O
torch.manual_seed(0)
P, N, B = 2, 2000, 192
for p in range(P):
    d = synth.batch_two_view(1, N, seed0=p)
    m = d['matches'][0].double()
    noise = synth.gumbel_noise((1, B, N), seed=3 + p)[0]
    idx, ret, _ = O.gumbel_topk(d['logits'][0], noise, 1.0, 5)
    smp = O.gather_samples(d['matches'][0], ret)
    E, ok, real = O.nister_5pt(smp.double())
    Ms = E.reshape(-1, 3, 3)[real.reshape(-1)]
    thr = O.normalized_threshold(0.75, d['K1'][0], d['K2'][0], False)
    t = 1.5 * float(thr)
    x1 = torch.cat([m[:, :2], torch.ones(N, 1, dtype=torch.float64)], 1)
    x2 = torch.cat([m[:, 2:], torch.ones(N, 1, dtype=torch.float64)], 1)
    a = torch.einsum('mij,ni->mnj', Ms, x2)          # M^T x2 : a_j = sum_i M_ij x2_i
    b = torch.einsum('mij,nj->mni', Ms, x1)          # M x1
    r = (a * x1[None]).sum(-1)
    J = a[..., 0] ** 2 + a[..., 1] ** 2 + b[..., 0] ** 2 + b[..., 1] ** 2
    cA = (Ms[:, :, :2] ** 2).sum((1, 2))             # columns 0,1
    cB = (Ms[:, :2, :] ** 2).sum((1, 2))             # rows 0,1
    q1 = (x1 ** 2).sum(-1); q2 = (x2 ** 2).sum(-1)
    Jub = cA[:, None] * q2[None] + cB[:, None] * q1[None]
    inl = r ** 2 < t * t * J
    cand = r ** 2 < 1.001 * t * t * Jub
    assert (inl & ~cand).sum() == 0
    per_lane = cand.reshape(Ms.shape[0], N // 16, 16).sum(-1).double()
    per_wave = cand[:, :1024].reshape(Ms.shape[0], -1).sum(-1).double()
    print(f'pair {p}: models {Ms.shape[0]}  inlier fraction {inl.double().mean():.4f}  candidate fraction {cand.double().mean():.4f}  '
          f'J/Jub median {float((J / Jub).median()):.3f}  candidates per lane(16 pts): mean {per_lane.mean():.2f} max {per_lane.max():.0f}  '
          f'per wave (1024 pts): mean {per_wave.mean():.1f} p99 {per_wave.quantile(0.99):.0f} max {per_wave.max():.0f}')
