"""Randomised cross-checks of the round-4 kernel paths (not part of the test suite: run on the GPU box,
`python scratch/fuzz_r4.py [trials]`): K4 with 16-slot halves (few pairs) vs the f64 oracle at random shapes incl. rows split over
blocks and invalid / NaN slots; the screened long-row sampler vs the unscreened one; f64 sampler backward vs autograd."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from differentiable_ransac_amd import ops, synth
from oracle import cpu_ref as O

dev = "cuda"
T = int(sys.argv[1]) if len(sys.argv) > 1 else 20
g = torch.Generator().manual_seed(2027)
ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
fails = 0


def check(name, ok, info=""):
    global fails
    if not ok:
        fails += 1
        print("FAIL", name, info)


# K4, 16-point kernel, grids below and above the small-grid switch (P * ceil(M / 128) < 512), rows of 1-3 chunks
for t in range(T):
    N = 16 * ri(17, 380)
    M = ri(1, 700)
    P = ri(1, 6)
    b = synth.batch_two_view(P, N, seed0=7000 + t)
    m = b["matches"]
    models = b["gt_E"][:, None] + 0.05 * torch.randn(P, M, 3, 3, generator=g)
    if M > 3:
        models[0, 1, 0, 0] = float("nan")
        models[0, 2] = 0.0
    valid = torch.rand(P, M, generator=g) > 0.4
    thr = 7.5e-4 * (1 + ri(0, 3))
    s, k = ops.msac_score(m.to(dev), models.to(dev), thr, True, valid.to(dev))
    s2, _ = ops.msac_score(m.to(dev), models.to(dev), thr, False, valid.to(dev))
    check("k4 nomask == mask scores", bool(torch.allclose(s.nan_to_num(-1), s2.nan_to_num(-1), rtol=1e-6, atol=1e-6)), (N, M, P))
    for p in range(P):
        rs, rm = O.msac_score(m[p].double(), models[p].double(), thr)
        fin = torch.isfinite(models[p]).flatten(1).all(1) & (models[p] != 0).flatten(1).any(1)
        v = valid[p] & fin
        err = (s[p].cpu().double() - rs).abs()[v]
        check("k4 score", bool((err <= 1e-4 * rs[v].abs().clamp(min=1)).all()), (N, M, P, float(err.max()) if v.any() else 0))
        one = torch.ones(N, 1, dtype=torch.float64)
        h1, h2 = torch.cat((m[p, :, :2].double(), one), 1), torch.cat((m[p, :, 2:].double(), one), 1)
        a = models[p].double().transpose(-1, -2) @ h2.T
        bb = models[p].double() @ h1.T
        r = (h1.T[None] * a).sum(-2)
        d2 = r ** 2 / (a[:, 0] ** 2 + a[:, 1] ** 2 + bb[:, 0] ** 2 + bb[:, 1] ** 2)
        thr2 = (1.5 * thr) ** 2
        near = (d2 - thr2).abs() <= 2e-5 * thr2
        diff = (k[p].cpu() != rm) & ~near
        check("k4 mask", int(diff[v].sum()) == 0, (N, M, P))
        check("k4 invalid", bool((s[p].cpu()[~valid[p]] == 0).all()) and not bool(k[p].cpu()[~valid[p]].any()), (N, M, P))
        bad = valid[p] & ~fin
        check("k4 nan", bool(torch.isnan(s[p].cpu()[bad]).all()) and not bool(k[p].cpu()[bad].any()), (N, M, P))

# long rows, index sets only: screened == unscreened, every launch shape (split / one wave per row), k = 1..5
for t in range(T):
    N = 4 * ri(520, 16000)
    B, P, k = ri(64, 300), ri(1, 3), ri(1, 5)
    if t % 4 == 0:
        B = ri(1500, 2500)     # more than 4096 rows with P = 3: a wave per row
    spread = [0.1, 1.0, 3.0, 8.0][t % 4]
    lg = (spread * torch.randn(P, N, generator=g)).to(dev)
    seed = ri(0, 2 ** 40)
    a = ops.gumbel_topk(lg, B, k, 1.0, None, seed=seed, soft=False, screen=False)["idx"]
    b = ops.gumbel_topk(lg, B, k, 1.0, None, seed=seed, soft=False, screen=True)["idx"]
    check("screen idx", torch.equal(a, b), (N, B, P, k, spread))

# f64 sampler + gather backward vs autograd of the dense formula
for t in range(max(3, T // 4)):
    P, B, N, k = ri(1, 3), ri(1, 40), ri(8, 600), ri(1, 5)
    matches = torch.randn(P, N, 4, generator=g, dtype=torch.float64)
    logits = torch.randn(P, N, generator=g, dtype=torch.float64)
    u = torch.rand(P, B, N, generator=g, dtype=torch.float64).clamp(1e-12, 1 - 1e-12)
    noise = -torch.log(-torch.log(u))
    gs = torch.randn(P, B, k, 4, generator=g, dtype=torch.float64)
    lg = logits.to(dev).requires_grad_(True)
    samples, w, idx = ops.SampleGather.apply(matches.to(dev), lg, B, k, 1.0, noise.to(dev), 0)
    (samples * gs.to(dev)).sum().backward()
    lr = logits.clone().requires_grad_(True)
    y = torch.softmax(lr[:, None, :] + noise, -1)
    ii = idx.long().cpu()
    ysel = torch.gather(y, 2, ii)
    st = (1.0 - ysel.detach()) + ysel
    pts = torch.gather(matches[:, None].expand(P, B, N, 4), 2, ii[..., None].expand(P, B, k, 4)) * st[..., None]
    (pts * gs).sum().backward()
    check("f64 bwd", float((lg.grad.cpu() - lr.grad).abs().max()) <= 1e-10 * max(1.0, float(lr.grad.abs().max())), (P, B, N, k))

print("fuzz_r4: %d failures in %d trials per family" % (fails, T))
