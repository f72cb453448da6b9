"""Randomised cross-checks of the round-3 kernels against their general counterparts / the oracle (not part of the test suite:
run once per change on the GPU box, `python scratch/fuzz_r3.py [trials]`)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from differentiable_ransac_amd import ops, synth
from oracle import cpu_ref as O

dev = "cuda"
T = int(sys.argv[1]) if len(sys.argv) > 1 else 20
g = torch.Generator().manual_seed(2026)
ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
fails = 0


def check(name, ok, info=""):
    global fails
    if not ok:
        fails += 1
        print("FAIL", name, info)


# K4 short rows vs the f64 oracle
for t in range(T):
    N, M, P = ri(1, 256), ri(1, 70), ri(1, 4)
    b = synth.batch_two_view(P, max(N, 8), seed0=5000 + t)
    m = b["matches"][:, :N].contiguous()
    models = b["gt_E"][:, None] + 0.05 * torch.randn(P, M, 3, 3, generator=g)
    valid = torch.rand(P, M, generator=g) > 0.3
    thr = 7.5e-4
    s, k = ops.msac_score(m.to(dev), models.to(dev), thr, True, valid.to(dev))
    for p in range(P):
        rs, rm = O.msac_score(m[p].double(), models[p].double(), thr)
        v = valid[p]
        err = (s[p].cpu().double() - rs).abs()[v]
        check("k4small score", bool((err <= 1e-4 * rs[v].abs().clamp(min=1)).all()), (N, M, float(err.max()) if v.any() else 0))
        a = models[p].double().transpose(-1, -2) @ torch.cat((m[p, :, 2:].double(), torch.ones(N, 1, dtype=torch.float64)), 1).T
        bb = models[p].double() @ torch.cat((m[p, :, :2].double(), torch.ones(N, 1, dtype=torch.float64)), 1).T
        r = (torch.cat((m[p, :, :2].double(), torch.ones(N, 1, dtype=torch.float64)), 1).T[None] * a).sum(-2)
        d2 = r ** 2 / (a[:, 0] ** 2 + a[:, 1] ** 2 + bb[:, 0] ** 2 + bb[:, 1] ** 2)
        thr2 = (1.5 * thr) ** 2
        near = (d2 - thr2).abs() <= 2e-5 * thr2
        diff = (k[p].cpu() != rm) & ~near
        check("k4small mask", int(diff[v].sum()) == 0, (N, M))
        check("k4small invalid", bool((s[p].cpu()[~v] == 0).all()) and not bool(k[p].cpu()[~v].any()), (N, M))

# long-row sampler: one-pass kernel (split / unsplit) vs the general kernel
for t in range(T):
    N = 4 * ri(520, 16000)
    B, P, k = ri(1, 200), ri(1, 3), ri(1, 5)
    lg = (torch.randn(P, N, generator=g) + 3.0 * (torch.rand(P, N, generator=g) > 0.5)).to(dev)
    seed = ri(0, 2 ** 40)
    one = ops.gumbel_topk(lg, B, k, 1.0, None, seed=seed)
    gen = ops.gumbel_topk(lg, B, k, 1.0, None, seed=seed, want_noise=True)
    check("stream idx", torch.equal(one["idx"], gen["idx"]), (N, B, P, k))
    check("stream soft", torch.allclose(one["y_sel"], gen["y_sel"], rtol=5e-6, atol=1e-9) and
          torch.allclose(one["lse"], gen["lse"], rtol=2e-6, atol=2e-6), (N, B, P, k))

# rigid residual 16-point kernel vs the general kernel
for t in range(T):
    N, M = 16 * ri(1, 900), ri(1, 130)
    rp = synth.rigid_pair(t, N)
    models = torch.eye(4).repeat(M, 1, 1)
    models[:, :3, :] = rp["gt_T"][:3, :].float()[None] + 0.03 * torch.randn(M, 3, 4, generator=g)
    pts = rp["matches"].float().to(dev)[None]
    res, masks = ops.rigid_residual(pts, models.to(dev)[None], 0.03, True)
    res_g, _ = ops.rigid_residual(pts, models.to(dev)[None], 0.03, False)
    check("k4r sums", float(((res - res_g).abs() / res_g).max()) < 3e-5, (N, M))
    Td = models.double()
    p3, q3 = rp["matches"][:, :3].double(), rp["matches"][:, 3:].double()
    d2 = ((q3[None] - (p3[None] @ Td[:, :3, :3].transpose(-1, -2) + Td[:, None, :3, 3])) ** 2).sum(-1)
    near = (d2 - 0.03).abs() < 1e-6
    check("k4r masks", bool(((masks[0].cpu() == (d2 < 0.03)) | near).all()), (N, M))

# MatchLoss residual kernels vs the oracle at random shapes
for t in range(T):
    P, N, M = ri(1, 3), ri(9, 5000), ri(1, 80)
    data = synth.batch_two_view(P, N, seed0=7000 + t)
    models = data["gt_E"][:, None] + 0.05 * torch.randn(P, M, 3, 3, generator=g)
    mask = torch.rand(P, N, generator=g) < 0.6
    mask[:, 0] = True
    valid = torch.rand(P, M, generator=g) > 0.2
    sums = ops.episym_sums(data["matches"].to(dev), mask.to(dev), models.to(dev), valid.to(dev))
    for p in range(P):
        ys = O.episym(data["matches"][p, mask[p], :2].double(), data["matches"][p, mask[p], 2:].double(), models[p].double())
        ref = torch.clamp(ys, max=1.0).sum(1)
        got = sums[p].cpu().double()
        check("episym", float(((got - ref).abs()[valid[p]] / ref[valid[p]].clamp(min=1e-12)).max()) < 2e-4 if valid[p].any() else True,
              (P, N, M))
        check("episym invalid", bool((got[~valid[p]] == 0).all()), (P, N, M))

# K6 of the 3-D path vs torch
for t in range(T):
    P, M, N = ri(1, 6), ri(1, 300), ri(1, 9000)
    pts = torch.rand(P, N, 6, generator=g).to(dev)
    models = torch.eye(4).repeat(P, M, 1, 1)
    models[:, :, :3, 3] = 0.2 * torch.randn(P, M, 3, generator=g)
    models = models.to(dev)
    valid = (torch.rand(P, M, generator=g) < 0.7).to(dev)
    res = (torch.rand(P, M, generator=g) * 5).to(dev)
    mask = torch.ones(P, N, dtype=torch.bool, device=dev)
    nb, nm, idx = ops.ransac3d_update(pts, models, valid, res, 0.08, None, None, mask)
    key = torch.where(valid, res, torch.full_like(res, float("inf")))
    val, _ = key.min(1)
    for p in range(P):
        if torch.isfinite(val[p]):
            first = int((key[p] == val[p]).nonzero()[0])
            check("r3d idx", int(idx[p]) == first, (P, M, N))
            Tm = models[p, first].cpu()
            d2 = ((pts[p, :, 3:].cpu() - (pts[p, :, :3].cpu() @ Tm[:3, :3].T + Tm[:3, 3])) ** 2).sum(-1)
            check("r3d mask", int((mask[p].cpu() != (d2 < 0.08)).sum()) <= 2, (P, M, N))
        else:
            check("r3d none", int(idx[p]) == -1 and not bool(mask[p].any()), (P, M, N))

print("fuzz done:", fails, "failure(s) in", T, "trials per family")
sys.exit(1 if fails else 0)
