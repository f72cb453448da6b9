"""K4 / K6 parity on the GPU: HIP kernel vs the CPU oracle and vs the reference's golden vectors."""
import pytest
import torch

from oracle import cpu_ref as O
from tests.conftest import load_golden

pytestmark = pytest.mark.gpu


def _check_scores_masks(scores, masks, ref_scores, ref_masks, d2_margin):
    # |dscore| <= 1e-4 * max(1, |score|); masks equal except within a few ulp of the threshold
    err = (scores.double().cpu() - ref_scores.double()).abs()
    tol = 1e-4 * ref_scores.double().abs().clamp(min=1.0)
    assert (err <= tol).all(), float((err / tol).max())
    diff = masks.cpu() != ref_masks
    assert (diff & ~d2_margin).sum() == 0, int((diff & ~d2_margin).sum())


def _margin(matches, models, thr, rel=1e-5):
    """points whose d2 is within rel of the squared threshold (mask may legitimately differ there)"""
    m64, md64 = matches.double(), models.double()
    thr2 = (1.5 * thr) ** 2
    n = m64.shape[0]
    one = torch.ones(n, 1, dtype=torch.float64)
    h1 = torch.cat((m64[:, :2], one), -1)
    h2 = torch.cat((m64[:, 2:], one), -1)
    a = md64.transpose(-1, -2) @ h2.T
    b = md64 @ h1.T
    r = (h1.T[None] * a).sum(-2)
    d2 = r ** 2 / (a[:, 0] ** 2 + a[:, 1] ** 2 + b[:, 0] ** 2 + b[:, 1] ** 2)
    return (d2 - thr2).abs() <= rel * thr2


def test_msac_golden(dev):
    from differentiable_ransac_amd.scorings import MSACScore
    g = load_golden("msac")
    for tag, dt in (("f32", torch.float32), ("f64", torch.float64)):
        m, md = g["matches"].to(dt), g["models"].to(dt)
        s, k = MSACScore("cuda").score(m.to(dev), md.to(dev), g["threshold"])
        assert k.dtype == torch.bool and k.shape == (48, 256)
        _check_scores_masks(s, k, g[f"scores_{tag}"], g[f"masks_{tag}"], _margin(m, md, g["threshold"]))


@pytest.mark.parametrize("N,M", [(2000, 1024), (2000, 37), (131, 5), (1, 1), (2049, 64), (5000, 40), (7, 33),
                                 (4096, 40), (6144, 200), (2048, 129)])   # 16-point kernel with the point range split over blocks
def test_msac_vs_oracle_shapes(dev, N, M):
    from differentiable_ransac_amd import ops, synth
    pair = synth.two_view_pair(100 + N, max(N, 8))
    matches = pair["matches"][:N].contiguous()
    gen = torch.Generator().manual_seed(M)
    models = pair["gt_E"][None] + 0.05 * torch.randn(M, 3, 3, generator=gen)
    models[0] = pair["gt_E"]
    thr = 7.5e-4
    rs, rm = O.msac_score(matches.double(), models.double(), thr)
    s, k = ops.msac_score(matches[None].to(dev), models[None].to(dev), thr)
    _check_scores_masks(s[0], k[0], rs, rm, _margin(matches, models, thr, 2e-5))
    s2, none = ops.msac_score(matches[None].to(dev), models[None].to(dev), thr, want_masks=False)
    assert none is None
    assert torch.allclose(s2, s, rtol=1e-6, atol=1e-6)


def test_msac_batched_pairs_and_nan_models(dev):
    from differentiable_ransac_amd import ops, synth
    P, N, M = 5, 512, 70
    b = synth.batch_two_view(P, N, seed0=300)
    gen = torch.Generator().manual_seed(1)
    models = b["gt_E"][:, None] + 0.02 * torch.randn(P, M, 3, 3, generator=gen)
    models[1, 3, 1, 1] = float("nan")
    models[2, 0] = float("inf")
    thr = torch.tensor([7.5e-4, 1e-3, 5e-4, 7.5e-4, 2e-3])
    s, k = ops.msac_score(b["matches"].to(dev), models.to(dev), thr.to(dev))
    for p in range(P):
        rs, rm = O.msac_score(b["matches"][p].double(), models[p].double(), float(thr[p]))
        ok = torch.isfinite(models[p]).all(-1).all(-1)
        _check_scores_masks(s[p][ok], k[p][ok], rs[ok], rm[ok],
                            _margin(b["matches"][p], models[p], float(thr[p]), 2e-5)[ok])
        assert torch.isnan(s[p][~ok]).all() and not k[p][~ok].any()
    # K6: arg-max, best mask, inlier count
    bi, bs, bm, bmask, inl = ops.select_best(b["matches"].to(dev), models.to(dev), s, thr.to(dev))
    for p in range(P):
        sc = s[p].cpu().clone()
        sc[torch.isnan(sc)] = -1
        assert int(bi[p]) == int(sc.argmax())
        assert float(bs[p]) == float(sc.max())
        assert torch.equal(bm[p].cpu(), models[p, int(bi[p])])
        assert torch.equal(bmask[p], k[p, int(bi[p])])
        assert int(inl[p]) == int(bmask[p].sum())


def test_msac_linearity_property_full_size(dev):
    """size-independent property at the benchmark shape: scaling a model leaves scores and masks unchanged,
    and score <= number of inliers <= N."""
    from differentiable_ransac_amd import ops, synth
    P, N, M = 2, 2000, 10240
    b = synth.batch_two_view(P, N, seed0=400)
    gen = torch.Generator().manual_seed(2)
    models = (b["gt_E"][:, None] + 0.05 * torch.randn(P, M, 3, 3, generator=gen)).to(dev)
    mt = b["matches"].to(dev)
    s1, k1 = ops.msac_score(mt, models, 7.5e-4)
    s2, k2 = ops.msac_score(mt, models * 4.0, 7.5e-4)   # power-of-two scale: bit-exact invariance
    assert torch.equal(s1, s2) and torch.equal(k1, k2)
    cnt = k1.sum(-1)
    assert (s1 <= cnt + 1e-3).all() and (cnt <= N).all()
    assert (s1[:, 0] >= 0).all()


def test_msac_valid_slots_are_skipped(dev):
    from differentiable_ransac_amd import ops, synth
    P, N, M = 3, 2000, 100
    b = synth.batch_two_view(P, N, seed0=500)
    gen = torch.Generator().manual_seed(3)
    models = (b["gt_E"][:, None] + 0.02 * torch.randn(P, M, 3, 3, generator=gen)).to(dev)
    valid = (torch.rand(P, M, generator=gen) > 0.5).to(dev)
    s_all, k_all = ops.msac_score(b["matches"].to(dev), models, 7.5e-4)
    s, k = ops.msac_score(b["matches"].to(dev), models, 7.5e-4, valid=valid)
    assert torch.equal(s[valid], s_all[valid]) and torch.equal(k[valid], k_all[valid])
    assert (s[~valid] == 0).all() and not k[~valid].any()
    s64, k64 = ops.msac_score(b["matches"].double().to(dev), models.double(), 7.5e-4, valid=valid)
    assert (s64[~valid] == 0).all() and not k64[~valid].any()
    assert torch.allclose(s64[valid].float(), s[valid], rtol=1e-4, atol=1e-4)
