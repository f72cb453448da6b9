"""Ragged / tiny / extreme inputs through every kernel (the reference has no tests; these are the edge cases its code
paths imply: empty tails, N not a multiple of the vector width, a single pair, a single hypothesis, huge logits)."""
import pytest
import torch

from oracle import cpu_ref as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("P,N,B", [(1, 5, 1), (3, 37, 5), (2, 1023, 33), (1, 2050, 7), (4, 16, 65)])
def test_batched_pipeline_odd_sizes_match_oracle(dev, P, N, B):
    from differentiable_ransac_amd import ops, synth
    data = synth.batch_two_view(P, N, seed0=800 + N)
    m, lg = data["matches"].to(dev), data["logits"].to(dev)
    noise = synth.gumbel_noise((P, B, N), seed=N)
    r = ops.gumbel_topk(lg, B, 5, 1.0, noise.to(dev), dense=True)
    smp = ops.gather(m, r["idx"], r["y_sel"])
    E, valid = ops.solve_nister5(smp)
    flat, vflat = E.reshape(P, B * 10, 3, 3), valid.reshape(P, B * 10)
    thr = torch.linspace(5e-4, 2e-3, P).to(dev)
    sc, mk = ops.msac_score(m, flat, thr, True, vflat)
    sc2, _ = ops.msac_score(m, flat, thr, False, vflat)
    assert torch.allclose(sc, sc2, rtol=1e-6, atol=1e-6)
    st = ops.RansacState(P, N, 5000, dev, torch.float32)
    ops.ransac_update(st, m, flat, vflat, sc, thr, B, 5)
    for p in range(P):
        idx, ret, ys = O.gumbel_topk(data["logits"][p], noise[p], 1.0, 5)
        assert torch.equal(r["idx"][p].cpu().long(), idx)
        assert torch.equal(r["ret"][p].cpu() != 0, ret != 0)
        Eo, ok, real = O.nister_5pt(O.gather_samples(data["matches"][p], ret).double())
        n_hip, n_or = int(valid[p].sum()), int(real[ok].sum())
        assert abs(n_hip - n_or) <= max(2, 0.02 * n_or)
        so, mo = O.msac_score(data["matches"][p].double(), flat[p].cpu().double(), float(thr[p]))
        v = vflat[p].cpu()
        if v.any():
            assert ((sc[p].cpu().double() - so).abs()[v] <= 1e-4 * so.abs().clamp(min=1)[v]).all()
            assert (mk[p].cpu()[v] != mo[v]).float().mean() < 1e-3
            best = int(torch.where(v, so, torch.full_like(so, -1)).argmax())
            assert abs(float(st.best_score[p]) - float(so[best])) <= 1e-4 * max(1.0, float(so[best]))
        assert (sc[p].cpu()[~v] == 0).all() and not mk[p].cpu()[~v].any()
        assert int(st.iters[p]) == B


def test_extreme_logits_and_all_points_selected(dev):
    from differentiable_ransac_amd import ops
    N, B = 8, 4
    logits = torch.tensor([[1e4, -1e4, 50.0, 0.0, -3.0, 2.0, 1e4, 7.0]])
    r = ops.gumbel_topk(logits.to(dev), B, 8, 1.0, None, seed=3, dense=True)      # k == N: every point is selected
    assert torch.equal(r["idx"].cpu(), torch.arange(8, dtype=torch.int32).expand(1, B, 8))
    assert torch.isfinite(r["lse"]).all() and torch.isfinite(r["y_soft"]).all()
    assert (r["y_soft"].sum(-1) - 1).abs().max() < 1e-5
    r = ops.gumbel_topk(logits.to(dev), B, 2, 1.0, None, seed=3)
    assert set(r["idx"].cpu().flatten().tolist()) <= {0, 6}                        # the two 1e4 logits always win


def test_wrong_device_dtype_and_shapes_raise(dev):
    from differentiable_ransac_amd import ops
    from differentiable_ransac_amd._lib import DransacError
    m = torch.rand(1, 16, 4)
    with pytest.raises(DransacError):
        ops.msac_score(m, torch.rand(1, 3, 3, 3), 1e-3)                            # CPU tensors: no fallback
    with pytest.raises(DransacError):
        ops.msac_score(m.half().to(dev), torch.rand(1, 3, 3, 3).half().to(dev), 1e-3)   # f16 unsupported (Q15)
    with pytest.raises(DransacError):
        ops.gumbel_topk(torch.rand(1, 16).to(dev), 4, 9, 1.0)                      # k > 8
    with pytest.raises(DransacError):
        ops.solve_stewenius5(torch.rand(3, 6, 4).to(dev))
    # empty inputs are rejected with DR_EINVAL (no kernel is launched), never a crash
    with pytest.raises(DransacError, match="bad sizes"):
        ops.msac_score(torch.rand(1, 16, 4).to(dev), torch.rand(1, 0, 3, 3).to(dev), 1e-3)
    with pytest.raises(DransacError):
        ops.solve_nister5(torch.rand(0, 5, 4).to(dev))
    with pytest.raises(DransacError):
        ops.gumbel_topk(torch.rand(1, 4).to(dev), 4, 5, 1.0)                       # k > N


def test_f64_end_to_end(dev):
    """`-pr 2`: the f64 entry points through the batched driver (Q17)."""
    from differentiable_ransac_amd import synth
    from differentiable_ransac_amd.ransac import BatchedRANSAC
    P, N, B = 2, 300, 64
    data = synth.batch_two_view(P, N, seed0=60, dtype=torch.float64)
    noise = [synth.gumbel_noise((P, B, N), seed=5, dtype=torch.float64).to(dev)]
    rn = BatchedRANSAC("nister", ransac_batch_size=B, threshold=0.75, max_iterations=B, refit=True)
    out = rn(data["matches"].to(dev), data["logits"].to(dev), data["K1"].to(dev), data["K2"].to(dev), gumbels=noise)
    assert out["model"].dtype == torch.float64
    for p in range(P):
        m, mask, score, it = O.ransac_test(data["matches"][p], data["logits"][p], [noise[0][p].cpu()], data["K1"][p],
                                           data["K2"][p], "nister", max_iterations=B)
        assert abs(float(out["score"][p]) - score) <= 1e-6 * max(1.0, score)
        assert torch.equal(out["mask"][p].cpu(), mask)
        assert (O.canonical(out["model"][p].cpu()) - O.canonical(m)).abs().max() < 1e-6


@pytest.mark.parametrize("P,N,M,masked,kept", [(1, 7, 1, True, True), (2, 33, 5, False, False), (3, 1030, 19, True, True),
                                               (2, 2500, 300, True, False), (2, 64, 1024, False, True)])
def test_match_loss_per_pair_reduction(dev, P, N, M, masked, kept):
    """The fused per-pair reduction (dr_match_loss_pair + dr_episym_bwd_pair) against the composition it replaces
    (episym sums, then the means in torch): with / without mask and keep flags, a pair without any selected point, a pair
    without any kept model (denominator clamped to 1), forward and backward."""
    from differentiable_ransac_amd import ops
    g = torch.Generator().manual_seed(P * 77 + N + M)
    m = (torch.rand(P, N, 4, generator=g) - 0.5).to(dev)
    E = torch.randn(P, M, 3, 3, generator=g)
    mask = (torch.rand(P, N, generator=g) < 0.5).to(dev) if masked else None
    keep = (torch.rand(P, M, generator=g) < 0.6).to(dev) if kept else None
    if masked and P > 1:
        mask[0] = False
    if kept and P > 1:
        keep[-1] = False
    wts = torch.rand(P, generator=g).to(dev)
    Ea = E.clone().to(dev).requires_grad_(True)
    got = ops.match_loss_per_pair(m, mask, Ea, keep)
    (got * wts).sum().backward()
    Eb = E.clone().to(dev).requires_grad_(True)
    sums = ops.episym_sums(m, mask, Eb, keep)
    n_in = mask.sum(1).float() if masked else torch.full((P,), float(N), device=dev)
    n_models = keep.sum(1).float() if kept else torch.full((P,), float(M), device=dev)
    want = sums.sum(1) / (n_in * n_models).clamp(min=1.0)
    (want * wts).sum().backward()
    assert got.shape == (P,)
    assert torch.allclose(got, want, rtol=2e-6, atol=1e-9)
    assert torch.allclose(Ea.grad, Eb.grad, rtol=1e-5, atol=1e-9 + 1e-6 * float(Eb.grad.abs().max()))
    if kept:
        assert (Ea.grad[~keep] == 0).all()


@pytest.mark.parametrize("P,N,M,masked", [(1, 7, 1, True), (2, 33, 5, False), (3, 1030, 19, True), (1, 2500, 9, True)])
def test_match_loss_kernel_odd_sizes(dev, P, N, M, masked):
    """MatchLoss kernels on ragged sizes (N not a multiple of the lane tile, more than one 2048-point chunk, model count
    not a multiple of the block tile, no mask, empty mask) against the oracle, forward and backward."""
    from differentiable_ransac_amd import ops
    g = torch.Generator().manual_seed(P * 1000 + N)
    m = (torch.rand(P, N, 4, generator=g) - 0.5)
    E = torch.randn(P, M, 3, 3, generator=g)
    mask = (torch.rand(P, N, generator=g) < 0.5) if masked else None
    if masked:
        mask[0] = False                      # a pair without a single selected point
    valid = torch.rand(P, M, generator=g) < 0.8
    Ed = E.clone().to(dev).requires_grad_(True)
    sums = ops.episym_sums(m.to(dev), None if mask is None else mask.to(dev), Ed, valid.to(dev))
    wts = torch.rand(P, M, generator=g)
    (sums * wts.to(dev)).sum().backward()
    Eo = E.double().clone().requires_grad_(True)
    tot = 0
    for p in range(P):
        sel = mask[p] if mask is not None else torch.ones(N, dtype=torch.bool)
        ys = O.episym(m[p, sel, :2].double(), m[p, sel, 2:].double(), Eo[p])
        want = torch.clamp(ys, max=1.0).sum(1) * valid[p]
        assert torch.allclose(sums[p].detach().cpu().double(), want.detach(), rtol=2e-4, atol=1e-5)
        tot = tot + (want * wts[p].double()).sum()
    if tot.requires_grad:
        tot.backward()
        ref = Eo.grad
        got = Ed.grad.cpu().double()
        assert (got - ref).abs().max() <= 2e-3 * ref.abs().max() + 1e-6


def test_pose_error_and_topdown_tiny_sizes(dev):
    from differentiable_ransac_amd import ops, synth
    d = synth.two_view_pair(3, 9, dtype=torch.float64)
    E = torch.stack((d["gt_E"], d["gt_E"] + 0.01 * torch.eye(3, dtype=torch.float64)))[None].to(dev)
    eq, et, which, votes = ops.pose_error(d["matches"][None].to(dev), E, d["R"][None].to(dev), d["t"][None].to(dev), want_votes=True)
    oq, ot, ow = O.pose_error(E[0].cpu(), d["matches"], d["R"], d["t"])
    assert torch.equal(which[0].cpu().long(), ow) and (eq[0].cpu() - oq).abs().max() < 1e-5 and int(votes.sum(-1).max()) <= 9
    # top-down draw: k == N takes every point; one point with all the mass is always drawn
    idx = ops.topdown_sample(torch.zeros(2, 5, device=dev), 64, 5, seed=1)
    assert (idx == torch.arange(5, device=dev, dtype=torch.int32)).all()
    lg = torch.full((1, 40), -30.0, device=dev)
    lg[0, 17] = 30.0
    idx = ops.topdown_sample(lg, 256, 3, seed=2)
    assert (idx == 17).any(-1).all() and (idx[..., 1:] > idx[..., :-1]).all()
