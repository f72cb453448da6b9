"""Round-4 additions on the GPU: the screened long-row sampler (index sets bit-identical to the unscreened kernel and to the
oracle on the reported noise), the one-pair call replayed as a HIP graph, the sampler backward on a tail group of very negative
logits, the f64 train-mode refusal."""
import pytest
import torch

from oracle import cpu_ref as O

pytestmark = pytest.mark.gpu


# ------------------------------------------------------------------------------------- K1, long rows, screened (dr_gumbel_topk_index_f32)
@pytest.mark.parametrize("P,B,N,k", [(1, 2048, 50000, 3),     # BASELINE configs[3]: four waves per row
                                     (3, 128, 4096, 5),       # four waves per row, k = 5 lists
                                     (2, 4096, 4096, 3),      # more than 4096 rows: a wave per row
                                     (1, 64, 20000, 1)])
def test_screened_long_row_sampler_draws_the_unscreened_index_sets(dev, P, B, N, k):
    from differentiable_ransac_amd import ops
    gen = torch.Generator().manual_seed(N + k)
    logits = (torch.randn(P, N, generator=gen) + 3.0 * (torch.rand(P, N, generator=gen) < 0.3)).to(dev)
    for seed in (1, 77, 123456789):
        a = ops.gumbel_topk(logits, B, k, 1.0, None, seed, soft=False, screen=False)["idx"]
        b = ops.gumbel_topk(logits, B, k, 1.0, None, seed, soft=False, screen=True)["idx"]
        assert torch.equal(a, b)
        assert bool((b[..., 1:] > b[..., :-1]).all()) and int(b.min()) >= 0 and int(b.max()) < N
    # the statistics kernel (soft=True) draws the same sets as well
    c = ops.gumbel_topk(logits, B, k, 1.0, None, 1, soft=True)["idx"]
    assert torch.equal(c, ops.gumbel_topk(logits, B, k, 1.0, None, 1, soft=False, screen=True)["idx"])


def test_screened_sampler_against_the_oracle_on_the_reported_noise(dev):
    """the general kernel reports its in-kernel noise; the oracle's top-k on that noise is what the screened kernel must draw"""
    from differentiable_ransac_amd import ops
    P, B, N, k = 1, 96, 8192, 3
    gen = torch.Generator().manual_seed(5)
    logits = torch.randn(P, N, generator=gen).to(dev)
    full = ops.gumbel_topk(logits, B, k, 1.0, None, 9, want_noise=True)
    scr = ops.gumbel_topk(logits, B, k, 1.0, None, 9, soft=False, screen=True)["idx"]
    idx, _, _ = O.gumbel_topk(logits[0].cpu(), full["gumbel"][0].cpu(), 1.0, k)
    assert torch.equal(scr[0].cpu().long(), idx.long())


def test_screened_sampler_falls_back_to_the_full_pass(dev):
    """rows in which fewer than k points reach the screening score repeat themselves unscreened: forced here through logits whose
    log-sum-exp is not finite (T = +inf: no point ever passes) and through a single dominant point (every row's count is 1 < k)"""
    from differentiable_ransac_amd import ops
    P, B, N, k = 2, 256, 4096, 3
    gen = torch.Generator().manual_seed(11)
    logits = torch.randn(P, N, generator=gen)
    logits[0, 17] = float("inf")
    logits[1, 100] = 60.0          # softmax mass ~1 on one point: T sits just below it, nothing else reaches T
    logits = logits.to(dev)
    for seed in (3, 4):
        a = ops.gumbel_topk(logits, B, k, 1.0, None, seed, soft=False, screen=False)["idx"]
        b = ops.gumbel_topk(logits, B, k, 1.0, None, seed, soft=False, screen=True)["idx"]
        assert torch.equal(a, b)
    assert bool((b[0] == 17).any(-1).all()) and bool((b[1] == 100).any(-1).all())


def test_screened_sampler_with_a_device_seed_equals_the_host_seed(dev):
    from differentiable_ransac_amd import ops
    P, B, N, k = 1, 512, 16384, 3
    logits = torch.randn(P, N, generator=torch.Generator().manual_seed(2)).to(dev)
    ds = ops.DeviceSeed(5, dev, 0)
    host = [(5 * 0x9E3779B97F4A7C15 + c) & (2 ** 64 - 1) for c in range(2)]
    for c in range(2):
        a = ops.gumbel_topk(logits, B, k, 1.0, None, ds.next(), soft=False)["idx"]
        b = ops.gumbel_topk(logits, B, k, 1.0, None, host[c], soft=False)["idx"]
        assert torch.equal(a, b)


# ------------------------------------------------------------------------------------- one pair per call (test.py:38)
def test_one_pair_graph_replay_equals_the_batched_eager_driver(dev):
    """model_cl.py:488-490 calls the path one pair at a time: the P = 1 call replayed as a HIP graph returns, call after call,
    what the batched eager driver with the same seed returns for that pair (pair 0 of a three-pair batch: the Philox counters
    carry the pair's index within the call, everything else is per pair)"""
    from differentiable_ransac_amd import synth
    from differentiable_ransac_amd.graphs import GraphedStep
    from differentiable_ransac_amd.ransac import BatchedRANSAC
    N, B = 2000, 1024
    d = synth.batch_two_view(3, N, seed0=31)
    m, lg, K1, K2 = (d[k_].to(dev) for k_ in ("matches", "logits", "K1", "K2"))
    kw = dict(ransac_batch_size=B, threshold=0.75, max_iterations=B, seed=7, refit=False, keep_masks=False)
    eager = BatchedRANSAC("nister", **kw)
    one = BatchedRANSAC("nister", **kw).device_seeds(dev)
    m1, lg1, K11, K21 = m[:1].clone(), lg[:1].clone(), K1[:1].clone(), K2[:1].clone()
    warm = 3
    for _ in range(warm):
        eager(m, lg, K1, K2)
    step = GraphedStep(lambda: one(m1, lg1, K11, K21), warmup=warm)
    for r in range(4):
        want = eager(m, lg, K1, K2)
        got = step()
        for key in ("model", "mask", "score", "inliers"):
            assert torch.equal(want[key][:1], got[key]), (key, r)


# ------------------------------------------------------------------------------------- sampler backward, tail group (round-3 advice)
@pytest.mark.parametrize("N", [1001, 1002, 1003])
def test_sampler_backward_tail_group_with_very_negative_logits(dev, N):
    """N % 4 != 0 and the last group's real logits around -100: the exponential-race form took its reference logit over the
    padding value 0 as well and overflowed (NaN gradients); against the f64 autograd of the oracle on the reported noise"""
    from differentiable_ransac_amd import ops
    P, B, k = 1, 64, 5
    gen = torch.Generator().manual_seed(N)
    logits = torch.randn(P, N, generator=gen)
    logits[0, 4 * (N // 4):] = -100.0 + torch.randn(N - 4 * (N // 4), generator=gen)
    logits = logits.to(dev)
    seed = 21
    fwd = ops.gumbel_topk(logits, B, k, 1.0, None, seed, want_noise=True)
    a_sel = torch.randn(P, B, k, generator=gen).to(dev)
    grad = ops.gumbel_topk_bwd(logits, None, seed, 1.0, fwd["idx"], fwd["lse"], a_sel)
    assert bool(torch.isfinite(grad).all())
    lg = logits[0].double().cpu().requires_grad_(True)
    y = torch.softmax(lg[None, :] + fwd["gumbel"][0].double().cpu(), -1)
    (torch.gather(y, 1, fwd["idx"][0].long().cpu()) * a_sel[0].double().cpu()).sum().backward()
    ref = lg.grad
    assert float((grad[0].double().cpu() - ref).abs().max()) <= 1e-5 * max(1.0, float(ref.abs().max()))


# ------------------------------------------------------------------------------------- f64 train mode (`-pr 2 -tr 1`, model_cl.py:164-169, Q17)
def test_f64_train_mode_f8_gradient_against_the_f64_oracle(dev):
    """the drop-in RANSAC in double precision, train mode, on the reference's own training fixture: chosen models and the gradient
    to the logits against torch autograd through the f64 oracle on the same noise (round 5: sampler, gather AND the 8-point
    backward with f64 in memory -- dr_solve_f8_bwd_f64 -- so the tolerance is a rounding-level one, not one f32 rounding)"""
    import numpy as np
    from differentiable_ransac_amd import estimators, samplers, scorings
    from differentiable_ransac_amd.ransac import RANSAC
    z = np.load(__import__("os").path.join(__import__("os").path.dirname(__file__), "golden", "ransac_train_f8.npz"))
    g = {k_: torch.from_numpy(z[k_]) for k_ in z.files if z[k_].ndim > 0}
    B = 32
    est = estimators.FundamentalMatrixEstimatorNew("cuda")
    smp = samplers.GumbelSoftmaxSampler(B, 8, device="cuda", data_type=torch.float64)
    r = RANSAC(est, smp, scorings.MSACScore("cuda"), train=True, ransac_batch_size=B, fmat=True, sampler_id=3,
               threshold=0.75, max_iterations=100)
    logits = g["logits"].double().to(dev).requires_grad_(True)
    models, _, _, _ = r(g["matches"].double().to(dev), logits, g["K1"].double().to(dev), g["K2"].double().to(dev),
                        g["gt"].double().to(dev), gumbels=[x.double().to(dev) for x in g["gumbels"]])
    chosen = torch.cat([models[k_] for k_ in sorted(models.keys())])
    assert chosen.dtype == torch.float64
    l64 = g["logits"].double().requires_grad_(True)
    o64 = torch.cat([O.ransac_train_batch(g["matches"].double(), l64, x.double(), g["gt"].double(), "f8")[0] for x in g["gumbels"]])
    s = torch.sign((chosen.detach().cpu() * o64.detach()).sum((-1, -2)))
    rel = (chosen.detach().cpu() * s[:, None, None] - o64.detach()).abs().amax((-1, -2)) / o64.detach().abs().amax((-1, -2))
    assert rel.max() < 1e-6, float(rel.max())                      # f64 forward: far below the f32 path's 1e-3
    w = g["grad_weight"].double()
    (o64 * w).sum().backward()
    (chosen * (w * s[:, None, None]).to(dev)).sum().backward()
    gl = logits.grad.cpu()
    assert gl.dtype == torch.float64 and bool(torch.isfinite(gl).all())
    assert (gl - l64.grad).abs().max() <= 1e-8 * l64.grad.abs().max(), (float((gl - l64.grad).abs().max()), float(l64.grad.abs().max()))


def test_f64_sampler_and_gather_backward_against_autograd(dev):
    """dr_gumbel_topk_bwd_f64 + dr_gather_bwd_f64 against torch autograd of the dense f64 formula on explicit noise"""
    from differentiable_ransac_amd import ops
    P, B, N, k = 2, 48, 300, 5
    gen = torch.Generator().manual_seed(8)
    matches = torch.randn(P, N, 4, generator=gen, dtype=torch.float64)
    logits = torch.randn(P, N, generator=gen, dtype=torch.float64)
    u = torch.rand(P, B, N, generator=gen, dtype=torch.float64).clamp(1e-12, 1 - 1e-12)
    noise = -torch.log(-torch.log(u))
    gs = torch.randn(P, B, k, 4, generator=gen, dtype=torch.float64)
    gw = torch.randn(P, B, k, generator=gen, dtype=torch.float64)
    lg = logits.to(dev).requires_grad_(True)
    mt = matches.to(dev).requires_grad_(True)
    samples, w, idx = ops.SampleGather.apply(mt, lg, B, k, 1.0, noise.to(dev), 0)
    ((samples * gs.to(dev)).sum() + (w * gw.to(dev)).sum()).backward()
    lr = logits.clone().requires_grad_(True)
    mr = matches.clone().requires_grad_(True)
    y = torch.softmax(lr[:, None, :] + noise, -1)
    ii = idx.long().cpu()
    ysel = torch.gather(y, 2, ii)
    st = (1.0 - ysel.detach()) + ysel                                # straight-through value, gumbel_sampler.py:38
    pts = torch.gather(mr[:, None].expand(P, B, N, 4), 2, ii[..., None].expand(P, B, k, 4)) * st[..., None]
    ((pts * gs).sum() + (ysel * gw).sum()).backward()
    assert (lg.grad.cpu() - lr.grad).abs().max() <= 1e-10 * max(1.0, float(lr.grad.abs().max()))
    assert (mt.grad.cpu() - mr.grad).abs().max() <= 1e-10 * max(1.0, float(mr.grad.abs().max()))


def test_f64_five_point_backward_equals_the_f32_path_to_rounding(dev):
    from differentiable_ransac_amd import ops, synth
    d = synth.batch_two_view(1, 64, seed0=3, inlier_ratio=1.0)
    smp = d["matches"][0, :60].reshape(12, 5, 4)
    gen = torch.Generator().manual_seed(1)
    gm = torch.randn(12, 10, 3, 3, generator=gen)
    out = {}
    for dt in (torch.float32, torch.float64):
        s_ = smp.to(dt).to(dev).requires_grad_(True)
        models, valid = ops.solve_essential(s_, None, "nister")
        (models * gm.to(dt).to(dev) * valid[..., None, None]).sum().backward()
        out[dt] = (s_.grad.double().cpu(), valid.cpu())
    assert out[torch.float64][0].dtype == torch.float64
    a, b = out[torch.float32][0], out[torch.float64][0]
    assert torch.isfinite(b).all()
    # the same tangent-space solve on f32-rounded inputs either way: agreement to a few 1e-4 of the gradient's scale
    assert (a - b).abs().max() <= 5e-3 * max(1.0, float(b.abs().max()))


# ------------------------------------------------------------------------------------- K3, few samples per block on small grids
@pytest.mark.parametrize("nsmp", [1024, 100, 4096, 3])
def test_five_point_solver_with_few_samples_per_block_equals_the_full_blocks(dev, nsmp):
    """calls with few samples run 16 / 8 / 4 samples per 64-lane block (latency of a one-pair call); a sample's solutions do not
    depend on the block it shares: the same samples inside a large batch (32 per block) come out bit-identical, f32 and f64 models"""
    from differentiable_ransac_amd import ops, synth
    d = synth.batch_two_view(1, 2000, seed0=9)
    r = ops.gumbel_topk(d["logits"].to(dev), nsmp, 5, 1.0, None, seed=3, soft=False)
    smp = ops.gather(d["matches"].to(dev), r["idx"])[0]                 # [nsmp, 5, 4]
    m_s, v_s = ops.solve_nister5(smp)
    big = torch.cat([smp, smp.flip(0).repeat(1 + 40000 // nsmp, 1, 1)])
    m_b, v_b = ops.solve_nister5(big, path=1)      # round 5: a batch of this size would take the two-phase kernel by itself
    assert torch.equal(v_s, v_b[:nsmp]) and torch.equal(m_s, m_b[:nsmp])
    assert int(v_s.sum()) > 2 * nsmp or nsmp < 10
    mh, m64, vh = ops.solve_nister5_hp(smp)
    mhb, m64b, vhb = ops.solve_nister5_hp(big, path=1)
    assert torch.equal(mh, mhb[:nsmp]) and torch.equal(m64, m64b[:nsmp]) and torch.equal(vh, vhb[:nsmp])
    ms, vs = ops.solve_stewenius5(smp)
    msb, vsb = ops.solve_stewenius5(big, path=1)
    assert torch.equal(ms, msb[:nsmp]) and torch.equal(vs, vsb[:nsmp])
    msf, vsf = ops.solve_stewenius5(big, path=2)   # Stewenius' two-phase kernel runs the same arithmetic per sample
    assert torch.equal(ms, msf[:nsmp]) and torch.equal(vs, vsf[:nsmp])


# ------------------------------------------------------------------------------------- 3-D driver: fused gather + solve, sums without a memset
@pytest.mark.parametrize("N", [4096, 1024, 1000])     # rows over two chunks / one chunk / the general residual kernel (N % 16 != 0)
def test_rigid_gather_solve_and_accumulated_sums_equal_the_separate_launches(dev, N):
    from differentiable_ransac_amd import ops, synth
    P, B = 2, 300
    items = [synth.rigid_pair(p, N) for p in range(P)]
    m = torch.stack([it["matches"] for it in items]).float().to(dev)
    lg = torch.stack([it["logits"] for it in items]).to(dev)
    idx = ops.gumbel_topk(lg, B, 3, 1.0, None, 5, soft=False)["idx"]
    for flag in (True, False):
        smp = ops.gather(m, idx)
        model, R, t, sc, valid = ops.solve_rigid(smp.reshape(P * B, 3, 6), None, flag)
        res0, mk0 = ops.rigid_residual(m, model.reshape(P, B, 4, 4), 0.03, True)
        sums = torch.full((P, B), 7.0, device=dev)                      # garbage on entry: cleared by the solve
        model2, valid2 = ops.solve_rigid_gather(m, idx, flag, zero_sums=sums)
        assert torch.equal(model2.reshape(P * B, 4, 4), model) and torch.equal(valid2.reshape(-1), valid)
        assert float(sums.abs().max()) == 0.0
        res1, mk1 = ops.rigid_residual(m, model2, 0.03, True, res=sums)
        assert torch.equal(mk0, mk1) and torch.allclose(res0, res1, rtol=1e-5, atol=1e-6)   # float atomics: order of the chunks
        res2, none = ops.rigid_residual(m, model2, 0.03, False, res=torch.zeros(P, B, device=dev))
        assert none is None and torch.allclose(res2, res0, rtol=1e-5, atol=1e-6)


def test_fused_uniform_sampler_gather_eight_point_solve_equals_the_three_launches(dev):
    from differentiable_ransac_amd import ops, synth
    P, N, B = 5, 128, 64
    d = synth.batch_two_view(P, N, seed0=12, pixel=True)
    m = d["matches"].to(dev)
    for seed in (0, 99, 2 ** 40 + 7):
        idx0 = ops.uniform_sample(P, B, 8, N, seed, dev)
        F0, v0 = ops.solve_f8(ops.gather(m, idx0))
        idx1, F1, v1 = ops.solve_f8_uniform(m, B, seed)
        assert torch.equal(idx0, idx1) and torch.equal(F0, F1) and torch.equal(v0, v1)
        assert int(idx1.max()) <= N - 2 and int(idx1.min()) >= 0
    ds = ops.DeviceSeed(3, dev, 0)
    a = ops.solve_f8_uniform(m, B, ds.next())
    b = ops.solve_f8_uniform(m, B, (3 * 0x9E3779B97F4A7C15) & (2 ** 64 - 1))
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
