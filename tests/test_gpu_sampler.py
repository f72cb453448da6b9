"""K1 / K1u / K2 parity on the GPU."""
import pytest
import torch

from oracle import cpu_ref as O
from tests import philox_ref
from tests.conftest import load_golden

pytestmark = pytest.mark.gpu


def test_gumbel_golden_index_sets_bit_exact(dev):
    from differentiable_ransac_amd.samplers import GumbelSoftmaxSampler
    for name, dt in (("gumbel_f32", torch.float32), ("gumbel_f64", torch.float64), ("gumbel_k8_tau05", torch.float32)):
        g = load_golden(name)
        B, N = g["gumbels"].shape
        s = GumbelSoftmaxSampler(B, g["k"], tau=g["tau"], device="cuda", data_type=dt)
        ret, y_soft = s.sample(g["logits"].to(dev), gumbels=g["gumbels"].to(dev))
        assert ret.shape == (B, N) and y_soft.shape == (B, N)
        assert torch.equal(ret.cpu() != 0, g["ret"] != 0)              # bit-exact index sets
        tol = 2e-6 if dt == torch.float32 else 1e-14
        assert (y_soft.cpu() - g["y_soft"]).abs().max() <= tol
        assert (ret.cpu() - g["ret"]).abs().max() <= tol
        assert (ret.cpu()[g["ret"] == 0] == 0).all()                    # non-selected entries exactly zero
        assert torch.equal(s.last_indices.cpu().long(), O.gumbel_topk(g["logits"], g["gumbels"], g["tau"], g["k"])[0])


@pytest.mark.parametrize("N,B,k", [(2000, 1024, 5), (2000, 256, 8), (131, 7, 3), (5, 3, 5), (50000, 16, 3), (17000, 9, 8)])
def test_gumbel_vs_oracle_explicit_noise(dev, N, B, k):
    from differentiable_ransac_amd import ops, synth
    P = 2
    logits = torch.stack([synth.two_view_pair(10 + p, N)["logits"] for p in range(P)])
    noise = synth.gumbel_noise((P, B, N), seed=N + k)
    r = ops.gumbel_topk(logits.to(dev), B, k, 1.0, noise.to(dev), dense=(N <= 2000))
    for p in range(P):
        idx, ret, y_soft = O.gumbel_topk(logits[p], noise[p], 1.0, k)
        assert torch.equal(r["idx"][p].cpu().long(), idx)
        ys = torch.gather(y_soft, 1, idx)
        assert (r["y_sel"][p].cpu() - ys).abs().max() <= 1e-6 + 1e-5 * ys.max()
        g = (logits[p][None] + noise[p]) / 1.0
        assert (r["lse"][p].cpu() - torch.logsumexp(g.double(), -1)).abs().max() < 1e-4
        if N <= 2000:
            assert torch.equal(r["ret"][p].cpu() != 0, ret != 0)
            assert (r["y_soft"][p].cpu() - y_soft).abs().max() < 2e-6


@pytest.mark.parametrize("N,B,k,dtype", [(2000, 1024, 5, torch.float32), (131, 7, 3, torch.float32), (5, 3, 5, torch.float32),
                                          (50000, 16, 3, torch.float32), (2000, 64, 8, torch.float64)])
def test_gumbel_index_only_mode(dev, N, B, k, dtype):
    """soft=False (y_sel = lse = NULL: what test mode consumes) returns exactly the index sets of the full kernel, with
    explicit noise and with the in-kernel generator, with ties (slow path) too; half-specified outputs are refused."""
    from differentiable_ransac_amd import ops, synth, _lib as L
    P = 2
    logits = torch.stack([synth.two_view_pair(20 + p, N)["logits"] for p in range(P)]).to(dtype).to(dev)
    noise = synth.gumbel_noise((P, B, N), seed=N + k).to(dtype).to(dev)
    for g, seed in ((noise, 0), (None, 77)):
        full = ops.gumbel_topk(logits, B, k, 0.7, g, seed=seed)
        only = ops.gumbel_topk(logits, B, k, 0.7, g, seed=seed, soft=False)
        assert only["y_sel"] is None and only["lse"] is None
        assert torch.equal(full["idx"], only["idx"])
    flat = torch.zeros(P, N, dtype=dtype, device=dev)
    assert torch.equal(ops.gumbel_topk(flat, B, k, 1.0, torch.zeros_like(noise), soft=False)["idx"].cpu(),
                       torch.arange(k, dtype=torch.int32).expand(P, B, k))
    with pytest.raises(ValueError):
        ops.gumbel_topk(logits, B, k, soft=False, dense=True)
    idx = torch.empty(P, B, k, dtype=torch.int32, device=dev)
    ysel = torch.empty(P, B, k, dtype=dtype, device=dev)
    with pytest.raises(L.DransacError):     # y_sel without lse
        L.call(f"dr_gumbel_topk_fwd_{L.suffix(dtype)}", L.ptr(logits), None, L.c_uint64(1), None, L.scalar(dtype, 1.0), L.c_int(P),
               L.c_int(B), L.c_int(N), L.c_int(k), L.ptr(idx), L.ptr(ysel), None, None, None, None, L.stream())


def test_gumbel_ties_take_the_slow_path(dev):
    from differentiable_ransac_amd import ops
    N, B, k = 300, 5, 4
    logits = torch.zeros(1, N)
    noise = torch.zeros(1, B, N)
    r = ops.gumbel_topk(logits.to(dev), B, k, 1.0, noise.to(dev))
    assert torch.equal(r["idx"].cpu(), torch.arange(k, dtype=torch.int32).expand(1, B, k))
    assert torch.allclose(r["y_sel"].cpu(), torch.full((1, B, k), 1.0 / N), rtol=1e-5)
    # a block of exactly-equal maxima larger than the candidate buffer
    logits[0, 100:200] = 5.0
    r = ops.gumbel_topk(logits.to(dev), B, k, 1.0, noise.to(dev))
    assert torch.equal(r["idx"].cpu(), (100 + torch.arange(k, dtype=torch.int32)).expand(1, B, k))


def test_gumbel_philox_mode(dev):
    from differentiable_ransac_amd import ops, synth
    P, N, B, k = 3, 2000, 64, 5
    logits = torch.stack([synth.two_view_pair(20 + p, N)["logits"] for p in range(P)]).to(dev)
    r1 = ops.gumbel_topk(logits, B, k, 1.0, None, seed=1234, want_noise=True)
    r2 = ops.gumbel_topk(logits, B, k, 1.0, None, seed=1234)
    r3 = ops.gumbel_topk(logits, B, k, 1.0, None, seed=1235)
    assert torch.equal(r1["idx"], r2["idx"]) and not torch.equal(r1["idx"], r3["idx"])
    noise = r1["gumbel"].cpu()
    # integer side of the RNG is reproduced exactly by the numpy restatement; the two logs differ by float rounding
    ref = torch.from_numpy(philox_ref.gumbel_noise_f32(1234, P, B, N))
    assert (noise - ref).abs().max() < 2e-5 * (1 + ref.abs().max())
    # given the noise the kernel reports, the oracle selects the same index sets, bit-exactly
    for p in range(P):
        idx, _, _ = O.gumbel_topk(logits[p].cpu(), noise[p], 1.0, k)
        assert torch.equal(r1["idx"][p].cpu().long(), idx)
    # statistical sanity: Gumbel(0,1) mean = Euler gamma, var = pi^2/6
    assert abs(float(noise.mean()) - 0.5772) < 0.01 and abs(float(noise.var()) - 1.6449) < 0.03
    # inliers (logit +3) dominate the selection
    inl = torch.stack([synth.two_view_pair(20 + p, N)["inliers"] for p in range(P)])
    frac = torch.gather(inl[:, None, :].expand(P, B, N), 2, r1["idx"].cpu().long()).float().mean()
    assert frac > 0.85


def test_gumbel_none_logits_and_uniform(dev):
    from differentiable_ransac_amd import ops
    from differentiable_ransac_amd.samplers import UniformSampler
    r = ops.gumbel_topk(None, 32, 5, 1.0, None, seed=7, N=500, device=dev)
    idx = r["idx"].cpu()
    assert idx.shape == (1, 32, 5) and (idx[..., 1:] > idx[..., :-1]).all() and idx.min() >= 0 and idx.max() < 500
    u = ops.uniform_sample(3, 64, 8, 128, seed=99, device=dev).cpu().long()
    assert torch.equal(u, torch.from_numpy(philox_ref.uniform_indices(99, 3, 64, 8, 128)))
    assert u.min() >= 0 and u.max() <= 126          # randint(0, N-1): last point never drawn
    s = UniformSampler(64, 8, device="cuda", seed=3)
    a = s.sample(128)
    assert a.shape == (64, 8) and a.dtype == torch.int64 and a.max() <= 126


def test_gather_and_straight_through(dev):
    from differentiable_ransac_amd import ops, synth
    N, B, k = 777, 33, 5
    pair = synth.two_view_pair(5, N)
    noise = synth.gumbel_noise((1, B, N), seed=3)
    r = ops.gumbel_topk(pair["logits"][None].to(dev), B, k, 1.0, noise.to(dev))
    smp = ops.gather(pair["matches"][None].to(dev), r["idx"], r["y_sel"])
    idx, ret, soft = O.gumbel_topk(pair["logits"], noise[0], 1.0, k)
    ref, w = O.gather_samples(pair["matches"], ret, soft)
    assert (smp[0].cpu() - ref).abs().max() < 1e-6
    raw = ops.gather(pair["matches"][None].to(dev), r["idx"], None)
    assert torch.equal(raw[0].cpu(), pair["matches"][idx])
    rp = synth.rigid_pair(1, 300)
    r = ops.gumbel_topk(rp["logits"][None].to(dev), 8, 3, 1.0, None, seed=5)
    raw = ops.gather(rp["matches"][None].to(dev), r["idx"], None)
    assert torch.equal(raw[0].cpu(), rp["matches"][r["idx"][0].cpu().long()])


@pytest.mark.parametrize("explicit", [True, False])
def test_sampler_gather_backward(dev, explicit):
    from differentiable_ransac_amd import ops, synth
    N, B, k, tau = 500, 300, 5, 0.7
    P = 2
    pairs = [synth.two_view_pair(30 + p, N) for p in range(P)]
    logits = torch.stack([q["logits"] for q in pairs])
    matches = torch.stack([q["matches"] for q in pairs])
    gen = torch.Generator().manual_seed(9)
    W = torch.randn(P, B, k, 4, generator=gen)
    V = torch.randn(P, B, k, generator=gen)
    lg = logits.to(dev).requires_grad_(True)
    mt = matches.to(dev).requires_grad_(True)
    if explicit:
        noise = synth.gumbel_noise((P, B, N), seed=77)
        smp, w, idx = ops.SampleGather.apply(mt, lg, B, k, tau, noise.to(dev), 0)
    else:
        noise = ops.gumbel_topk(logits.to(dev), B, k, tau, None, seed=4242, want_noise=True)["gumbel"].cpu()
        smp, w, idx = ops.SampleGather.apply(mt, lg, B, k, tau, None, 4242)
    ((smp * W.to(dev)).sum() + (w * V.to(dev)).sum()).backward()
    # oracle: torch autograd through the reference's op sequence, f64
    l64 = logits.double().requires_grad_(True)
    m64 = matches.double().requires_grad_(True)
    loss = 0
    for p in range(P):
        _, ret, soft = O.gumbel_topk(l64[p], noise[p].double(), tau, k)
        s_, w_ = O.gather_samples(m64[p], ret, soft)
        loss = loss + (s_ * W[p].double()).sum() + (w_ * V[p].double()).sum()
    loss.backward()
    gl, gm = lg.grad.cpu().double(), mt.grad.cpu().double()
    assert (gl - l64.grad).abs().max() <= 2e-4 * l64.grad.abs().max()
    assert (gm - m64.grad).abs().max() <= 1e-5 * m64.grad.abs().max()


def _pl_set_probs(w, k):
    """exact probability of every k-subset under sequential sampling without replacement (Plackett-Luce)."""
    import itertools
    n = len(w)
    probs = {}
    for perm in itertools.permutations(range(n), k):
        p, rest = 1.0, sum(w)
        for i in perm:
            p *= w[i] / rest
            rest -= w[i]
        key = tuple(sorted(perm))
        probs[key] = probs.get(key, 0.0) + p
    return probs


def test_topdown_sampler_has_the_gumbel_topk_distribution(dev):
    """The top-down sampler draws the index SET of the Gumbel top-k sampler: exact Plackett-Luce set probabilities on a
    small problem (chi-square), and the dense Gumbel kernel run on the same problem as a cross-check."""
    from differentiable_ransac_amd import ops
    logits = torch.tensor([[0.0, 1.0, -1.0, 2.0, 0.5, float("-inf"), 1.5]], device=dev)
    N, k, B = logits.shape[1], 3, 400000
    w = torch.exp(logits[0].double().cpu()).tolist()
    probs = _pl_set_probs(w, k)
    for name, idx in (("topdown", ops.topdown_sample(logits, B, k, seed=5)),
                      ("gumbel", ops.gumbel_topk(logits, B, k, 1.0, None, seed=5)["idx"])):
        idx = idx[0].long().cpu()
        assert idx.shape == (B, k)
        assert (idx[:, 1:] > idx[:, :-1]).all()                       # ascending, distinct
        assert (idx != 5).all()                                       # a zero-probability point is never drawn
        code = (idx * torch.tensor([N * N, N, 1])).sum(1)
        chi2, cells = 0.0, 0
        for key, pr in probs.items():
            if pr <= 0:
                continue
            obs = int((code == key[0] * N * N + key[1] * N + key[2]).sum())
            chi2 += (obs - B * pr) ** 2 / (B * pr)
            cells += 1
        # chi-square with (cells - 1) = 19 degrees of freedom: P(chi2 > 50) < 1.3e-4
        assert cells == 20 and chi2 < 50.0, (name, chi2)


def test_topdown_sampler_full_size_inclusion_frequencies(dev):
    """C2 size: per-point inclusion frequencies of the top-down sampler against the dense Gumbel kernel (both ~ the
    same Plackett-Luce law), and basic structure."""
    from differentiable_ransac_amd import ops, synth
    d = synth.batch_two_view(2, 2000, seed0=11)
    lg = d["logits"].to(dev)
    B, k = 65536, 5
    a = ops.topdown_sample(lg, B, k, seed=1)
    b = ops.gumbel_topk(lg, B, k, 1.0, None, seed=2)["idx"]
    for idx in (a, b):
        assert idx.shape == (2, B, k) and int(idx.min()) >= 0 and int(idx.max()) < 2000
        assert (idx[..., 1:] > idx[..., :-1]).all()
    for p in range(2):
        fa = torch.bincount(a[p].flatten().long(), minlength=2000).double() / B
        fb = torch.bincount(b[p].flatten().long(), minlength=2000).double() / B
        # inclusion probability of a point is a few 1e-3; the difference of two empirical frequencies has
        # sigma ~ sqrt(2 f / B) ~ 3e-4: 6 sigma bound on the maximum over 2000 points
        assert (fa - fb).abs().max() < 6 * (2 * fb.max() / B) ** 0.5 + 1e-4
        # first-order check against the soft-max itself: sum_n f_n = k
        assert abs(float(fa.sum()) - k) < 1e-9
    # different hypotheses and different seeds draw different sets; the same seed reproduces
    assert not torch.equal(a[0, 0], a[0, 1]) or not torch.equal(a[0, 1], a[0, 2])
    assert torch.equal(a, ops.topdown_sample(lg, B, k, seed=1))
    assert not torch.equal(a, ops.topdown_sample(lg, B, k, seed=3))
