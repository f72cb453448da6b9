"""H row: the RANSAC drivers (reference API and batched) against the reference's golden runs and the oracle."""
import pytest
import torch

from oracle import cpu_ref as O
from tests.conftest import load_golden

pytestmark = pytest.mark.gpu


def _make(name, B, train, max_it, weighted=0):
    from differentiable_ransac_amd.estimators import EssentialMatrixEstimatorNister, FundamentalMatrixEstimatorNew
    from differentiable_ransac_amd.ransac import RANSAC
    from differentiable_ransac_amd.samplers import GumbelSoftmaxSampler
    from differentiable_ransac_amd.scorings import MSACScore
    fmat = name == "f8"
    est = FundamentalMatrixEstimatorNew("cuda") if fmat else EssentialMatrixEstimatorNister("cuda")
    smp = GumbelSoftmaxSampler(B, 8 if fmat else 5, device="cuda")
    return RANSAC(est, smp, MSACScore("cuda"), fmat=fmat, train=train, ransac_batch_size=B,
                  sampler_id=3 if fmat else 2, weighted=weighted, threshold=0.75, max_iterations=max_it)


@pytest.mark.parametrize("name", ["nister", "f8"])
def test_train_mode_matches_reference_run(dev, name):
    g = load_golden(f"ransac_train_{name}")
    r = _make(name, 32, True, 100)
    logits = g["logits"].to(dev).requires_grad_(True)
    models, _, _, iters = r(g["matches"].to(dev), logits, g["K1"].to(dev), g["K2"].to(dev), g["gt"].to(dev),
                            gumbels=[x.to(dev) for x in g["gumbels"]])
    assert iters == g["iterations"] and sorted(models.keys()) == [0, 32, 64, 96]
    chosen = torch.cat([models[k] for k in sorted(models.keys())])
    ref = g["chosen"]
    if name == "f8":
        assert chosen.shape == ref.shape
        s = torch.sign((chosen.detach().cpu() * ref).sum((-1, -2)))
        rel = (chosen.detach().cpu() * s[:, None, None] - ref).abs().amax((-1, -2)) / ref.abs().amax((-1, -2))
        # the reference ran in f32 (its own f32-vs-f64 spread on this fixture is ~1e-3): compare with the f64 oracle too
        out64 = torch.cat([O.ransac_train_batch(g["matches"].double(), g["logits"].double(), x.double(),
                                                g["gt"].double(), "f8")[0] for x in g["gumbels"]])
        s64 = torch.sign((chosen.detach().cpu().double() * out64).sum((-1, -2)))
        rel64 = (chosen.detach().cpu().double() * s64[:, None, None] - out64).abs().amax((-1, -2)) / out64.abs().amax((-1, -2))
        assert rel64.max() < 1e-3 and rel64.median() < 1e-5, (rel64.max(), rel64.median())
        assert rel.median() < 1e-3
        # gradient w.r.t. the logits.  Exact arbiter: torch autograd through the f64 oracle on the same noise.  F's sign is
        # a LAPACK artefact, so all three runs (ours, oracle f64, reference f32) are aligned to the reference's signs.
        l64 = g["logits"].double().requires_grad_(True)
        o64 = torch.cat([O.ransac_train_batch(g["matches"].double(), l64, x.double(), g["gt"].double(), "f8")[0]
                         for x in g["gumbels"]])
        sref = torch.sign((o64.detach() * ref.double()).sum((-1, -2)))
        (o64 * g["grad_weight"].double() * sref[:, None, None]).sum().backward()
        (chosen * (g["grad_weight"] * s[:, None, None]).to(dev)).sum().backward()
        gl = logits.grad.cpu().double()
        assert (gl - l64.grad).abs().max() <= 2e-3 * l64.grad.abs().max(), ((gl - l64.grad).abs().max(), l64.grad.abs().max())
        gr = g["grad_logits"].double()     # the reference's own f32 autograd run
        assert (l64.grad - gr).abs().max() <= 5e-2 * gr.abs().max(), ((l64.grad - gr).abs().max(), gr.abs().max())
    else:
        # five-point: chosen = closest-to-GT real solution per sample; the reference (f32 LAPACK path) is noisy, so the
        # arbiter is the f64 oracle run on the same noise
        for b, x in enumerate(g["gumbels"]):
            idx, ret, _ = O.gumbel_topk(g["logits"], x, 1.0, 5)
            smp = O.gather_samples(g["matches"], ret).double()
            E, ok, real = O.nister_5pt(smp)
            mine = models[32 * b].detach().cpu().double()
            assert mine.shape[0] == 32
            for i in range(32):
                cand = O.canonical(E[i][real[i]])
                d = (O.canonical(mine[i])[None] - cand).abs().amax((-1, -2)).min()
                assert d < 1e-4, float(d)
        (chosen ** 2).sum().backward()
        assert torch.isfinite(logits.grad).all()


@pytest.mark.parametrize("name", ["nister", "f8"])
def test_test_mode_matches_reference_run(dev, name):
    g = load_golden(f"ransac_test_{name}")
    r = _make(name, 16, False, 5000)
    model, mask, score, iters = r(g["matches"].to(dev), g["logits"].to(dev), g["K1"].to(dev), g["K2"].to(dev), None,
                                  gumbels=[x.to(dev) for x in g["gumbels"]])
    # exact arbiter: the f64 oracle on the same noise (the reference's own f32 run takes two more batches on the
    # five-point fixture because its f32 solver finds one inlier fewer: 432 vs 400 iterations)
    dt = torch.float64
    mo, masko, so, ito = O.ransac_test(g["matches"].to(dt), g["logits"].to(dt), [x.to(dt) for x in g["gumbels"]],
                                       g["K1"].to(dt), g["K2"].to(dt), name)
    assert iters == ito
    assert (mask.cpu() != masko).sum() <= 1
    assert abs(float(score) - so) <= 1e-3 * max(1.0, so)
    assert (O.canonical(model.cpu().double()) - O.canonical(mo)).abs().max() < 1e-4
    # the reference's own f32 run (a different best model may win there: its f32 solver is noisier and, on the five-point
    # fixture, stops two batches later) -- same ballpark, not the arbiter
    assert abs(iters - g["iterations"]) <= 2 * 16
    assert abs(int(mask.sum()) - int(g["best_mask"].sum())) <= 3
    assert abs(float(score) - g["best_score"]) <= 2e-2 * max(1.0, g["best_score"])


def test_batched_matches_per_pair_driver(dev):
    from differentiable_ransac_amd import synth
    from differentiable_ransac_amd.ransac import BatchedRANSAC
    P, N, B = 4, 1000, 128
    data = synth.batch_two_view(P, N, seed0=40)
    noise = [synth.gumbel_noise((P, B, N), seed=50 + r).to(dev) for r in range(3)]
    rn = BatchedRANSAC("nister", ransac_batch_size=B, threshold=0.75, max_iterations=3 * B, refit=True)
    out = rn(data["matches"].to(dev), data["logits"].to(dev), data["K1"].to(dev), data["K2"].to(dev), gumbels=noise)
    for p in range(P):
        m, mask, score, it = O.ransac_test(data["matches"][p].double(), data["logits"][p].double(),
                                           [n[p].cpu().double() for n in noise], data["K1"][p].double(),
                                           data["K2"][p].double(), "nister", max_iterations=3 * B)
        assert int(out["iterations"][p]) == it
        assert abs(float(out["score"][p]) - score) <= 2e-3 * max(1.0, score), (float(out["score"][p]), score)
        assert (out["mask"][p].cpu() != mask).sum() <= 2
        # recovered E close to the ground truth
        d = (O.canonical(out["model"][p].cpu().double()) - O.canonical(data["gt_E"][p].double())).abs().max()
        assert d < 0.05


def test_fivepoint_backward_finite_difference(dev):
    """Implicit-function backward of the five-point solvers vs central finite differences of the f64 CPU oracle
    (each solution tracked by proximity, sign-aligned).  The straight-through sampler makes a finite difference of the
    WHOLE train path meaningless (its forward value is constant in the logits), so the chain is validated link by link:
    sampler+gather in test_gpu_sampler.py, the solver here."""
    from differentiable_ransac_amd import ops, synth
    pair = synth.two_view_pair(90, 200, inlier_ratio=1.0, noise=2e-3, dtype=torch.float64)
    B = 12
    smp = pair["matches"][: 5 * B].reshape(B, 5, 4).contiguous()
    W = torch.randn(B, 10, 3, 3, generator=torch.Generator().manual_seed(4), dtype=torch.float64)
    s32 = smp.float().to(dev).requires_grad_(True)
    E, valid = ops.solve_essential(s32, None, "nister")
    (E * W.float().to(dev) * valid[..., None, None]).sum().backward()
    g = s32.grad.cpu().double()
    E0, v0 = E.detach().cpu().double(), valid.cpu()

    def loss_cpu(xb, b):
        Eo, ok, real = O.nister_5pt(xb[None])
        tot = torch.zeros((), dtype=torch.float64)
        for sl in range(10):
            if not bool(v0[b, sl]):
                continue
            cand = Eo[0][real[0]]
            d1 = (cand - E0[b, sl]).abs().amax((-1, -2))
            d2 = (cand + E0[b, sl]).abs().amax((-1, -2))
            j = torch.minimum(d1, d2).argmin()
            sgn = 1.0 if d1[j] <= d2[j] else -1.0
            tot = tot + (sgn * cand[j] * W[b, sl]).sum()
        return tot

    eps = 2e-5   # the oracle's own solutions are good to ~1e-9: smaller steps drown in that noise
    num = torch.zeros_like(smp)
    base = smp.float().double()
    for b in range(B):
        for k in range(5):
            for d in range(4):
                xp, xm = base[b].clone(), base[b].clone()
                xp[k, d] += eps
                xm[k, d] -= eps
                num[b, k, d] = (loss_cpu(xp, b) - loss_cpu(xm, b)) / (2 * eps)
    rel = (g - num).abs().amax((-1, -2)) / num.abs().amax((-1, -2)).clamp(min=1e-9)
    assert rel.median() < 1e-3 and rel.max() < 2e-2, (rel.median(), rel.max())
    # Stewenius shares the backward (same manifold, same constraints)
    s32b = smp.float().to(dev).requires_grad_(True)
    E2, v2 = ops.solve_essential(s32b, None, "stewenius")
    assert int(v2.sum()) == int(valid.sum())
    (E2 ** 2 * v2[..., None, None]).sum().backward()
    assert torch.isfinite(s32b.grad).all()
    # ... and must give the Nister path's gradient for a loss that does not depend on the order or sign of the solutions
    Wc = torch.tensor([[0.3, -1.2, 0.7], [1.1, 0.4, -0.6], [-0.9, 0.8, 1.5]], device=dev)
    grads = {}
    for solver in ("nister", "stewenius"):
        s_ = smp.float().to(dev).clone().requires_grad_(True)
        E_, v_ = ops.solve_essential(s_, None, solver)
        ((E_ ** 2 * Wc) * v_[..., None, None]).sum().backward()
        grads[solver] = s_.grad
    scale = grads["nister"].abs().amax((-1, -2)).clamp(min=1e-6)
    rel = (grads["nister"] - grads["stewenius"]).abs().amax((-1, -2)) / scale
    assert rel.median() < 1e-4 and rel.max() < 5e-2, (rel.median(), rel.max())


def test_batched_train_path_runs_and_is_finite(dev):
    from differentiable_ransac_amd import synth
    from differentiable_ransac_amd.ransac import BatchedRANSAC
    P, N, B = 3, 400, 64
    data = synth.batch_two_view(P, N, seed0=70, inlier_ratio=0.8)
    lg = data["logits"].to(dev).requires_grad_(True)
    tr = BatchedRANSAC("nister", ransac_batch_size=B, train=True, max_iterations=2 * B)
    chosen, keep = tr(data["matches"].to(dev), lg, gt_model=data["gt_E"].to(dev))
    assert chosen.shape == (P, 2 * B, 3, 3) and keep.shape == (P, 2 * B) and keep.float().mean() > 0.9
    gt = data["gt_E"].to(dev)[:, None]
    loss = torch.minimum(((chosen - gt) ** 2).sum((-1, -2)), ((chosen + gt) ** 2).sum((-1, -2)))[keep].mean()
    loss.backward()
    assert torch.isfinite(lg.grad).all() and float(lg.grad.abs().sum()) > 0


def test_ransac3d_train_matches_reference_run(dev):
    from differentiable_ransac_amd.estimators import RigidTransformationSVDBasedSolver
    from differentiable_ransac_amd.ransac import RANSAC3D
    from differentiable_ransac_amd.samplers import GumbelSoftmaxSampler
    from differentiable_ransac_amd.scorings import MSACScore
    g = load_golden("ransac3d_train")
    r3 = RANSAC3D(RigidTransformationSVDBasedSolver(device="cuda"), GumbelSoftmaxSampler(32, 3, device="cuda"),
                  MSACScore("cuda"), train=True, ransac_batch_size=32, sampler_id=2, max_iterations=64)
    logits = g["logits"].to(dev).requires_grad_(True)
    models, residuals, means, _, iters = r3(g["matches"].to(dev), logits, None,
                                            gumbels=[x.to(dev) for x in g["gumbels"]])
    assert iters == g["iterations"]
    ms = torch.cat([models[k] for k in sorted(models)]).detach().cpu()
    err = (ms - g["models"]).abs().amax((-1, -2))
    assert (err < 1e-4).float().mean() > 0.85 and err.max() < 5e-3      # flag=True: R = I up to the reference's f32 noise (Q9)
    res = torch.cat([residuals[k] for k in sorted(residuals)]).detach().cpu()
    assert ((res - g["residuals"]).abs() / g["residuals"]).max() < 5e-3
    mr = torch.stack([means[k] for k in sorted(means)]).detach().cpu()
    assert (mr - g["mean_residuals"]).abs().max() < 5e-3 * g["mean_residuals"].abs().max()
    sum(means.values()).backward()
    assert torch.isfinite(logits.grad).all() and float(logits.grad.abs().sum()) > 0


def test_match_loss_kernel_vs_reference_formula(dev):
    """dr_episym_fwd/bwd against a torch restatement of loss.py:107-153 / cv_utils.py:680-695 in f64."""
    from differentiable_ransac_amd import synth
    from differentiable_ransac_amd.loss import MatchLoss
    P, N, M = 3, 700, 45
    data = synth.batch_two_view(P, N, seed0=600)
    gen = torch.Generator().manual_seed(8)
    models = data["gt_E"][:, None] + 0.05 * torch.randn(P, M, 3, 3, generator=gen)
    keep = torch.rand(P, M, generator=gen) > 0.3
    mask = data["inliers"].clone()
    md = models.to(dev).requires_grad_(True)
    loss = MatchLoss()(md, data["matches"].to(dev), mask.to(dev), keep.to(dev))
    loss.backward()

    m64 = models.double().requires_grad_(True)
    tot = 0
    for p in range(P):
        x1 = torch.cat((data["matches"][p, mask[p], :2].double(), torch.ones(int(mask[p].sum()), 1, dtype=torch.float64)), 1)
        x2 = torch.cat((data["matches"][p, mask[p], 2:].double(), torch.ones(int(mask[p].sum()), 1, dtype=torch.float64)), 1)
        F = m64[p][keep[p]]
        Fx1 = torch.einsum("mij,nj->mni", F, x1)
        Ftx2 = torch.einsum("mji,nj->mni", F, x2)
        r = (x2[None] * Fx1).sum(-1)
        ys = r ** 2 * (1 / (Fx1[..., 0] ** 2 + Fx1[..., 1] ** 2 + 1e-15) + 1 / (Ftx2[..., 0] ** 2 + Ftx2[..., 1] ** 2 + 1e-15))
        tot = tot + torch.clamp(ys, max=1.0).mean()
    ref = tot / P
    ref.backward()
    assert abs(float(loss) - float(ref)) <= 1e-5 * abs(float(ref))
    g, gr = md.grad.cpu().double(), m64.grad
    assert (g[~keep] == 0).all()
    assert (g - gr).abs().max() <= 2e-4 * gr.abs().max()


def test_match_loss_kernel_vs_reference_golden(dev):
    """the fused residual kernel against the reference's own batch_episym output (tests/golden/episym.npz)"""
    from differentiable_ransac_amd import ops
    g = load_golden("episym")
    m = g["matches"].float()[None].to(dev)
    sums = ops.episym_sums(m, g["inliers"][None].to(dev), g["models"].float()[None].to(dev))
    ref = torch.clamp(g["ys_f64"], max=1.0).sum(1)
    assert ((sums[0].cpu().double() - ref).abs() / ref).max() < 1e-4


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_ransac_init_threshold_and_state(dev, dtype):
    """dr_ransac_init against the torch expression of ransac.py:49-53 (K1[0,0] twice: Q3) and the documented state."""
    from differentiable_ransac_amd import ops
    from differentiable_ransac_amd.ransac import normalized_threshold
    P, N = 5, 77
    g = torch.Generator().manual_seed(3)
    K1 = torch.eye(3, dtype=dtype).repeat(P, 1, 1)
    K2 = torch.eye(3, dtype=dtype).repeat(P, 1, 1)
    K1[:, 0, 0] = 900 + 200 * torch.rand(P, generator=g, dtype=dtype)
    K1[:, 1, 1] = 900 + 200 * torch.rand(P, generator=g, dtype=dtype)
    K2[:, 0, 0] = 5000.0                                  # must not enter the formula
    K2[:, 1, 1] = 900 + 200 * torch.rand(P, generator=g, dtype=dtype)
    st, thr = ops.ransac_init(P, N, 5000, 0.75, K1.to(dev), K2.to(dev), dev, dtype)
    want = normalized_threshold(0.75, K1, K2, False)
    assert torch.allclose(thr.cpu(), want, rtol=4 * torch.finfo(dtype).eps, atol=0)
    assert (st.best_score == 0).all() and (st.iters == 0).all() and (st.best_inliers == 0).all()
    assert (st.max_iters == 5000.0).all() and not st.best_mask.any()
    assert torch.equal(st.best_model.cpu(), torch.eye(3, dtype=dtype).repeat(P, 1, 1))
    # shared [3,3] calibration and the no-calibration (fundamental matrix) mode
    _, thr1 = ops.ransac_init(P, N, 100, 0.75, K1[0].to(dev), K2[0].to(dev), dev, dtype)
    assert torch.allclose(thr1.cpu(), want[0].expand(P), rtol=4 * torch.finfo(dtype).eps, atol=0)
    _, thr2 = ops.ransac_init(P, N, 100, 0.75, None, None, dev, dtype)
    assert torch.equal(thr2.cpu(), torch.full((P,), 0.75, dtype=dtype))


def test_dropin_fused_path_equals_plugin_path(dev):
    """RANSAC.__call__ in test mode: the device-resident batched driver (fused=True, the default when the plugins are
    this package's) and the per-batch plugin path give the same result for the same explicit noise."""
    g = load_golden("ransac_test_nister")
    args = (g["matches"].to(dev), g["logits"].to(dev), g["K1"].to(dev), g["K2"].to(dev), None)
    noise = [x.to(dev) for x in g["gumbels"]]
    out = {}
    for fused in (True, False):
        r = _make("nister", 16, False, 5000)
        r.fused = fused
        out[fused] = r(*args, gumbels=noise)
    (ma, ka, sa, ia), (mb, kb, sb, ib) = out[True], out[False]
    assert ia == ib
    assert torch.equal(ka, kb)
    assert abs(float(sa) - float(sb)) <= 1e-4 * max(1.0, abs(float(sb)))
    assert (O.canonical(ma.cpu().double()) - O.canonical(mb.cpu().double())).abs().max() < 1e-5


def test_batched_topdown_sampling_recovers_the_pose(dev):
    """BatchedRANSAC with the top-down draw of the index sets: same quality of result as with the Gumbel sampler."""
    from differentiable_ransac_amd import synth
    from differentiable_ransac_amd.ransac import BatchedRANSAC
    P, N, B = 8, 2000, 1024
    data = synth.batch_two_view(P, N, seed0=70)
    args = (data["matches"].to(dev), data["logits"].to(dev), data["K1"].to(dev), data["K2"].to(dev))
    res = {}
    for samp in ("gumbel", "topdown"):
        rn = BatchedRANSAC("nister", ransac_batch_size=B, threshold=0.75, max_iterations=B, refit=True, sampling=samp)
        out = rn(*args)
        d = (O.canonical(out["model"].cpu().double()) - O.canonical(data["gt_E"].double())).abs().amax((-1, -2))
        res[samp] = (d, out["inliers"].float().mean().item())
        assert d.max() < 0.05 and (out["iterations"] == B).all()
    assert abs(res["gumbel"][1] - res["topdown"][1]) < 0.05 * res["gumbel"][1]
    with pytest.raises(ValueError):
        BatchedRANSAC("nister", train=True, sampling="topdown")


def test_ransac_layers_forward(dev):
    """Row H: RANSACLayer / RANSACLayer3D forward = the driver call + the reference's post-processing (concatenate the
    per-batch models, drop NaN rows, wall time), and the batched counterpart of the per-pair loop."""
    import types
    from differentiable_ransac_amd import synth
    from differentiable_ransac_amd.layers import RANSACLayer, RANSACLayer3D, batched_forward
    g = load_golden("ransac_train_nister")
    opt = types.SimpleNamespace(fmat=False, sampler=2, ransac_batch_size=32, tr=True, weighted=0, threshold=0.75, precision=1,
                                device="cuda")
    layer = RANSACLayer(opt)
    logits = g["logits"].to(dev).requires_grad_(True)
    Es, secs = layer(g["matches"].to(dev), logits, g["K1"].to(dev), g["K2"].to(dev), None, None, g["gt"].to(dev),
                     gumbels=[x.to(dev) for x in g["gumbels"]])
    direct = _make("nister", 32, True, 100)
    models, _, _, _ = direct(g["matches"].to(dev), g["logits"].to(dev), g["K1"].to(dev), g["K2"].to(dev), g["gt"].to(dev),
                             gumbels=[x.to(dev) for x in g["gumbels"]])
    want = torch.cat([models[k] for k in sorted(models.keys())])
    assert Es.shape == want.shape and torch.allclose(Es, want, atol=1e-6) and secs > 0
    Es.sum().backward()
    assert torch.isfinite(logits.grad).all() and logits.grad.abs().sum() > 0
    # test mode: one model per pair; the batched counterpart recovers the same geometry for every pair of a batch
    opt.tr, opt.ransac_batch_size = False, 512
    P, N = 4, 1000
    data = synth.batch_two_view(P, N, seed0=90)
    ret, per_pair = batched_forward(opt, data["matches"].to(dev), data["logits"].to(dev), data["K1"].to(dev), data["K2"].to(dev))
    single = RANSACLayer(opt)
    for p in range(P):
        E1, _ = single(data["matches"][p].to(dev), data["logits"][p].to(dev), data["K1"][p].to(dev), data["K2"][p].to(dev),
                       None, None)
        for E in (E1, ret[p]):
            assert E.shape == (3, 3)
            assert (O.canonical(E.cpu().double()) - O.canonical(data["gt_E"][p].double())).abs().max() < 0.05
    # 3-D layer, train mode, against the reference run
    g3 = load_golden("ransac3d_train")
    opt3 = types.SimpleNamespace(fmat=False, sampler=2, ransac_batch_size=32, tr=True, weighted=0, threshold=0.75, precision=1,
                                 device="cuda")
    l3 = RANSACLayer3D(opt3)
    l3.estimator.max_iterations = 64
    Ts, loss, avg_loss, secs = l3(g3["matches"].to(dev), g3["logits"].to(dev), None, gumbels=[x.to(dev) for x in g3["gumbels"]])
    err = (Ts.cpu() - g3["models"]).abs().amax((-1, -2))
    assert (err < 1e-4).float().mean() > 0.85 and err.max() < 5e-3      # flag=True: R = I up to the reference's f32 noise (Q9)
    assert abs(float(loss) - float(g3["residuals"].mean())) < 5e-3 * float(g3["residuals"].mean())
    assert abs(float(avg_loss) - float(g3["mean_residuals"].mean())) < 5e-3 * float(g3["mean_residuals"].abs().max())
