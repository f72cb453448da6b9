"""Round 6 on the GPU: super-rounds -- a device round of R consecutive batches of `ransac_batch_size` hypotheses, walked IN ORDER by
dr_ransac_update with the stop rule of ransac.py:55-144 -- against the batch-by-batch loop they replace; the drop-in call at the
reference's default batch size (`-rbs 64`, utils.py:33) as one replayed graph.  Reference: ransac.py:55-200, model_cl.py:488-511."""
import pytest
import torch

from tests.conftest import load_golden
from tests.test_gpu_round5 import _hard_pairs

pytestmark = pytest.mark.gpu

KEYS = ("model", "mask", "score", "inliers", "iterations")


@pytest.mark.parametrize("solver,B", [("nister", 64), ("stewenius", 64), ("f8", 64), ("nister", 16), ("nister", 100)])
def test_super_rounds_equal_the_batch_by_batch_loop(dev, solver, B):
    """same base seed, three schedules: one batch per device round (the host loop of rounds 1-5), the automatic rounds of 1024
    hypotheses, and (1024, 4096) as the drop-in's graph uses -- in-kernel noise keyed per batch, so the hypotheses are the same and
    (model, mask, score, inliers, iterations) must be equal bit for bit, with the stop read back on the host and taken on the device"""
    from differentiable_ransac_amd.ransac import BatchedRANSAC
    m, lg, K1, K2 = _hard_pairs(dev, 4, pixel=solver == "f8")
    kw = dict(ransac_batch_size=B, threshold=0.75, max_iterations=3000, seed=21, refit=True)
    out = {}
    for name, sh, devt in (("loop", False, False), ("auto", None, False), ("auto_dev", None, True), ("graph_plan", (1024, 4096), True)):
        drv = BatchedRANSAC(solver, **kw)
        drv.super_hypotheses = sh
        drv.device_termination = devt
        n = len(drv.plan())
        if sh is False:
            assert n == -(-3000 // B)
        else:
            assert n <= 4
        for call in range(2):                       # the second call: seeds advanced by the number of BATCHES, not of device rounds
            out[name, call] = drv(m, lg, K1, K2)
    for name in ("auto", "auto_dev", "graph_plan"):
        for call in range(2):
            for key in KEYS:
                assert torch.equal(out["loop", call][key], out[name, call][key]), (name, call, key)
    its = out["loop", 0]["iterations"].tolist()
    assert all(i % B == 0 for i in its) and max(its) > 1024 and (solver == "f8" or min(its) < max(its)), its


def test_super_round_sampler_rows_are_the_batches_rows(dev):
    """dr_gumbel_topk_gather_f32(sub = B): row b of the super-round = row b % B of the call with seed + b // B, for the
    register kernel (N = 2000), the general kernel (N = 1999: not a multiple of 4) and the streaming kernel (N = 4096)"""
    from differentiable_ransac_amd import ops
    torch.manual_seed(0)
    for N in (2000, 1999, 4096):
        m = torch.rand(3, N, 4, device=dev)
        lg = torch.randn(3, N, device=dev)
        B, R, seed = 48, 5, 0xFFFFFFFFFFFFFFFE           # (the seed wraps around 2^64 inside the round)
        idx, smp = ops.gumbel_topk_gather(m, lg, R * B, 5, 1.0, seed, sub=B)      # (N = 1999: the general kernel + a gather launch)
        for j in range(R):
            ij, sj = ops.gumbel_topk_gather(m, lg, B, 5, 1.0, (seed + j) & (2 ** 64 - 1))
            assert torch.equal(idx[:, j * B:(j + 1) * B], ij), (N, j)
            assert torch.equal(smp[:, j * B:(j + 1) * B], sj), (N, j)
        # a device seed (graph replay): the same rows
        ds = ops.DeviceSeed(7, dev)
        seeds = ds.next_n(R)
        idx_d, _ = ops.gumbel_topk_gather(m, lg, R * B, 5, 1.0, seeds[0], sub=B)
        for j in range(R):
            ij, _ = ops.gumbel_topk_gather(m, lg, B, 5, 1.0, seeds[j])
            assert torch.equal(idx_d[:, j * B:(j + 1) * B], ij), (N, j)


def test_ransac_update_walks_sub_batches_in_order(dev):
    """dr_ransac_update(sub_models): hand-made scores -- the walk must take the FIRST arg-max of every sub-batch in order, update only
    on a strictly better score (or at iteration 0), recompute the bound after every update and stop where the loop stops"""
    from differentiable_ransac_amd import ops, synth
    from oracle import cpu_ref as O
    pair = synth.two_view_pair(5, 512, inlier_ratio=0.6)
    m = pair["matches"][None].to(dev)
    B, S, R = 8, 10, 12
    g = torch.Generator().manual_seed(3)
    models = torch.randn(1, R * B * S, 3, 3, generator=g).to(dev)
    models[0, 3 * B * S + 7] = pair["gt_E"].float().to(dev)            # the true model sits in sub-batch 3
    valid = (torch.rand(1, R * B * S, generator=g) > 0.3).to(dev)
    valid[0, 3 * B * S + 7] = True
    thr = torch.full((1,), 2e-3, device=dev)
    scores, _ = ops.msac_score(m, models, thr, want_masks=False, valid=valid)
    # reference walk on the host: one ops.ransac_update per sub-batch on a second state
    st_a, _ = ops.ransac_init(1, 512, 5000, 2e-3, None, None, dev, torch.float32)
    st_b, _ = ops.ransac_init(1, 512, 5000, 2e-3, None, None, dev, torch.float32)
    ops.ransac_update(st_a, m, models, valid, scores, thr, B, 5, sub_models=B * S)
    for j in range(R):
        sl = slice(j * B * S, (j + 1) * B * S)
        ops.ransac_update(st_b, m, models[:, sl].contiguous(), valid[:, sl].contiguous(), scores[:, sl].contiguous(), thr, B, 5)
    for name in ("best_score", "best_model", "best_mask", "best_inliers", "iters", "max_iters"):
        assert torch.equal(getattr(st_a, name), getattr(st_b, name)), name
    it = int(st_a.iters[0])
    assert it < R * B and it >= 4 * B, it                # the true model (60 % inliers) ends the walk before the last sub-batch
    assert float(st_a.max_iters[0]) <= it


def test_dropin_call_at_the_reference_default_batch_size_is_one_graph(dev):
    """`-rbs 64`, max_iterations 5000 (utils.py:33, model_cl.py:216-219): 79 batches = two device rounds in one replayed graph; pair
    after pair the results of the eager batch-by-batch driver with the same base seed; K = None uses the threshold as is, also after
    a call with intrinsics (round-5 advice: the graph's staged K buffers)"""
    from differentiable_ransac_amd import estimators, samplers, scorings
    from differentiable_ransac_amd.ransac import RANSAC, BatchedRANSAC
    m, lg, K1, K2 = _hard_pairs(dev, 5)
    smp = samplers.GumbelSoftmaxSampler(64, 5, device=dev, seed=9)
    rn = RANSAC(estimators.EssentialMatrixEstimatorNister(dev), smp, scorings.MSACScore(dev), train=False,
                ransac_batch_size=64, sampler_id=2, threshold=0.75, max_iterations=5000)
    base = (9 * 0x9E3779B97F4A7C15) & (2 ** 64 - 1)
    ref = BatchedRANSAC("nister", ransac_batch_size=64, threshold=0.75, max_iterations=5000, seed=base, refit=True)
    ref.super_hypotheses = False                            # 79 device rounds of one batch, stop read back on the host
    ref.calls = 2 * 79                                      # the two warm-up calls of the capture drew 79 seeds each
    for p in range(5):
        model, mask, score, iters = rn(m[p], lg[p], K1[p], K2[p], None)
        want = ref(m[p:p + 1], lg[p:p + 1], K1[p:p + 1], K2[p:p + 1])
        ref.calls = (3 + p) * 79                            # a stopped host loop consumed fewer seeds than the graph's 79 per call
        assert torch.equal(model, want["model"][0]) and torch.equal(mask, want["mask"][0]), p
        assert torch.equal(score, want["score"][0]) and int(iters) == int(want["iterations"][0]), p
    assert len(rn._graphs) == 1 and rn._graph_rounds == 2
    # no intrinsics after a call with intrinsics
    thr_px = 0.75 / 800.0
    rn2 = RANSAC(estimators.EssentialMatrixEstimatorNister(dev), samplers.GumbelSoftmaxSampler(64, 5, device=dev, seed=9),
                 scorings.MSACScore(dev), train=False, ransac_batch_size=64, sampler_id=2, threshold=thr_px, max_iterations=5000)
    rn2(m[0], lg[0], K1[0], K2[0], None)
    _, _, score_none, _ = rn2(m[1], lg[1], None, None, None)
    eager = RANSAC(estimators.EssentialMatrixEstimatorNister(dev), samplers.GumbelSoftmaxSampler(64, 5, device=dev, seed=9),
                   scorings.MSACScore(dev), train=False, ransac_batch_size=64, sampler_id=2, threshold=thr_px, max_iterations=5000)
    eager.graph = False
    eager.sampler.calls = 3                                  # rn2's graph took one sampler seed at capture, then one per... see below
    _, mask_e, score_e, _ = eager(m[1], lg[1], None, None, None)
    # different seeds (the eager path advances the sampler's counter differently): compare the SCALE of the score, which the
    # threshold sets -- with stale intrinsics the threshold would be 800x smaller and next to nothing would be an inlier
    assert float(score_none) > 0.5 * float(score_e) > 0


def test_graph_cache_is_bounded(dev):
    """one captured call per point count, least recently used evicted (round-5 advice: variable-N inputs)"""
    from differentiable_ransac_amd import estimators, samplers, scorings, synth
    from differentiable_ransac_amd.ransac import RANSAC
    rn = RANSAC(estimators.EssentialMatrixEstimatorNister(dev), samplers.GumbelSoftmaxSampler(256, 5, device=dev),
                scorings.MSACScore(dev), train=False, ransac_batch_size=256, sampler_id=2, threshold=0.75, max_iterations=256)
    rn.max_graphs = 2
    for N in (400, 512, 640, 512, 400):
        pair = synth.two_view_pair(N, N)
        model, mask, score, _ = rn(pair["matches"].to(dev), pair["logits"].to(dev), pair["K1"].to(dev), pair["K2"].to(dev), None)
        assert mask.shape == (N,) and float(score) > 0
        assert len(rn._graphs) <= 2
    assert [k[0] for k in rn._graphs] == [512, 400]


@pytest.mark.parametrize("name", ["nister", "f8"])
def test_reference_test_run_through_super_rounds(dev, name):
    """ransac_test_{nister,f8}.npz: the reference's own test-mode run (27 batches of 16 with its recorded noise) as ONE device round
    of 27 sub-batches with the stop taken on the device: bit-equal to the batch-by-batch loop on the same noise, and -- like
    tests/test_gpu_drivers.py::test_test_mode_matches_reference_run -- the f64 oracle's (iterations, mask, score, model) on that
    noise (the arbiter: the reference's f32 run stops two batches later on the five-point fixture), the reference's in the ballpark"""
    from differentiable_ransac_amd.ransac import BatchedRANSAC
    from oracle import cpu_ref as O
    g = load_golden(f"ransac_test_{name}")
    noise = [x[None].to(dev) for x in g["gumbels"]]
    B = noise[0].shape[1]
    args = (g["matches"][None].to(dev), g["logits"][None].to(dev), g["K1"][None].to(dev), g["K2"][None].to(dev))
    kw = dict(ransac_batch_size=B, threshold=0.75, max_iterations=5000, refit=True, num_samples=8 if name == "f8" else None)
    drv = BatchedRANSAC(name, **kw)
    drv.device_termination = True
    assert drv.plan(len(noise)) == [len(noise)]
    out = drv(*args, gumbels=noise)
    loop = BatchedRANSAC(name, **kw)
    loop.super_hypotheses = False
    want = loop(*args, gumbels=noise)
    for key in KEYS:
        assert torch.equal(out[key], want[key]), key
    dt = torch.float64
    mo, masko, so, ito = O.ransac_test(g["matches"].to(dt), g["logits"].to(dt), [x.to(dt) for x in g["gumbels"]],
                                       g["K1"].to(dt), g["K2"].to(dt), name)
    assert int(out["iterations"][0]) == ito
    assert (out["mask"][0].cpu() != masko).sum() <= 1
    assert abs(float(out["score"][0]) - so) <= 1e-3 * max(1.0, so)
    assert (O.canonical(out["model"][0].cpu().double()) - O.canonical(mo)).abs().max() < 1e-4
    assert abs(int(out["iterations"][0]) - int(g["iterations"])) <= 2 * B
    assert abs(int(out["mask"][0].sum()) - int(g["best_mask"].sum())) <= 3


def test_batched_forward_has_no_per_pair_synchronisation(dev):
    """layers.batched_forward (model_cl.py:240-242,488-511 as one call): train mode hands out the ragged per-pair model lists from ONE
    read-back, the F branch de-normalises all pairs in one expression -- no aten::nonzero / aten::item per pair in a profiler trace of
    a 32-pair call; the lists equal the per-pair boolean-mask gathers, gradients flow to the logits"""
    import types
    from torch.profiler import ProfilerActivity, profile
    from differentiable_ransac_amd import synth
    from differentiable_ransac_amd.layers import batched_forward, denormalize_pts
    from differentiable_ransac_amd.ransac import BatchedRANSAC
    P, N, B = 32, 512, 64
    data = synth.batch_two_view(P, N, seed0=130)
    m, lg, K1, K2, gt = (data[k].to(dev) for k in ("matches", "logits", "K1", "K2", "gt_E"))
    opt = types.SimpleNamespace(fmat=False, sampler=2, ransac_batch_size=B, tr=True, weighted=0, threshold=0.75, precision=1, device="cuda")
    drv = BatchedRANSAC("nister", ransac_batch_size=B, train=True, threshold=0.75, max_iterations=100, seed=4)
    lg_a = lg.clone().requires_grad_(True)
    batched_forward(opt, m, lg_a, K1, K2, gt=gt, driver=drv)             # warm-up (first-call attributes, allocator)
    drv.calls = 0
    with profile(activities=[ProfilerActivity.CPU]) as prof:
        ret, _ = batched_forward(opt, m, lg_a, K1, K2, gt=gt, driver=drv)
    names = [e.name for e in prof.events()]
    assert names.count("aten::nonzero") == 0, names.count("aten::nonzero")
    assert names.count("aten::item") + names.count("aten::_local_scalar_dense") <= 2 * 2     # the ONE counts.tolist()
    # the same call by hand: boolean-mask gather per pair
    drv.calls = 0
    chosen, keep = drv(m, lg.clone().requires_grad_(True), K1, K2, gt_model=gt)     # (the same autograd path: logits that require grad)
    assert len(ret) == P
    for p in range(P):
        assert torch.equal(ret[p], chosen[p][keep[p]]), p
    torch.cat(ret).square().sum().backward()
    assert torch.isfinite(lg_a.grad).all() and float(lg_a.grad.abs().sum()) > 0
    # F branch, test mode: the vectorised de-normalisation = cv_utils.denormalize_pts per pair
    dpx = synth.batch_two_view(4, 256, seed0=140)
    im1 = torch.tensor([[480.0, 640.0], [600.0, 800.0], [1000.0, 1000.0], [768.0, 1024.0]], device=dev)
    im2 = im1.flip(0).contiguous()
    norm = dpx["matches"].to(dev) / 1000.0
    optf = types.SimpleNamespace(fmat=True, sampler=3, ransac_batch_size=64, tr=False, weighted=0, threshold=0.75, precision=1, device="cuda")
    seen = {}

    class Spy(BatchedRANSAC):
        def __call__(self, pts, *a, **k):
            seen["pts"] = pts
            return super().__call__(pts, *a, **k)
    retf, _ = batched_forward(optf, norm, dpx["logits"].to(dev), None, None, im1, im2,
                              driver=Spy("f8", ransac_batch_size=64, threshold=0.75, max_iterations=128))
    assert len(retf) == 4 and retf[0].shape == (3, 3)
    for p in range(4):
        want = torch.cat((denormalize_pts(norm[p, :, 0:2], im1[p]), denormalize_pts(norm[p, :, 2:4], im2[p])), -1)
        assert torch.equal(seen["pts"][p], want), p


def test_one_logarithm_sampler_draws_the_same_index_sets(dev):
    """K1 in the exponential-race form (round 6: key = exp(lmax - logit) * log2 u, one logarithm per element) against the two-logarithm
    form of rounds 1-5 (logit - ln(-ln u)) and against the f32 oracle on the noise the general kernel dumps for the same seed
    (gumbel_sampler.py:30-36): same index sets (32 x 1024 x 2000 here; 0 of 131 072 rows differed at 128 pairs on two seeds,
    scratch/k1_race_check.py); wild logits (one-hot, non-finite, a span beyond 80) keep the two-logarithm form, bit for bit"""
    from differentiable_ransac_amd import ops, synth
    P, B, N, k = 32, 1024, 2000, 5
    d = synth.batch_two_view(P, N, seed0=11)
    m, lg = d["matches"].to(dev), d["logits"].to(dev)
    ia, sa = ops.gumbel_topk_gather(m, lg, B, k, 1.0, 99, race=True)
    ib, sb = ops.gumbel_topk_gather(m, lg, B, k, 1.0, 99, race=False)
    differ = int((ia != ib).any(-1).sum())
    assert differ <= 1, differ                                   # (near-ties at rounding level are the only legitimate differences)
    assert torch.equal(sa[(ia == ib).all(-1)], sb[(ia == ib).all(-1)])
    r = ops.gumbel_topk(lg[:8], B, k, 1.0, None, 99, want_noise=True)
    top = torch.topk(lg[:8, None, :] + r["gumbel"], k, dim=-1).indices.sort(-1).values.int()
    assert int((top != ia[:8]).any(-1).sum()) <= 1
    # sub-batches (super-rounds) through the race form: row b = row b % sub of the call seeded seed + b // sub
    isub, _ = ops.gumbel_topk_gather(m, lg, 4 * 64, k, 1.0, 5, sub=64, race=True)
    for j in range(4):
        ij, _ = ops.gumbel_topk_gather(m, lg, 64, k, 1.0, 5 + j, race=True)
        assert torch.equal(isub[:, 64 * j:64 * (j + 1)], ij), j
    # wild logits: the flagged pairs take the two-logarithm form
    wild = lg.clone()
    wild[0] = -1000.0
    wild[0, 17] = 0.0                                             # one-hot: span 1000
    wild[1, 5] = float("-inf")
    wild[2, 7] = float("nan")
    wild[3] = torch.linspace(-100.0, 0.0, N, device=dev)          # span 100 > 80
    iw, _ = ops.gumbel_topk_gather(m, wild, B, k, 1.0, 3, race=True)
    ix, _ = ops.gumbel_topk_gather(m, wild, B, k, 1.0, 3, race=False)
    assert torch.equal(iw[:4], ix[:4])
    assert int((iw[4:] != ix[4:]).any(-1).sum()) <= 1
    assert (iw[0] == 17).any(-1).all() and not (iw[1] == 5).any() and not (iw[2] == 7).any()


@pytest.mark.parametrize("shape", [(6, 512, 2000, 5), (3, 256, 64, 5), (4, 300, 1000, 8), (2, 1024, 2048, 1), (5, 128, 260, 2)])
def test_one_logarithm_sampler_selection_on_wave_masks(dev, shape):
    """The race form selects on compare masks (round 6, gumbel_topk.hip `DR_K1_SALU_SELECT`): a threshold search on the count of lane
    maxima, candidates dealt to lanes in index order, ranking only when more than k pass.  Every branch of it against the two-logarithm
    form (whose selection is the candidate list of rounds 2-5; identical to the mask selection on 99 cases, scratch/k1_select_check.py):
    ordinary logits; flat logits; a few dominant points (the search cannot bracket the count: the row takes the list path);
    quantised logits; a span of 79 (the largest the form accepts)"""
    from differentiable_ransac_amd import ops, synth
    P, B, N, k = shape
    d = synth.batch_two_view(P, N, seed0=3)
    m, lg = d["matches"].to(dev), d["logits"].to(dev)
    dom = lg.clone()
    dom[:, [3, N // 2, N - 1]] += 30.0
    cases = {"synthetic": lg, "flat": torch.zeros_like(lg), "dominant": dom, "quantised": torch.round(lg),
             "span79": lg / lg.abs().max() * 39.5, "peaked": lg * 8.0}
    for tag, l2 in cases.items():
        ia, sa = ops.gumbel_topk_gather(m, l2, B, k, 1.0, 21, race=True)
        ib, sb = ops.gumbel_topk_gather(m, l2, B, k, 1.0, 21, race=False)
        assert (ia[..., 1:] > ia[..., :-1]).all() and ia.min() >= 0 and ia.max() < N, tag      # k distinct points, ascending
        same = (ia == ib).all(-1)
        assert int((~same).sum()) <= max(1, P * B // 20000), (tag, int((~same).sum()))       # (near-ties at rounding level)
        assert torch.equal(sa[same], sb[same]), tag
        assert torch.equal(sa, torch.gather(m, 1, ia.reshape(P, B * k, 1).expand(-1, -1, 4).long()).reshape(P, B, k, 4)), tag
        if tag == "dominant" and k >= 3:
            assert all((ia == j).any(-1).float().mean() > 0.99 for j in (3, N // 2, N - 1))


def test_one_logarithm_sampler_in_train_mode(dev, monkeypatch):
    """Train mode through the one-logarithm form (round 6: dr_gumbel_topk_gather_soft_f32's race_ws): the keys w_n log2 u_n give the
    winners AND the soft-max statistics -- y_n = (1 / -key_n) / sum_m (1 / -key_m), lse = lmax + ln sum - ln ln 2 (gumbel_sampler.py:
    33-38 restated) -- against the two-logarithm form of the same launch: index sets, weights, log-sum-exps, samples, and the gradient
    to the logits through the unchanged backward launch; wild pairs (non-finite logits, a span beyond 80) keep the other form, bit
    for bit"""
    from differentiable_ransac_amd import ops, synth
    P, B, N, k = 32, 1024, 2000, 5
    d = synth.batch_two_view(P, N, seed0=5)
    m = d["matches"].to(dev)
    lg0 = d["logits"].to(dev)
    lg0[1] = torch.linspace(-100.0, 0.0, N, device=dev)           # span 100 > 80: keeps the two-logarithm form
    lg0[2, 9] = float("-inf")
    out = {}
    for on in (True, False):
        monkeypatch.setattr(ops, "K1_RACE_SOFT", on)
        lg = lg0.clone().requires_grad_(True)
        smp, y, idx = ops.SampleGather.apply(m, lg, B, k, 1.0, None, 77)
        wgt = torch.linspace(0.5, 1.5, P * B * k * 4, device=dev).reshape(P, B, k, 4)
        (smp * wgt).sum().backward()
        out[on] = (idx.clone(), y.detach().clone(), smp.detach().clone(), lg.grad.clone())
    (ia, ya, sa, ga), (ib, yb, sb, gb) = out[True], out[False]
    same = (ia == ib).all(-1)
    assert int((~same).sum()) <= 2, int((~same).sum())            # (near-ties at rounding level)
    assert torch.equal(ia[1:3], ib[1:3]) and torch.equal(ya[1:3], yb[1:3]) and torch.equal(sa[1:3], sb[1:3])
    rel = ((ya - yb).abs() / yb.abs().clamp_min(1e-30))[same]
    assert float(rel.max()) < 5e-5, float(rel.max())
    assert float((sa - sb).abs()[same].max()) < 1e-5
    fin = torch.isfinite(gb)
    assert torch.equal(torch.isfinite(ga), fin)
    assert float((ga - gb)[fin].abs().max()) <= 2e-4 * float(gb[fin].abs().max()), float((ga - gb)[fin].abs().max())
    # the statistics against their definition: y = softmax(logits + G)[idx] on the noise the general kernel dumps for this seed
    r = ops.gumbel_topk(lg0[:4], B, k, 1.0, None, 77, want_noise=True)
    monkeypatch.setattr(ops, "K1_RACE_SOFT", True)
    _, y4, i4 = ops.SampleGather.apply(m[:4].contiguous(), lg0[:4].contiguous(), B, k, 1.0, None, 77)   # (4 pairs: below the automatic choice)
    soft = torch.softmax((lg0[:4, None, :] + r["gumbel"]).double(), -1)
    want = torch.gather(soft, 2, ia[:4].long())
    ok = (ia[:4] == r["idx"]).all(-1)
    ok[2] = False                                                    # (the pair with a -inf logit is compared above)
    assert float(((ya[:4].double() - want).abs() / want)[ok].max()) < 5e-5


@pytest.mark.parametrize("shape", [(3, 256, 64, 5), (4, 300, 1000, 8), (2, 512, 2048, 1), (5, 130, 260, 3)])
def test_one_logarithm_sampler_in_train_mode_shapes(dev, monkeypatch, shape):
    """the same through short rows, k = 1 and k = 8, a row count that is no multiple of the block's four rows -- the automatic choice
    forced on (ops._RACE_MIN), flat / quantised / dominant logits: winners, weights, log-sum-exps and the weights' sum rule"""
    from differentiable_ransac_amd import ops, synth
    P, B, N, k = shape
    monkeypatch.setattr(ops, "_RACE_MIN", (1, 1))
    d = synth.batch_two_view(P, N, seed0=9)
    m, lg = d["matches"].to(dev), d["logits"].to(dev)
    dom = lg.clone()
    dom[:, [1, N // 2]] += 25.0
    for tag, l2 in {"synthetic": lg, "flat": torch.zeros_like(lg), "quantised": torch.round(lg), "dominant": dom}.items():
        out = {}
        for on in (True, False):
            monkeypatch.setattr(ops, "K1_RACE_SOFT", on)
            smp, y, idx = ops.SampleGather.apply(m, l2.contiguous(), B, k, 1.0, None, 31)
            out[on] = (idx, y, smp)
        (ia, ya, sa), (ib, yb, sb) = out[True], out[False]
        same = (ia == ib).all(-1)
        assert int((~same).sum()) <= max(1, P * B // 10000), (tag, int((~same).sum()))
        assert (ia[..., 1:] > ia[..., :-1]).all() and ia.min() >= 0 and ia.max() < N, tag
        rel = ((ya - yb).abs() / yb.abs().clamp_min(1e-30))[same]
        assert float(rel.max()) < 1e-4, (tag, float(rel.max()))
        assert float((sa - sb).abs()[same].max()) < 1e-4, tag
        assert float(ya.sum(-1).max()) <= 1.0 + 1e-5 and float(ya.min()) >= 0.0, tag


@pytest.mark.parametrize("tr", [False, True])
def test_layer_leaves_its_inputs_untouched(dev, tr):
    """The E branch of RANSACLayer.forward hands its input on without the copy of model_cl.py:239 (round 6): points, logits and the
    calibration matrices read the same after the call as before, in test mode (replayed graph and eager) and in train mode"""
    import types
    from differentiable_ransac_amd import layers, synth
    d = synth.batch_two_view(2, 2000, seed0=21)
    m, lg, K1, K2, gt = (d[k].to(dev) for k in ("matches", "logits", "K1", "K2", "gt_E"))
    im = torch.tensor([1000.0, 1000.0], device=dev)
    opt = types.SimpleNamespace(fmat=False, sampler=2, ransac_batch_size=1024, tr=tr, weighted=0, threshold=0.75, precision=1,
                                device=str(dev))
    layer = layers.RANSACLayer(opt)
    for graph in ((True, False) if not tr else (False,)):
        layer.estimator.graph = graph
        for p in range(2):
            before = [t.clone() for t in (m[p], lg[p], K1[p], K2[p])]
            Es, _ = layer(m[p], lg[p], K1[p], K2[p], im, im, gt[p] if tr else None)
            assert torch.isfinite(Es).all()
            for was, now in zip(before, (m[p], lg[p], K1[p], K2[p])):
                assert torch.equal(was, now)


@pytest.mark.parametrize("rbs", [64, 1024])
def test_dropin_fundamental_call_as_a_graph_equals_the_batch_by_batch_driver(dev, rbs):
    """`-fmat 1 -sam 3 -tr 0` through the replayed call (packed one-pair state, LSQ refit on the inliers of the best mask that lives in
    that buffer): pair after pair the results of the eager batch-by-batch driver with the same base seed"""
    from differentiable_ransac_amd import estimators, samplers, scorings
    from differentiable_ransac_amd.ransac import RANSAC, BatchedRANSAC
    m, lg, K1, K2 = _hard_pairs(dev, 3, pixel=True)
    n_batches = -(-1000 // rbs)
    rn = RANSAC(estimators.FundamentalMatrixEstimatorNew(dev), samplers.GumbelSoftmaxSampler(rbs, 8, device=dev, seed=5),
                scorings.MSACScore(dev), fmat=True, train=False, ransac_batch_size=rbs, sampler_id=3, threshold=0.75,
                max_iterations=1000)
    base = (5 * 0x9E3779B97F4A7C15) & (2 ** 64 - 1)
    ref = BatchedRANSAC("f8", ransac_batch_size=rbs, threshold=0.75, max_iterations=1000, seed=base, refit=True)
    ref.super_hypotheses = False
    for p in range(3):
        ref.calls = (2 + p) * n_batches               # two warm-up calls of the capture, then one call per pair, n_batches seeds each
        model, mask, score, iters = rn(m[p], lg[p], K1[p], K2[p], None)
        want = ref(m[p:p + 1], lg[p:p + 1], K1[p:p + 1], K2[p:p + 1])
        assert torch.equal(mask, want["mask"][0]) and int(iters) == int(want["iterations"][0]), p
        assert torch.equal(score, want["score"][0]) and torch.equal(model, want["model"][0]), p
    assert len(rn._graphs) == 1


def test_f64_nonminimal_fivepoint_backward(dev):
    """`-sam 3 -fmat 0 -tr 1 -pr 2` (nister.py:64-65, model_cl.py:164-169): the backward of the NON-minimal five-point solve with
    samples, weights, models and gradients f64 in memory (dr_solve_nister5_nm_bwd_f64; rounds 3-5 went through the f32-I/O kernel)
    against central differences of the f64 forward itself (1e-6 steps) and against the f32-I/O kernel's gradient"""
    from differentiable_ransac_amd import ops, synth
    pair = synth.two_view_pair(91, 200, inlier_ratio=1.0, noise=2e-3, dtype=torch.float64)
    B, n = 8, 8
    smp = pair["matches"][: n * B].reshape(B, n, 4).contiguous().to(dev)
    gen = torch.Generator().manual_seed(5)
    wts = (0.5 + torch.rand(B, n, generator=gen, dtype=torch.float64)).to(dev)
    W = torch.randn(B, 10, 3, 3, generator=gen, dtype=torch.float64).to(dev)
    s64 = smp.clone().requires_grad_(True)
    w64 = wts.clone().requires_grad_(True)
    E, valid = ops.solve_essential(s64, w64, "nister")
    assert E.dtype == torch.float64 and int(valid.sum()) >= B
    (E * W * valid[..., None, None]).sum().backward()
    g, gw = s64.grad.clone(), w64.grad.clone()
    assert g.dtype == torch.float64 and gw.dtype == torch.float64
    E0, v0 = E.detach(), valid

    def loss(x, w):
        Ep, vp = ops.solve_essential(x, w, "nister")
        # slots are ordered by root: a step of 1e-6 keeps the order; sign-align every slot with the unperturbed model
        sgn = torch.sign((Ep * E0).sum((-1, -2), keepdim=True))
        return ((sgn * Ep) * W * (v0 & vp)[..., None, None]).sum((1, 2, 3))      # per sample

    eps = 1e-6
    num = torch.zeros_like(smp)
    numw = torch.zeros_like(wts)
    with torch.no_grad():
        for k in range(n):
            for d in range(4):
                xp, xm = smp.clone(), smp.clone()
                xp[:, k, d] += eps
                xm[:, k, d] -= eps
                num[:, k, d] = (loss(xp, wts) - loss(xm, wts)) / (2 * eps)
            wp, wm = wts.clone(), wts.clone()
            wp[:, k] += eps
            wm[:, k] -= eps
            numw[:, k] = (loss(smp, wp) - loss(smp, wm)) / (2 * eps)
    rel = (g - num).abs().amax((-1, -2)) / num.abs().amax((-1, -2)).clamp(min=1e-9)
    relw = (gw - numw).abs().amax(-1) / numw.abs().amax(-1).clamp(min=1e-9)
    assert rel.median() < 1e-5 and rel.max() < 1e-3, (rel.median(), rel.max())
    assert relw.median() < 1e-5 and relw.max() < 1e-3, (relw.median(), relw.max())
    # the f32-I/O kernel on the same (rounded) inputs: the same gradient to f32 rounding of inputs and outputs
    s32 = smp.float().requires_grad_(True)
    w32 = wts.float().requires_grad_(True)
    E32, v32 = ops.solve_essential(s32, w32, "nister")
    (E32 * W.float() * v32[..., None, None]).sum().backward()
    if torch.equal(v32, valid):
        r32 = (s32.grad.double() - g).abs().amax((-1, -2)) / g.abs().amax((-1, -2)).clamp(min=1e-9)
        assert r32.median() < 1e-2, r32.median()


def test_super_rounds_equal_the_loop_on_random_configurations(dev):
    """random (solver, ransac_batch_size, max_iterations, N, pairs, plan): device rounds of several batches against one batch per
    round -- incl. batch sizes that do not divide 1024 or max_iterations, short rows (the wave-per-model scoring kernel), rows that are
    not a multiple of four points (the general sampler kernel), more than 16 sub-batches per round (a wave per sub-batch in
    dr_ransac_update) and a plan whose last round is partial"""
    import random
    from differentiable_ransac_amd import synth
    from differentiable_ransac_amd.ransac import BatchedRANSAC
    rng = random.Random(1234)
    seen_wave_per_sub = False
    for trial in range(10):
        solver = rng.choice(["nister", "stewenius", "f8", "nister"])
        B = rng.choice([8, 24, 50, 64, 100, 128, 200])
        max_it = rng.choice([300, 777, 1500, 2600])
        N = rng.choice([64, 200, 513, 1000, 2000])
        P = rng.choice([1, 3, 6])
        plan = rng.choice([None, (512, 512), (1024, 4096), (256, 1024, 2048), (4096,)])
        ratio = rng.choice([0.25, 0.4, 0.6])
        items = [synth.two_view_pair(700 + 10 * trial + p, N, inlier_ratio=ratio, pixel=solver == "f8") for p in range(P)]
        st = lambda k: torch.stack([it[k] for it in items]).to(dev)
        m, lg, K1, K2 = st("matches"), st("logits"), st("K1"), st("K2")
        kw = dict(ransac_batch_size=B, threshold=0.75, max_iterations=max_it, seed=100 + trial, refit=bool(trial % 2))
        loop = BatchedRANSAC(solver, **kw)
        loop.super_hypotheses = False
        sup = BatchedRANSAC(solver, **kw)
        sup.super_hypotheses = plan
        sup.device_termination = bool(trial % 3) and len(sup.plan()) <= 16
        seen_wave_per_sub = seen_wave_per_sub or max(sup.plan()) > 16
        a, b = loop(m, lg, K1, K2), sup(m, lg, K1, K2)
        for key in KEYS:
            assert torch.equal(a[key], b[key]), (trial, solver, B, max_it, N, P, plan, key)
    assert seen_wave_per_sub
