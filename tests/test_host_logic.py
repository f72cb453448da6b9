"""Host-side logic that needs no GPU: driver configuration, plugin detection, sharding arithmetic."""
import pytest
import torch


def _plugins(fmat=False, k=5, uniform=False):
    from differentiable_ransac_amd.estimators import (EssentialMatrixEstimator, EssentialMatrixEstimatorNister,
                                                      FundamentalMatrixEstimatorNew)
    from differentiable_ransac_amd.samplers import GumbelSoftmaxSampler, UniformSampler
    from differentiable_ransac_amd.scorings import MSACScore
    est = FundamentalMatrixEstimatorNew("cuda") if fmat else EssentialMatrixEstimatorNister("cuda")
    smp = UniformSampler(64, k) if uniform else GumbelSoftmaxSampler(64, k, device="cuda")
    return est, smp, MSACScore("cuda"), EssentialMatrixEstimator


def test_dropin_ransac_picks_the_fused_driver_only_for_its_own_plugins():
    from differentiable_ransac_amd.ransac import RANSAC
    est, smp, sc, Stew = _plugins()
    assert RANSAC(est, smp, sc, sampler_id=2)._fused_solver() == "nister"
    assert RANSAC(est, smp, sc, sampler_id=2, train=True)._fused_solver() is None          # train mode: plugin path
    assert RANSAC(Stew("cuda"), smp, sc, sampler_id=2)._fused_solver() == "stewenius"
    estF, smp8, scF, _ = _plugins(fmat=True, k=8)
    assert RANSAC(estF, smp8, scF, fmat=True, sampler_id=3)._fused_solver() == "f8"
    assert RANSAC(estF, smp8, scF, fmat=False, sampler_id=3)._fused_solver() is None       # inconsistent flags
    _, smpu, _, _ = _plugins(uniform=True)
    assert RANSAC(est, smpu, sc, sampler_id=0)._fused_solver() is None                      # uniform sampler

    class MyScore(type(sc)):                                                                # a user's subclass is custom
        pass
    assert RANSAC(est, smp, MyScore("cuda"), sampler_id=2)._fused_solver() is None
    with pytest.raises(NotImplementedError):
        RANSAC(est, smp, sc, lo=3)


def test_batched_ransac_configuration_errors():
    from differentiable_ransac_amd.ransac import BatchedRANSAC
    with pytest.raises(ValueError):
        BatchedRANSAC("nister", sampling="sobol")
    with pytest.raises(ValueError):
        BatchedRANSAC("nister", train=True, sampling="topdown")
    with pytest.raises(ValueError):
        BatchedRANSAC("nister", weighted=1, sampling="topdown")
    with pytest.raises(KeyError):
        BatchedRANSAC("dlt")
    rn = BatchedRANSAC("f8", ransac_batch_size=64)
    assert (rn.k, rn.S, rn.fmat, rn.sync_every) == (8, 1, True, None)
    # round 6: device rounds of 1024 hypotheses when the batch is smaller (ceil(5000 / 64) = 79 batches), one batch per round otherwise
    assert rn.plan() == [16, 16, 16, 16, 15] and sum(rn.plan()) == 79
    rn.super_hypotheses = (1024, 4096)
    assert rn.plan() == [16, 63]
    rn.super_hypotheses = False
    assert rn.plan() == [1] * 79
    assert BatchedRANSAC("nister", ransac_batch_size=1024).plan() == [1] * 5
    assert BatchedRANSAC("f8", ransac_batch_size=64, weighted=1).plan() == [1] * 79       # weighted refit: batch by batch
    assert BatchedRANSAC("nister", ransac_batch_size=64, weighted=1).plan() == [1] * 79   # weighted minimal solves need every batch's soft weights
    assert BatchedRANSAC("nister", ransac_batch_size=1, max_iterations=5000).plan()[0] == 512   # dr_ransac_update's cap
    assert BatchedRANSAC("nister", ransac_batch_size=1024).sync_every is None      # = max(1, 256 // hypotheses per device round)


def test_ops_refuse_cpu_tensors_without_touching_the_gpu():
    from differentiable_ransac_amd import ops
    from differentiable_ransac_amd._lib import DransacError
    with pytest.raises(DransacError):
        ops.msac_score(torch.rand(1, 16, 4), torch.rand(1, 3, 3, 3), 1e-3)
    with pytest.raises(DransacError):
        ops.topdown_sample(torch.rand(1, 16), 4, 3)


def test_pose_loss_has_no_cpu_path_and_calibrate_matches_the_reference_formulas():
    from differentiable_ransac_amd import _lib
    from differentiable_ransac_amd.loss import PoseLoss, calibrate
    for svd in (False, True):      # both decompositions run on the GPU only (no CPU fallback anywhere in the product)
        with pytest.raises(_lib.DransacError):
            PoseLoss().forward_average(torch.zeros(1, 1, 3, 3), torch.zeros(1, 4, 2), torch.zeros(1, 4, 2), torch.eye(3)[None],
                                       torch.ones(1, 3), svd=svd)
    # E = K2^T F K1 ; points: pts * max(im_size) + (w/2, h/2), then (p - c) / f
    g = torch.Generator().manual_seed(0)
    F = torch.randn(2, 3, 3, 3, generator=g)
    K1 = torch.tensor([[[500.0, 0, 320], [0, 510, 240], [0, 0, 1]]]).repeat(2, 1, 1)
    K2 = torch.tensor([[[700.0, 0, 400], [0, 690, 300], [0, 0, 1]]]).repeat(2, 1, 1)
    im1, im2 = torch.tensor([[480.0, 640.0]] * 2), torch.tensor([[600.0, 800.0]] * 2)
    p1, p2 = torch.rand(2, 5, 2, generator=g) - 0.5, torch.rand(2, 5, 2, generator=g) - 0.5
    Es, q1, q2 = calibrate(F, p1, p2, K1, K2, im1, im2)
    for b in range(2):
        assert torch.allclose(Es[b], K2[b].T @ F[b] @ K1[b], atol=1e-3)
        px = p1[b] * 640.0 + torch.tensor([320.0, 240.0])
        assert torch.allclose(q1[b], (px - torch.tensor([320.0, 240.0])) / torch.tensor([500.0, 510.0]), atol=1e-6)
        px2 = p2[b] * 800.0 + torch.tensor([400.0, 300.0])
        assert torch.allclose(q2[b], (px2 - torch.tensor([400.0, 300.0])) / torch.tensor([700.0, 690.0]), atol=1e-6)


def test_ransac_layers_wire_the_plugins_like_the_reference():
    """model_cl.py:160-232, 516-575: solver / sampler / iteration budget chosen from the option namespace."""
    import types
    from differentiable_ransac_amd.layers import RANSACLayer, RANSACLayer3D, batched_forward, denormalize_pts
    from differentiable_ransac_amd.estimators import EssentialMatrixEstimatorNister, FundamentalMatrixEstimatorNew
    from differentiable_ransac_amd.samplers import GumbelSoftmaxSampler, UniformSampler
    opt = types.SimpleNamespace(fmat=False, sampler=2, ransac_batch_size=64, tr=True, weighted=0, threshold=0.75, precision=1,
                                device="cuda")
    l = RANSACLayer(opt)
    assert isinstance(l.estimator.estimator, EssentialMatrixEstimatorNister) and l.estimator.max_iterations == 100
    assert isinstance(l.estimator.sampler, GumbelSoftmaxSampler) and l.estimator.sampler.num_samples == 5
    opt.tr = False
    assert RANSACLayer(opt).estimator.max_iterations == 5000
    opt.fmat, opt.sampler, opt.tr = True, 3, True
    l = RANSACLayer(opt)
    assert isinstance(l.estimator.estimator, FundamentalMatrixEstimatorNew) and l.estimator.max_iterations == 1000
    assert l.estimator.sampler.num_samples == 8
    opt.sampler = 0
    assert isinstance(RANSACLayer(opt).estimator.sampler, UniformSampler)
    with pytest.raises(NotImplementedError):
        batched_forward(opt, torch.zeros(1, 8, 4), torch.zeros(1, 8), None, None)
    opt.sampler = 2
    assert RANSACLayer3D(opt).estimator.max_iterations == 1000
    opt.precision = 0
    with pytest.raises(NotImplementedError):
        RANSACLayer(opt)
    # cv_utils.denormalize_pts: pts * max(im_size) + (w/2, h/2), im_size = (h, w)
    out = denormalize_pts(torch.tensor([[0.0, 0.0], [0.5, -0.25]]), torch.tensor([480.0, 640.0]))
    assert torch.equal(out, torch.tensor([[320.0, 240.0], [640.0, 80.0]]))


def test_philox_restatement_known_answers():
    """tests/philox_ref.py (the numpy restatement the in-kernel stream is pinned by) against the Random123 known-answer
    vectors of philox4x32-10; the kernels run the same round function seven times (philox_ref.ROUNDS)."""
    import numpy as np
    from tests import philox_ref as R
    kat = [(0, (0, 0, 0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           (0xffffffffffffffff, (0xffffffff,) * 4, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x299f31d0 << 32) | 0xa4093822, (0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344),
            (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for key, ctr, out in kat:
        got = R.philox4x32_10(key, *[np.uint32(c) for c in ctr])
        assert tuple(int(x) for x in got) == out
    assert R.ROUNDS == 7
    a = R.philox4x32(5, np.arange(4, dtype=np.uint32), 1, 2, 0)
    b = R.philox4x32(5, np.arange(4, dtype=np.uint32), 1, 2, 0, rounds=10)
    assert not np.array_equal(a[0], b[0])
    # words are uniform enough for a smoke check: mean of 2^16 draws of the top 24 bits
    w = R.philox4x32(77, np.arange(1 << 14, dtype=np.uint32), 3, 1, 0)
    u = np.concatenate([(x >> np.uint32(8)).astype(np.float64) for x in w]) / 2.0 ** 24
    assert abs(u.mean() - 0.5) < 0.005 and abs(u.var() - 1 / 12) < 0.002


def test_async_gradient_bucket_without_a_process_group_is_a_no_op_pipeline():
    """single process (N = 1): the bucket does no collective but keeps the same program order -- step i + 1 is enqueued before
    the wait on bucket i, buffers alternate, drain() empties the pipeline"""
    import torch
    from differentiable_ransac_amd import sharding
    b = sharding.AsyncGradientBucket(8, "cpu", None)
    seen = []

    def step():
        b.bucket().fill_(float(len(seen) + 1))
        seen.append(len(seen))
    s = sharding.OverlappedStep(step, b)
    for _ in range(3):
        s()
    b.drain()
    tr = b.trace
    assert tr == [("enqueued", 0), ("launch", 0), ("enqueued", 1), ("wait", 0), ("launch", 1), ("enqueued", 2), ("wait", 1),
                  ("launch", 2), ("wait", 2)]
    assert float(b.buf[0][0]) == 3.0 and float(b.buf[1][0]) == 2.0 and b.exposed_events == []
    assert b.wait() is None


def test_async_gradient_bucket_never_hands_out_a_live_buffer():
    """round-5 advice: with two buckets in flight, bucket() returns the buffer of the OLDEST in-flight all-reduce -- it must be
    waited for before the backward writes into it, its averaged contents must reach the hook, and a launch() that bypassed
    bucket() must raise instead of reducing an overwritten buffer"""
    import pytest
    import torch
    from differentiable_ransac_amd import sharding
    b = sharding.AsyncGradientBucket(4, "cpu", None)
    got = []
    b.on_overrun = lambda t, i: got.append((i, float(t[0])))
    b.bucket().fill_(1.0)
    b.launch()
    b.bucket().fill_(2.0)
    b.launch()                    # both buffers in flight
    nxt = b.bucket()              # waits for bucket 0 first, hands its result to the hook, THEN returns the buffer
    assert got == [(0, 1.0)] and b.waited == 1 and float(b.overrun_reduced[0]) == 1.0
    nxt.fill_(3.0)
    assert float(b.overrun_reduced[0]) == 1.0      # a copy: refilling the buffer does not touch the handed-out result
    b.launch()
    assert b.trace == [("launch", 0), ("launch", 1), ("wait", 0), ("launch", 2)]
    with pytest.raises(RuntimeError):
        b.launch()                # a third in-flight bucket without going through bucket()
    b.drain()
    assert b.waited == b.issued == 3


def test_f64_match_loss_selects_dropped_slots_out():
    """round-5 advice: NaN models in keep=False slots (invalid f8 / LSQ hypotheses) must not poison the f64 MatchLoss or its
    gradient -- the f32 kernels skip such slots; the torch-op f64 path has to agree on exactly these inputs"""
    import torch
    from differentiable_ransac_amd import ops
    torch.manual_seed(0)
    P, N, M = 2, 50, 6
    m = torch.rand(P, N, 4, dtype=torch.float64)
    md = torch.randn(P, M, 3, 3, dtype=torch.float64)
    keep = torch.rand(P, M) > 0.3
    mask = torch.rand(P, N) > 0.2
    bad = md.clone()
    bad[~keep] = float("nan")
    md.requires_grad_(True)
    bad.requires_grad_(True)
    l1 = ops._match_loss_mean_f64(m, mask, md, keep)
    l2 = ops._match_loss_mean_f64(m, mask, bad, keep)
    l1.backward()
    l2.backward()
    assert torch.isfinite(l2) and float(l1.detach()) == float(l2.detach())
    assert torch.isfinite(bad.grad).all() and torch.equal(md.grad[keep], bad.grad[keep])
    assert float(bad.grad[~keep].abs().max()) == 0.0
