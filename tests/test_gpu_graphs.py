"""Device-resident sampler seeds and HIP-graph replay of whole steps: the replayed step must do exactly what the eager
driver does, call after call (same seeds -> same hypotheses -> same models, masks, gradients)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_device_seed_sequence_equals_the_host_formula(dev):
    from differentiable_ransac_amd import ops
    base = 0xDEADBEEFCAFEF00D
    ds = ops.DeviceSeed(base, dev, calls=5)
    got = [int(ds.next().item()) & (2 ** 64 - 1) for _ in range(4)]
    want = [(base * 0x9E3779B97F4A7C15 + c) & (2 ** 64 - 1) for c in range(5, 9)]
    assert got == want


@pytest.mark.parametrize("N,B,k,dt", [(2000, 256, 5, torch.float32), (130, 64, 8, torch.float32), (500, 64, 3, torch.float64)])
def test_samplers_with_a_device_seed_equal_the_by_value_seed(dev, N, B, k, dt):
    from differentiable_ransac_amd import ops
    P = 3
    lg = torch.randn(P, N, device=dev, dtype=dt)
    seed = 0x1234567890ABCDEF
    st = torch.tensor([seed - 2 ** 64 if seed >= 2 ** 63 else seed], dtype=torch.int64, device=dev)
    for soft in (True, False):
        a = ops.gumbel_topk(lg, B, k, 1.0, None, seed, soft=soft)
        b = ops.gumbel_topk(lg, B, k, 1.0, None, st, soft=soft)
        assert torch.equal(a["idx"], b["idx"])
        if soft:
            assert torch.equal(a["y_sel"], b["y_sel"]) and torch.equal(a["lse"], b["lse"])
    if dt == torch.float32:
        r = ops.gumbel_topk(lg, B, k, 1.0, None, seed)
        a_sel = torch.randn(P, B, k, device=dev)
        g1 = ops.gumbel_topk_bwd(lg, None, seed, 1.0, r["idx"], r["lse"], a_sel)
        g2 = ops.gumbel_topk_bwd(lg, None, st, 1.0, r["idx"], r["lse"], a_sel)
        assert torch.allclose(g1, g2, rtol=1e-5, atol=1e-6)     # float atomics: order differs between launches
        assert torch.equal(ops.topdown_sample(lg, B, min(k, 5), seed), ops.topdown_sample(lg, B, min(k, 5), st))
    assert torch.equal(ops.uniform_sample(P, B, k, N, seed, dev), ops.uniform_sample(P, B, k, N, st, dev))
    with pytest.raises(Exception):
        ops.gumbel_topk(lg, B, k, 1.0, None, st.to(torch.int32))


@pytest.mark.parametrize("solver,sampling,P,N,B", [("nister", "gumbel", 4, 2000, 256), ("f8", "uniform", 8, 128, 64)])
def test_graph_replay_of_a_test_mode_call_equals_the_eager_driver(dev, solver, sampling, P, N, B):
    from differentiable_ransac_amd import synth
    from differentiable_ransac_amd.graphs import GraphedStep
    from differentiable_ransac_amd.ransac import BatchedRANSAC
    d = synth.batch_two_view(P, N, seed0=40, pixel=(solver == "f8"))
    m, lg, K1, K2 = (d[k].to(dev) for k in ("matches", "logits", "K1", "K2"))
    kw = dict(ransac_batch_size=B, train=False, threshold=0.75, max_iterations=B, seed=77, keep_masks=True, refit=False,
              sampling=sampling)
    eager = BatchedRANSAC(solver, **kw)
    graphed = BatchedRANSAC(solver, **kw).device_seeds(dev)
    warm = 3
    for _ in range(warm):              # GraphedStep runs fn `warm` times; the capture pass records, it does not execute
        eager(m, lg, K1, K2)
    step = GraphedStep(lambda: graphed(m, lg, K1, K2), warmup=warm)
    for r in range(4):
        want = eager(m, lg, K1, K2)
        got = step()
        for key in ("model", "mask", "score", "inliers"):
            assert torch.equal(want[key], got[key]), (key, r)
        assert torch.equal(want["masks"], got["masks"])
    # new data goes into the captured input buffers
    d2 = synth.batch_two_view(P, N, seed0=400, pixel=(solver == "f8"))
    m.copy_(d2["matches"].to(dev)); lg.copy_(d2["logits"].to(dev))
    want, got = eager(m, lg, K1, K2), step()
    assert torch.equal(want["model"], got["model"]) and torch.equal(want["mask"], got["mask"])


def test_graph_replay_of_a_train_step_equals_the_eager_step(dev):
    from differentiable_ransac_amd import synth
    from differentiable_ransac_amd.graphs import GraphedStep
    from differentiable_ransac_amd.loss import MatchLoss
    from differentiable_ransac_amd.ransac import BatchedRANSAC
    P, N, B = 4, 2000, 128
    d = synth.batch_two_view(P, N, seed0=9)
    m, gt, inl = d["matches"].to(dev), d["gt_E"].to(dev), d["inliers"].to(dev)
    ml = MatchLoss()

    def make(device_seeds):
        tr = BatchedRANSAC("nister", ransac_batch_size=B, train=True, max_iterations=B, seed=5)
        if device_seeds:
            tr.device_seeds(dev)
        lg = d["logits"].to(dev).clone().requires_grad_(True)

        def step():
            lg.grad = None
            chosen, keep = tr(m, lg, gt_model=gt)
            loss = ml(chosen, m, inl, keep)
            loss.backward()
            return loss, lg.grad
        return step

    eager, fn = make(False), make(True)
    warm = 3
    for _ in range(warm):
        eager()
    step = GraphedStep(fn, warmup=warm)
    for r in range(3):
        l0, g0 = eager()
        l1, g1 = step()
        assert torch.isfinite(g1).all() and (g1 != 0).any()
        assert torch.allclose(l0, l1, rtol=1e-5, atol=1e-7), r
        # float atomics in the sampler backward: summation order differs between launches
        assert torch.allclose(g0, g1, rtol=1e-3, atol=1e-6 * float(g0.abs().max())), r


def test_graph_replay_of_the_3d_driver_equals_the_eager_driver(dev):
    from differentiable_ransac_amd import synth
    from differentiable_ransac_amd.graphs import GraphedStep
    from differentiable_ransac_amd.ransac import BatchedRANSAC3D
    P, N, B = 2, 4096, 128                     # 4096 points: the one-pass sampler kernel, the packed residual kernel
    items = [synth.rigid_pair(p, N) for p in range(P)]
    m = torch.stack([i["matches"] for i in items]).to(dev)
    lg = torch.stack([i["logits"] for i in items]).to(dev)
    kw = dict(ransac_batch_size=B, train=False, threshold=0.03, max_iterations=B, seed=21, flag=False, keep_masks=True)
    eager = BatchedRANSAC3D(**kw)
    graphed = BatchedRANSAC3D(**kw).device_seeds(dev)
    warm = 3
    for _ in range(warm):
        eager(m, lg)
    step = GraphedStep(lambda: graphed(m, lg), warmup=warm)
    for r in range(3):
        want, got = eager(m, lg), step()
        for key in want:
            if torch.is_tensor(want[key]):
                if key == "residual":
                    # the residual sum of a model is accumulated over the point chunks with float atomics: four chunks of 1024
                    # points here (round 5: eight points per lane; two chunks of 2048 -- a commutative pair -- before), so the last
                    # bit depends on the order in which the blocks arrive
                    assert torch.allclose(want[key], got[key], rtol=1e-6, atol=0.0), (key, r)
                else:
                    assert torch.equal(want[key], got[key]), (key, r)
