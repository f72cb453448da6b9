"""tests/golden/clnet_logits.npz -- reader-produced pairs scored by the reference's network (generated in the build
container by tests/golden/gen_clnet_logits.py): the fixture is self-consistent and usable as the sampler's input."""
import numpy as np
import torch

from oracle import cpu_ref as O
from tests.conftest import GOLDEN


def test_fixture_shapes_and_separation():
    z = np.load(GOLDEN + "/clnet_logits.npz")
    P, N = z["matches"].shape[:2]
    assert z["matches"].shape == (P, N, 4) and z["log_probs"].shape == (P, N) and z["side"].shape == (P, N, 3)
    assert np.allclose(np.exp(z["log_probs"]), z["weights"], rtol=1e-5, atol=1e-7)                       # -p 2 vs -p 1
    assert np.allclose(z["weights"] / z["weights"].sum(-1, keepdims=True), z["weights_normalized"], rtol=1e-4, atol=1e-9)
    inl = z["geometric_inliers"]
    assert 0.3 < inl.mean() < 0.7
    for p in range(P):   # the network (trained on real SIFT matches) separates the geometric inliers of these pairs
        assert z["log_probs"][p][inl[p]].mean() - z["log_probs"][p][~inl[p]].mean() > 3.0


def test_oracle_ransac_with_network_scores_finds_the_geometry():
    z = np.load(GOLDEN + "/clnet_logits.npz")
    m, lg = torch.from_numpy(z["matches"][0]).double(), torch.from_numpy(z["log_probs"][0]).double()
    K1, K2 = torch.from_numpy(z["K1"][0]).double(), torch.from_numpy(z["K2"][0]).double()
    g = torch.Generator().manual_seed(0)
    noise = [-torch.log(-torch.log(torch.rand(32, m.shape[0], generator=g, dtype=torch.float64).clamp(1e-12, 1 - 1e-12)))
             for _ in range(4)]
    model, mask, score, iters = O.ransac_test(m, lg, noise, K1, K2, "nister", max_iterations=128, refit=False)
    inl = torch.from_numpy(z["geometric_inliers"][0])
    # the pairs carry 1 px of noise against a 1.125 px threshold, so the inlier set of the ESTIMATED model and that of the
    # ground truth overlap, not coincide: 848 / 958 points, 759 in common on this fixture
    assert int(mask.sum()) > 0.8 * int(inl.sum())
    assert float((mask & inl).sum()) / max(1, int(mask.sum())) > 0.85
