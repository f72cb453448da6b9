"""Nister five-point estimator plugin -- interface of
estimators/essential_matrix_estimator_nister.py:30-67 of the reference."""
import torch

from .. import ops


class EssentialMatrixEstimatorNister(object):
    """estimate_model(matches [B,n,4], weights [B,n] | None, **kw) -> E [10*B,3,3].

    Differences from the reference, all documented in DESIGN.md:
      * only REAL roots produce solutions (the reference keeps Re(z) of complex roots, Q10); a sample's
        real solutions come first (ascending root), the remaining slots are eye(3);
      * numerically failed samples are kept as 10 x eye(3) instead of being dropped, so the output
        always has 10*B rows and stays aligned with the sample index;
      * n > 5 runs the same solver on all points (nister.py:64-65; pymagsac is never used).
    `estimate_model_slots` returns the fixed-shape ([B,10,3,3], valid [B,10]) pair the batched driver uses.
    """

    def __init__(self, device='cuda'):
        self.sample_size = 5
        self.device = device

    def estimate_model(self, matches, weights=None, K1=None, K2=None, inlier_indices=None, best_model=None,
                       unnormalzied_threshold=None, best_score=0):
        if matches.shape[1] < self.sample_size:
            return None
        models, _ = self.estimate_model_slots(matches, weights)
        return models.reshape(-1, 3, 3)

    def estimate_minimal_model(self, pts, weights=None):
        return self.estimate_model(pts, weights)

    def estimate_model_slots(self, matches, weights=None):
        return ops.solve_essential(matches, weights, "nister")
