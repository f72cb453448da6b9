"""Stewenius five-point estimator plugin -- interface of
estimators/essential_matrix_estimator_stewenius.py:5-18 of the reference."""
from .. import ops


class EssentialMatrixEstimator(object):
    """estimate_model(matches [B,5,4]) -> E [10*B,3,3]; `weights` is accepted and ignored like in the
    reference (stewenius.py:38-42 builds unweighted rows).  Solutions have unit Frobenius norm (the
    reference leaves LAPACK's eigenvector scale; MSAC is scale invariant); real solutions first, ascending
    action-matrix eigenvalue, unused slots eye(3)."""

    def __init__(self, device='cuda'):
        self.sample_size = 5
        self.device = device  # the reference forgets to set it (Q6)

    def estimate_model(self, matches, weights=None):
        if matches.shape[1] < self.sample_size:
            return None
        models, _ = self.estimate_model_slots(matches[:, :5].contiguous())
        return models.reshape(-1, 3, 3)

    def estimate_minimal_model(self, pts, weights=None):
        return self.estimate_model(pts, weights)

    def estimate_model_slots(self, matches, weights=None):
        return ops.solve_essential(matches, None, "stewenius")
