"""Fundamental-matrix estimator plugin -- interface of FundamentalMatrixEstimatorNew,
estimators/fundamental_matrix_estimator.py:161-308 of the reference."""
from .. import ops


class FundamentalMatrixEstimatorNew(object):
    """estimate_model(matches [B,n,4], weights [B,n] | None):
         n == 7 -> 7-point, F [4*B,3,3] (unit norm, invalid slots eye(3); the CORRECT maths, Q7/Q8)
         n  > 7 -> Hartley-normalised 8-point / LSQ, F [B,3,3] (un-normalised, no rank-2 projection)."""

    def __init__(self, device='cuda', weighted=0):
        self.sample_size = 7
        self.device = device
        self.weighted = weighted
        self.eps = 1e-8

    def estimate_model(self, matches, weights=None):
        if matches.shape[1] == self.sample_size:
            return self.estimate_minimal_model(matches, weights)
        elif matches.shape[1] > self.sample_size:
            return ops.solve_fundamental8(matches, weights)[0]
        return None

    def estimate_minimal_model(self, pts, weights=None):
        return ops.solve_f7(pts)[0].reshape(-1, 3, 3)

    def estimate_model_slots(self, matches, weights=None):
        if matches.shape[1] == 7:
            return ops.solve_f7(matches)
        F, valid = ops.solve_fundamental8(matches, weights)
        return F.unsqueeze(1), valid.unsqueeze(1)


# the reference also exports the older class name; both resolve to the working implementation
FundamentalMatrixEstimator = FundamentalMatrixEstimatorNew
