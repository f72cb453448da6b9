"""Fundamental-matrix estimator plugin -- interface of FundamentalMatrixEstimatorNew,
estimators/fundamental_matrix_estimator.py:161-308 of the reference."""
import torch

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

    def normalize(self, matches):
        """fundamental_matrix_estimator.py:177-228: Hartley normalisation per sample, matches [B,n,4] ->
        (normalised [B,n,4], T1 [B,3,3], T2t [B,3,3]) with x1' = T1 x1 and T2t the TRANSPOSE of the second image's transform
        (estimate_non_minimal_model forms T2t F T1).  estimate_model does not come through here -- dr_solve_f8 normalises in
        registers; this is the reference's public helper, a handful of device ops on the caller's tensors."""
        import math
        mass = matches.mean(dim=1, keepdim=True)
        c = matches - mass
        r1 = math.sqrt(2) / c[..., :2].norm(dim=2).mean(dim=1)
        r2 = math.sqrt(2) / c[..., 2:].norm(dim=2).mean(dim=1)
        ratio = torch.stack((r1, r1, r2, r2), dim=-1)
        z, o = torch.zeros_like(r1), torch.ones_like(r1)
        m = mass[:, 0]
        T1 = torch.stack((r1, z, -r1 * m[:, 0], z, r1, -r1 * m[:, 1], z, z, o), dim=-1).view(-1, 3, 3)
        T2t = torch.stack((r2, z, z, z, r2, z, -r2 * m[:, 2], -r2 * m[:, 3], o), dim=-1).view(-1, 3, 3)
        return c * ratio[:, None, :], T1, T2t

    def estimate_non_minimal_model(self, pts, T1, T2t, weights=None):
        """fundamental_matrix_estimator.py:230-260: (weighted) LSQ eight-point solve on ALREADY normalised points, then T2t F T1
        (T1 = None: the model of the normalised points).  The solve is dr_solve_f8 -- whose own Hartley step is the identity up
        to rounding on normalised input."""
        F = ops.solve_fundamental8(pts, weights)[0]
        if T1 is None:
            return F
        return T2t @ F @ T1

    def estimate_minimal_model(self, pts, weights=None):
        return ops.solve_f7(pts)[0].reshape(-1, 3, 3)

    def estimate_model_slots(self, matches, weights=None):
        if matches.shape[1] == 7:
            return ops.solve_f7(matches)
        F, valid = ops.solve_fundamental8(matches, weights)
        return F.unsqueeze(1), valid.unsqueeze(1)


# the reference also exports the older class name; both resolve to the working implementation
FundamentalMatrixEstimator = FundamentalMatrixEstimatorNew
