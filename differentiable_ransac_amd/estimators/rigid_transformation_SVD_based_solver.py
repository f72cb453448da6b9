"""Rigid-transformation solver plugin -- interface of
estimators/rigid_transformation_SVD_based_solver.py:4-89 of the reference."""
import torch

from .. import ops


class RigidTransformationSVDBasedSolver:
    def __init__(self, data_type=torch.float32, device='cuda'):
        self.data_type = data_type
        self.device = device
        self.sample_size = 3

    def estimate_model(self, data, weights=None, sample_indices=None, flag=True):
        """data [B,n>=3,6] -> (model [B,4,4], R [B,3,3], t [B,3], scale [B]).  flag=True reproduces the
        reference default (SVD of cov^T cov => R ~ I, Q9); flag=False is the usual Kabsch solution, with R
        returned in the reference's row-vector convention (= R_true^T)."""
        assert data.shape[-1] == 6 and data.shape[-2] >= 3
        if sample_indices is not None:
            data = torch.index_select(data, 0, sample_indices)
        model, R, t, scale, valid = ops.solve_rigid_autograd(data, weights, flag)
        self.last_valid = valid
        return model, R, t, scale

    def squared_residual(self, pts1, pts2, descriptor, threshold=0.03):
        """pts1, pts2 [N,3]; descriptor [B,4,3] = model[:, :3, :]^T -> (sum_n d2 [B], mean d2, mask [B,N])."""
        assert pts1.shape[1] == 3
        pts = torch.cat((pts1, pts2), dim=1).unsqueeze(0)
        B = descriptor.shape[0]
        model = torch.zeros(B, 4, 4, device=descriptor.device, dtype=descriptor.dtype)
        model[:, :3, :] = descriptor.transpose(-1, -2)
        model[:, 3, 3] = 1
        res, mask = ops.rigid_residual_autograd(pts.to(descriptor.dtype), model.unsqueeze(0), threshold)
        return res[0], res[0].sum() / (B * pts1.shape[0]), mask[0]
