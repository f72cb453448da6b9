"""MatchLoss -- the clamped symmetric-epipolar training loss of the reference (loss.py:107-153), on the fused kernel
`dr_episym_fwd/bwd`.

The reference obtains the ground-truth inlier mask from `cv2.recoverPose` (cheirality of the triangulated points);
OpenCV is outside this package, so the mask is an INPUT here (`gt_mask [P,N] bool`; None = all points).  Everything
else follows the reference: per pair, mean over (models x masked points) of min(error, 1); then the mean over pairs."""
import torch

from . import ops


class MatchLoss(object):
    def __init__(self, fmat=False):
        self.fmat = fmat

    def forward(self, models, matches, gt_mask=None, keep=None):
        """models [P,M,3,3] (E, or F already mapped to normalised coordinates), matches [P,N,4] normalised, gt_mask [P,N],
        keep [P,M] bool (models to average over; None = all) -> scalar loss."""
        sums = ops.episym_sums(matches, gt_mask, models, keep)
        P, N, _ = matches.shape
        n_in = gt_mask.sum(1).to(sums.dtype) if gt_mask is not None else torch.full((P,), float(N), device=sums.device)
        n_models = keep.sum(1).to(sums.dtype) if keep is not None else torch.full((P,), float(sums.shape[1]), device=sums.device)
        per_pair = sums.sum(1) / (n_in * n_models).clamp(min=1.0)
        return per_pair.mean()

    __call__ = forward
