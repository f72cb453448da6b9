"""MSAC scoring plugin -- same interface as the reference's scorings/msac_score.py:4-55."""
import torch

from .. import ops


class MSACScore(object):
    """score(matches [N,4], models [M,3,3], threshold) -> (scores [M], masks [M,N] bool).

    Runs dr_msac_score on the GPU; `matches` may also be batched [P,N,4] with models [P,M,3,3]
    (then `threshold` may be a [P] tensor) -- the batched form is what the fused driver uses.
    """

    def __init__(self, device="cuda"):
        self.device = device
        self.provides_inliers = True

    def score(self, matches, models, threshold=0.75, want_masks=True):
        batched = matches.dim() == 3
        m = matches if batched else matches.unsqueeze(0)
        md = models if batched else models.unsqueeze(0)
        if md.dtype != m.dtype:  # the reference lets torch type-promote (Q17)
            md = md.to(m.dtype)
        scores, masks = ops.msac_score(m, md.reshape(md.shape[0], -1, 3, 3), threshold, want_masks)
        if batched:
            return scores, masks
        return scores[0], (masks[0] if masks is not None else None)
