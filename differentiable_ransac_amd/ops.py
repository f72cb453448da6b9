"""Functional wrappers over the C ABI (include/dransac.h) for batched [P, ...] GPU tensors, and
the torch.autograd.Function classes built on them.  Every function enqueues on the current
torch stream and returns freshly allocated torch tensors; nothing synchronises.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import _lib as L
from ._lib import c_int, c_uint64, ptr, stream


def _thr_tensor(threshold, P: int, like: torch.Tensor) -> torch.Tensor:
    if isinstance(threshold, torch.Tensor):
        t = threshold.to(device=like.device, dtype=like.dtype).reshape(-1)
        if t.numel() == 1 and P > 1:
            t = t.expand(P)
        return t.contiguous()
    return torch.full((P,), float(threshold), device=like.device, dtype=like.dtype)


# ------------------------------------------------------------------------------------------ K4 / K6
def msac_score(matches: torch.Tensor, models: torch.Tensor, threshold, want_masks: bool = True
               ) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """matches [P,N,4], models [P,M,3,3] (or [P,M,9]) -> scores [P,M], masks [P,M,N] bool | None."""
    P, N, _ = matches.shape
    M = models.shape[1]
    matches = matches.contiguous()
    models = models.contiguous()
    thr = _thr_tensor(threshold, P, matches)
    scores = torch.empty((P, M), device=matches.device, dtype=matches.dtype)
    masks = torch.empty((P, M, N), device=matches.device, dtype=torch.bool) if want_masks else None
    L.call(f"dr_msac_score_{L.suffix(matches.dtype)}", ptr(matches), ptr(models), ptr(thr), c_int(P), c_int(M),
           c_int(N), ptr(scores), ptr(masks), stream())
    return scores, masks


def select_best(matches: torch.Tensor, models: torch.Tensor, scores: torch.Tensor, threshold,
                valid: Optional[torch.Tensor] = None):
    """Per pair: (best_idx [P] int32, best_score [P], best_model [P,3,3], best_mask [P,N] bool, inliers [P] int32)."""
    P, N, _ = matches.shape
    M = models.shape[1]
    dev, dt = matches.device, matches.dtype
    thr = _thr_tensor(threshold, P, matches)
    best_idx = torch.empty((P,), device=dev, dtype=torch.int32)
    best_score = torch.empty((P,), device=dev, dtype=dt)
    best_model = torch.empty((P, 3, 3), device=dev, dtype=dt)
    best_mask = torch.empty((P, N), device=dev, dtype=torch.bool)
    inliers = torch.empty((P,), device=dev, dtype=torch.int32)
    v = None if valid is None else valid.contiguous().view(torch.uint8)
    L.call(f"dr_select_best_{L.suffix(dt)}", ptr(matches.contiguous()), ptr(models.contiguous()), ptr(v),
           ptr(scores.contiguous()), ptr(thr), c_int(P), c_int(M), c_int(N), ptr(best_idx), ptr(best_score),
           ptr(best_model), ptr(best_mask), ptr(inliers), stream())
    return best_idx, best_score, best_model, best_mask, inliers
