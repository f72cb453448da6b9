"""The two training losses that follow the hot path: MatchLoss and PoseLoss.

MatchLoss -- the clamped symmetric-epipolar training loss of the reference (loss.py:107-153), on the fused kernel
`dr_episym_fwd/bwd`.

The reference obtains the ground-truth inlier mask from `cv2.recoverPose(gt_E, pts1, pts2)` (cheirality of the
triangulated points, loss.py:99,134).  Pass either the mask (`gt_mask [P,N] bool`) or the ground-truth essential matrices
(`gt_E [P,3,3]`: the mask is then computed by `ops.recover_pose_mask`, the same Horn decomposition + cheirality vote as
the pose-error kernel); neither = all points.  Everything else follows the reference: per pair, mean over (models x
masked points) of min(error, 1); then the mean over pairs."""
import torch

from . import ops


def calibrate(models, pts1, pts2, K1, K2, im_size1, im_size2):
    """The F branch shared by the three losses (loss.py:36-49,81-92,118-121): E = K2^T F K1 and the points mapped from
    image-size-normalised to calibrated coordinates -- normalize_keypoints_tensor(denormalize_pts(pts, im_size), K)
    (cv_utils.py:35-45, feature_utils.py:40-49).  models [P,M,3,3], pts [P,N,2], K [P,3,3], im_size [P,2] = (h, w)."""
    Es = K2.transpose(-1, -2)[:, None] @ models @ K1[:, None]

    def to_camera(pts, K, im_size):
        scale = im_size.max(dim=-1).values[:, None, None]
        centre = torch.stack((im_size[:, 1] / 2, im_size[:, 0] / 2), dim=-1)[:, None, :]
        px = pts * scale + centre
        c = torch.stack((K[:, 0, 2], K[:, 1, 2]), dim=-1)[:, None, :]
        f = torch.stack((K[:, 0, 0], K[:, 1, 1]), dim=-1)[:, None, :]
        return (px - c) / f
    return Es, to_camera(pts1, K1, im_size1), to_camera(pts2, K2, im_size2)


class MatchLoss(object):
    def __init__(self, fmat=False):
        self.fmat = fmat

    def reference_forward(self, models, gt_E, pts1, pts2, K1=None, K2=None, im_size1=None, im_size2=None, topk_flag=False,
                          k=1, keep=None):
        """The reference's signature (loss.py:114): models [P,M,3,3] (F in the F branch), gt_E [P,3,3], pts [P,N,2].
        topk_flag: average only the k models with the smallest mean error per pair."""
        if self.fmat:
            models, pts1, pts2 = calibrate(models, pts1, pts2, K1, K2, im_size1, im_size2)
        matches = torch.cat((pts1, pts2), dim=-1).to(models.dtype).contiguous()
        if not topk_flag:
            return self.forward(models, matches, keep=keep, gt_E=gt_E.to(models.dtype))
        with torch.no_grad():
            gt_mask = ops.recover_pose_mask(matches, gt_E.to(models.dtype))[0][:, 0]
        sums = ops.episym_sums(matches, gt_mask, models, keep)
        per_model = sums / gt_mask.sum(1, keepdim=True).to(sums.dtype).clamp(min=1.0)
        if keep is not None:
            per_model = torch.where(keep, per_model, torch.full_like(per_model, float("inf")))
        return torch.topk(per_model, k=k, dim=1, largest=False).values.mean(1).mean()

    def forward(self, models, matches, gt_mask=None, keep=None, gt_E=None):
        """models [P,M,3,3] (E, or F already mapped to normalised coordinates), matches [P,N,4] normalised, gt_mask [P,N]
        or gt_E [P,3,3], keep [P,M] bool (models to average over; None = all) -> scalar loss."""
        if gt_mask is None and gt_E is not None:
            with torch.no_grad():
                gt_mask = ops.recover_pose_mask(matches, gt_E)[0][:, 0]
        # per pair: sum of the clamped errors / max(#masked points * #kept models, 1), then the mean over the pairs -- two
        # launches forward, one backward (ops._MatchLossMean)
        return ops.match_loss_mean(matches, gt_mask, models, keep)

    __call__ = forward


class PoseLoss(object):
    """PoseLoss (loss.py:11-68): per pair the mean over the pair's models of (err_R + err_t) / 2 in degrees, then the mean
    over pairs; `svd=False` (Horn decomposition, what train.py:82-93 passes; differentiable) or `svd=True` (decompose_E,
    forward only).  One launch (`dr_pose_error_fwd` / `dr_pose_error_svd_fwd`) instead of a Python loop over models with
    four cv2.triangulatePoints calls each."""

    def __init__(self, fmat=False):
        self.fmat = fmat

    def forward_average(self, estimated_models, pts1, pts2, gt_R, gt_t, K1=None, K2=None, im_size1=None, im_size2=None,
                        svd=False, keep=None):
        """estimated_models [P,M,3,3] (F in the F branch); pts1, pts2 [P,N,2] (calibrated coordinates, or image-size
        normalised ones + K1, K2, im sizes in the F branch); gt_R [P,3,3]; gt_t [P,3]; keep [P,M] bool (models to average
        over, e.g. the solver's validity flags; None = all) -> scalar loss."""
        # svd=True: decompose_E (cv_utils.py:83-116) instead of Horn's closed form; forward only (ops.pose_error)
        if self.fmat:
            estimated_models, pts1, pts2 = calibrate(estimated_models, pts1, pts2, K1, K2, im_size1, im_size2)
        matches = torch.cat((pts1, pts2), dim=-1).to(estimated_models.dtype).contiguous()
        err_R, err_t, _, _ = ops.pose_error(matches, estimated_models, gt_R, gt_t, svd=bool(svd))
        per_model = (err_R + err_t) / 2
        if keep is not None:
            k = keep.to(per_model.dtype)
            per_pair = (per_model * k).sum(1) / k.sum(1).clamp(min=1.0)
        else:
            per_pair = per_model.mean(1)
        return per_pair.mean()

    __call__ = forward_average


class ClassificationLoss(object):
    """ClassificationLoss (loss.py:71-104), essential-matrix branch: binary cross-entropy between the per-correspondence
    inlier probabilities and the inlier mask of `cv2.recoverPose(gt_E, pts1, pts2)` (here `ops.recover_pose_mask`)."""

    def __init__(self, fmat=False):
        self.fmat = fmat

    def forward(self, gt_E, matches, probs, K1=None, K2=None, im_size1=None, im_size2=None):
        """gt_E [P,3,3], matches [P,N,4] (calibrated, or image-size normalised + K1, K2, im sizes in the F branch), probs
        [P,N] in (0,1) -> scalar loss."""
        if self.fmat:
            _, p1, p2 = calibrate(gt_E[:, None], matches[..., :2], matches[..., 2:], K1, K2, im_size1, im_size2)
            matches = torch.cat((p1, p2), dim=-1).contiguous()
        with torch.no_grad():
            gt_mask = ops.recover_pose_mask(matches, gt_E)[0][:, 0]
        return torch.nn.functional.binary_cross_entropy(probs, gt_mask.to(probs.dtype))

    __call__ = forward
