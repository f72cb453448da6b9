"""The RANSAC layers of the reference (the callers of the hot path), without the CLNet that feeds them:

  RANSACLayer      model_cl.py:160-256   plugin wiring from the option namespace + forward for one image pair
  RANSACLayer3D    model_cl.py:516-595   the same for 3-D point registration
  batched_forward  model_cl.py:488-511   the per-pair Python loop of DeepRansac_CLNet.forward as ONE BatchedRANSAC call

`opt` is the reference's argparse namespace (utils.py:30-77); only the fields the reference reads here are used:
fmat, sampler, ransac_batch_size, tr, weighted, threshold, precision, device.  Returns are the reference's:
(models with NaN rows dropped, wall time of the RANSAC call in seconds).
"""
import time

import torch

from .estimators import EssentialMatrixEstimatorNister, FundamentalMatrixEstimatorNew, RigidTransformationSVDBasedSolver
from .ransac import RANSAC, RANSAC3D, BatchedRANSAC
from .samplers import GumbelSoftmaxSampler, UniformSampler
from .scorings import MSACScore


def denormalize_pts(pts: torch.Tensor, im_size: torch.Tensor) -> torch.Tensor:
    """cv_utils.denormalize_pts (:35-45): undo the image-size normalisation of the F branch; im_size = (height, width)."""
    return pts * max(im_size) + torch.stack((im_size[1] / 2, im_size[0] / 2)).to(pts)


def _data_type(opt):
    if opt.precision == 0:
        raise NotImplementedError("half precision is not supported (it never worked upstream either: SURVEY Q15)")
    return torch.float64 if opt.precision == 2 else torch.float32


def _sampler(opt, sample_size, data_type):
    # model_cl.py:178-207: 0 uniform; 1, 2 Gumbel with the solver's sample size; anything else Gumbel with 8 points
    if opt.sampler == 0:
        return UniformSampler(opt.ransac_batch_size, sample_size)
    k = sample_size if opt.sampler in (1, 2) else 8
    return GumbelSoftmaxSampler(opt.ransac_batch_size, k, device=opt.device, data_type=data_type)


def _drop_nan(models: torch.Tensor) -> torch.Tensor:
    if models.dim() == 2:                      # test mode returns one [3,3] model
        return models
    keep = ~torch.isnan(models).flatten(1).any(1)
    return models[keep]


class RANSACLayer(torch.nn.Module):
    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        dt = _data_type(opt)
        solver = (FundamentalMatrixEstimatorNew(opt.device, opt.weighted) if opt.fmat
                  else EssentialMatrixEstimatorNister(opt.device))
        sampler = _sampler(opt, solver.sample_size, dt)
        max_iters = (1000 if opt.tr else 5000) if opt.fmat else (100 if opt.tr else 5000)      # model_cl.py:213-219
        self.estimator = RANSAC(solver, sampler, MSACScore(opt.device), max_iterations=max_iters, fmat=opt.fmat,
                                train=opt.tr, ransac_batch_size=opt.ransac_batch_size, sampler_id=opt.sampler,
                                weighted=opt.weighted, threshold=opt.threshold)
        # The second return value of forward() is a wall time upstream (test.py:100 averages it into "Run time").  The replayed
        # test-mode call returns before the device has finished (nothing is read back), so by default it is the ENQUEUE time of the
        # call; "sync" waits for the pair's result first -- the reference's meaning, at the price of the host running ahead.
        self.timing = "enqueue"
        # The reference drops NaN models from what the layer returns (model_cl.py:240-242).  This package's estimators never return
        # one -- a failed hypothesis comes back as eye(3) with valid = 0 and is dropped by the driver -- so for them the filter is the
        # identity, and running it costs three launches and a host synchronisation (the boolean-mask index) per pair: skipped.
        # Third-party estimators keep it.
        self._finite_models = hasattr(solver, "estimate_model_slots")

    def forward(self, points, weights, K1, K2, im_size1, im_size2, ground_truth=None, gumbels=None):
        """points [N,4], weights (logits) [N] -> (Es, seconds).  Train: Es [n_batches * B', 3, 3] with autograd to
        `weights`; test: Es [3,3]."""
        # (model_cl.py:239 clones because the F branch overwrites the coordinates; nothing below writes to the E branch's input --
        #  the drivers never mutate theirs -- and the copy was one 5 us launch of the 0.14 ms a test-mode pair costs)
        points_ = points.clone() if self.opt.fmat else points
        if self.opt.fmat:
            points_[:, 0:2] = denormalize_pts(points[:, 0:2], im_size1)
            points_[:, 2:4] = denormalize_pts(points[:, 2:4], im_size2)
        t0 = time.time()
        models, _, _, _ = self.estimator(points_, weights, K1, K2, ground_truth, gumbels=gumbels)
        if self.timing == "sync" and points_.is_cuda:
            torch.cuda.current_stream(points_.device).synchronize()
        dt = time.time() - t0
        if self.opt.tr:
            vals = list(models.values())
            Es = vals[0] if len(vals) == 1 else torch.cat(vals)      # (one batch, the reference's training setting: nothing to copy)
        else:
            Es = models
        own = self._finite_models and hasattr(self.estimator.estimator, "estimate_model_slots")   # (the plugin may have been swapped)
        return (Es if own else _drop_nan(Es)), dt


class RANSACLayer3D(torch.nn.Module):
    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        dt = _data_type(opt)
        solver = RigidTransformationSVDBasedSolver()
        sampler = _sampler(opt, solver.sample_size, dt)
        self.estimator = RANSAC3D(solver, sampler, MSACScore(opt.device), max_iterations=1000, fmat=opt.fmat, train=opt.tr,
                                  ransac_batch_size=opt.ransac_batch_size, sampler_id=opt.sampler, weighted=opt.weighted,
                                  threshold=opt.threshold)

    def forward(self, points, weights, ground_truth=None, gumbels=None):
        """points [N,6], weights [N] -> (T [n,4,4], mean residual sum, mean of the mean residuals, seconds) in train mode
        (model_cl.py:579-595; the reference's test branch reads undefined variables, SURVEY Q4: here (T [4,4], time))."""
        t0 = time.time()
        models, residuals, avg_residuals, _, _ = self.estimator(points, weights, ground_truth, gumbels=gumbels)
        dt = time.time() - t0
        if not self.opt.tr:
            return models, dt
        Ts = torch.cat(list(models.values()))
        loss = torch.cat(list(residuals.values()))
        avg_loss = sum(avg_residuals.values()) / len(avg_residuals)
        keep = ~torch.isnan(Ts).flatten(1).any(1)
        return Ts[keep], loss.mean(), avg_loss, dt


def batched_forward(opt, points, weights, K1, K2, im_size1=None, im_size2=None, gt=None, driver=None):
    """DeepRansac_CLNet.forward's loop over the pairs of a batch (model_cl.py:488-511) as one BatchedRANSAC call.
    points [P,N,4], weights [P,N], K1/K2 [P,3,3], im sizes [P,2] (F branch), gt [P,3,3] (train) ->
    (list of per-pair model tensors like the reference's `ret`, seconds per pair).  Pass `driver` to reuse one
    BatchedRANSAC across calls."""
    P = points.shape[0]
    if opt.sampler == 0:
        raise NotImplementedError("the batched driver samples with the Gumbel sampler (sampler ids 1-3); "
                                  "use RANSACLayer for the uniform sampler")
    if driver is None:
        solver = ("f8" if opt.sampler not in (1, 2) else "f7") if opt.fmat else "nister"
        max_iters = (1000 if opt.tr else 5000) if opt.fmat else (100 if opt.tr else 5000)
        driver = BatchedRANSAC(solver, ransac_batch_size=opt.ransac_batch_size, train=bool(opt.tr), threshold=opt.threshold,
                               max_iterations=max_iters, weighted=opt.weighted)
    pts = points
    if opt.fmat:
        # cv_utils.denormalize_pts (:35-45) for all pairs at once: pts * max(im_size) + (width / 2, height / 2), the maximum taken
        # ON THE DEVICE (round 5 looped over the pairs with Python's max() on device tensors: one synchronisation per pair and image)
        def denorm(xy, im):
            im = im.to(xy)
            return xy * im.max(-1).values[:, None, None] + torch.stack((im[:, 1] / 2, im[:, 0] / 2), -1)[:, None, :]
        pts = torch.cat((denorm(points[..., 0:2], im_size1), denorm(points[..., 2:4], im_size2)), -1)
    t0 = time.time()
    if opt.tr:
        chosen, keep = driver(pts, weights, K1, K2, gt_model=gt)
        # The reference's `ret` is ragged (per pair the models that survived, model_cl.py:240-242).  ONE read-back for the whole
        # batch -- the per-pair counts -- instead of a boolean-mask gather (= a synchronisation) per pair: a stable sort moves every
        # pair's kept models to the front in their original order, the per-pair results are views of that tensor.
        counts = keep.sum(1)
        order = torch.argsort((~keep).to(torch.uint8), dim=1, stable=True)
        packed = torch.gather(chosen, 1, order[:, :, None, None].expand(-1, -1, 3, 3))
        ret = [packed[p, :n] for p, n in enumerate(counts.tolist())]
    else:
        out = driver(pts, weights, K1, K2)
        ret = list(out["model"].unbind(0))
    return ret, (time.time() - t0) / P
