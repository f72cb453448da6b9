"""RANSAC drivers -- same classes, constructor arguments and return values as the reference's
ransac.py (RANSAC :6-299, RANSAC3D :303-450), plus BatchedRANSAC, which runs the same algorithm
over a whole (image-pair x hypothesis) grid in a handful of launches instead of the reference's
per-pair Python loop (model_cl.py:488).

Everything numeric happens in libdransac.so through differentiable_ransac_amd.ops; these classes
only sequence launches and keep the (tiny) per-pair state.
"""
from __future__ import annotations

import collections
import math
from typing import Dict

import torch

from . import ops


import os as _os
_FOLD_SETUP = _os.environ.get("DRANSAC_FOLD_SETUP", "1") != "0"   # A/B: 0 = seed launch and result gather as separate nodes (round 5)


def adaptive_iteration_number(inlier_number, point_number, sample_size, confidence=0.999, eps=1e-5,
                              max_iterations=5000):
    """ransac.py:202-215."""
    ratio = float(inlier_number) / float(point_number)
    prob = 1.0 - ratio ** sample_size
    if prob >= 1.0 - eps:
        return max_iterations
    return max(0.0, math.log10(1.0 - confidence) / math.log10(1 - ratio ** sample_size + eps))


def normalized_threshold(threshold, K1, K2, fmat):
    """ransac.py:49-53 -- (K1[0,0] + K1[1,1] + K1[0,0] + K2[1,1]) / 4, K1[0,0] twice on purpose (Q3)."""
    if fmat:
        return threshold
    return threshold / ((K1[..., 0, 0] + K1[..., 1, 1] + K1[..., 0, 0] + K2[..., 1, 1]) / 4)


def _takes_argument(fn) -> bool:
    """True when `fn` (a bound method) accepts one positional argument."""
    import inspect
    try:
        params = [p for p in inspect.signature(fn).parameters.values()
                  if p.kind in (p.POSITIONAL_ONLY, p.POSITIONAL_OR_KEYWORD, p.VAR_POSITIONAL)]
    except (TypeError, ValueError):
        return False
    return len(params) >= 1


def _is_gumbel(sampler_id):
    # ids 2 and 3 in the reference; id 1 builds a Gumbel sampler there too but then crashes (Q16): treated as 2
    return sampler_id in (1, 2, 3)


class RANSAC(object):
    """Drop-in for the reference's RANSAC (ransac.py:6-200): __call__(matches [N,4], logits [N], K1, K2, gt_model)
    -> (best_model | {iteration: models}, best_mask, best_score, iterations)."""

    def __init__(self, estimator, sampler, scoring, fmat=False, train=False, ransac_batch_size=64, sampler_id=0,
                 weighted=0, threshold=1e-3, confidence=0.999, max_iterations=5000, lo=0, lo_iters=64, eps=1e-5):
        self.estimator = estimator
        self.sampler = sampler
        self.scoring = scoring
        self.lo = lo
        self.lo_iters = lo_iters
        self.fmat = fmat
        self.train = train
        self.ransac_batch_size = ransac_batch_size
        self.sampler_id = sampler_id
        self.weighted = weighted
        self.threshold = threshold
        self.confidence = confidence
        self.max_iterations = max_iterations
        self.eps = eps
        self._soft0 = None       # y_soft of hypothesis 0 of the last batch (weighted F refit, ransac.py:151-153)
        self.fused = True        # test mode with this package's own plugins: run on the device-resident batched driver
        self.graph = True        # ... and, when a call is at most 8 device rounds, as ONE replayed HIP graph per call (_GraphedCall)
        # hypotheses per device round of the replayed call (BatchedRANSAC.super_hypotheses): the first 2048 hypotheses -- however many
        # batches of `ransac_batch_size` that is -- then everything `max_iterations` still allows in rounds of 4096, walked batch by
        # batch on the device with the loop's own stop rule: `-rbs 64 / 5000` is two device rounds, like `-rbs 1024`.  One pair
        # leaves the chip empty, so a device round of 2048 hypotheses costs 15-20 us more than one of 1024 while a second round
        # costs 70-90: measured per pair at 2000 points, 32 pairs of 36 % inliers (24 of them need a second batch of 1024), ms:
        #   (1024, 1024) 0.210 | (1024, 1024, 4096) 0.189 | (2048, 4096) 0.148 | (2048, 1024, 4096) 0.156 | (3072, 2048) 0.157 |
        #   (5120,) 0.168; with ransac_batch_size = 64: 0.221 | 0.198 | 0.154 | 0.165 | 0.165 | 0.175   (scratch/runs/r6_gpu_e.sh)
        self.graph_hypotheses = (2048, 4096)
        self.max_graphs = 8      # replayed calls kept (one per point count and device), least recently used evicted
        self._fast = None
        self._fast_cfg = None
        self._graphs = collections.OrderedDict()
        if lo:
            raise NotImplementedError("local optimisation is out of scope (it never ran in the reference either: "
                                      "lo defaults to 0 and lo=3 raises TypeError, SURVEY Q2)")

    def adaptive_iteration_number(self, inlier_number, point_number, confidence):
        """ransac.py:202-215, the reference's method form (sample size, eps and the cap come from the object)."""
        return adaptive_iteration_number(inlier_number, point_number, self.estimator.sample_size, confidence, self.eps,
                                         self.max_iterations)

    # -- one batch: sample -> gather -> solve; returns models [B,S,3,3], valid [B,S], soft weights
    def _hypotheses(self, matches, logits, gumbels=None):
        B = self.ransac_batch_size
        k = self.sampler.num_samples
        own_sampler = hasattr(self.sampler, "_next_seed") and hasattr(self.sampler, "tau")
        if _is_gumbel(self.sampler_id) and own_sampler:
            seed = self.sampler._next_seed()
            g = None if gumbels is None else gumbels.unsqueeze(0)
            lg = logits.unsqueeze(0).to(matches.dtype)
            if not self.train and not self.weighted:
                # test mode: `points[samples != 0]` (ransac.py:65) -- the index sets and the points themselves
                idx = ops.gumbel_topk(lg, B, k, self.sampler.tau, g, seed, soft=False)["idx"]
                samples, w = ops.gather(matches.unsqueeze(0), idx)[0], None
            else:
                samples, w, _ = ops.SampleGather.apply(matches.unsqueeze(0), lg, B, k, self.sampler.tau, g, seed)
                samples, w = samples[0], w[0]
                if self.weighted and self.fmat and not self.train:      # the refit's `soft_weights[0, ...]` (ransac.py:151-153)
                    self._soft0 = ops.soft_weights_row0(lg, k, self.sampler.tau, g, seed)[0]
        elif _is_gumbel(self.sampler_id):
            # a third-party sampler with the reference's duck-typed contract only (ransac.py:63-65,73):
            # sample(logits) -> (ret [B,N], y_soft [B,N]); the straight-through gather in plain torch ops
            ret, y_soft = self.sampler.sample(logits)
            pts = matches.repeat([ret.shape[0], 1, 1]) * ret.unsqueeze(-1)
            samples = pts[ret != 0].view(ret.shape[0], -1, matches.shape[-1])
            w = y_soft[ret != 0].view(ret.shape[0], -1)
            self._soft0 = y_soft[0].detach()
        else:
            # the reference calls sampler.sample() (ransac.py:59); this package's UniformSampler takes the point count.  The call
            # form is read off the signature -- catching TypeError would also swallow one raised INSIDE a plugin and call it twice
            idx = self.sampler.sample(matches.shape[0]) if _takes_argument(self.sampler.sample) else self.sampler.sample()
            samples, w = matches[idx], None
        wts = w if self.weighted else None
        if hasattr(self.estimator, "estimate_model_slots"):
            return self.estimator.estimate_model_slots(samples, wts)
        # a third-party estimator with the reference's contract only (ransac.py:71-76): estimate_model(samples[, w]) ->
        # [S*B', 3, 3]; slots = consecutive groups per sample, every finite model valid
        models = self.estimator.estimate_model(samples, wts) if wts is not None else self.estimator.estimate_model(samples)
        nb = samples.shape[0]
        # fixed slots: exactly S models per sample, in sample order.  S is the estimator's `solutions_per_sample` when it declares
        # one; otherwise 1 / 10 / 4 for [B,3,3] / five-point / seven-point outputs of the reference's shapes (a ragged output that
        # merely happens to divide evenly would be mis-grouped silently, so nothing is inferred from divisibility alone)
        S = getattr(self.estimator, "solutions_per_sample", None)
        if S is None and models is not None and models.shape[0] in (nb, 10 * nb, 4 * nb):
            S = models.shape[0] // nb
        if models is None or S is None or models.shape[0] != S * nb:
            raise ValueError("a plugin estimator must return exactly S models per sample in sample order (declare "
                             "`solutions_per_sample`): got %s for %d samples" % (None if models is None else tuple(models.shape), nb))
        models = models.reshape(nb, S, 3, 3)
        return models, torch.isfinite(models).flatten(2).all(-1)

    def _make_fast(self, solver, seed=0, super_hypotheses=None):
        drv = BatchedRANSAC(solver, ransac_batch_size=self.ransac_batch_size, train=False, threshold=self.threshold,
                            confidence=self.confidence, max_iterations=self.max_iterations, tau=self.sampler.tau,
                            weighted=self.weighted, refit=True, eps=self.eps, seed=seed)
        drv.super_hypotheses = super_hypotheses
        return drv

    def _fused_solver(self):
        """Name of the BatchedRANSAC solver equivalent to this object's plugins, or None (custom plugins, uniform
        sampler, train mode): test mode then runs on the device-resident driver with P = 1 -- same result, no host
        round-trip per batch except the termination read-back (1.6 -> 0.5 ms per pair at 2000 points x 1024 hypotheses)."""
        from .estimators import (EssentialMatrixEstimator, EssentialMatrixEstimatorNister, FundamentalMatrixEstimatorNew)
        from .samplers import GumbelSoftmaxSampler
        from .scorings import MSACScore
        if self.train or not _is_gumbel(self.sampler_id) or type(self.sampler) is not GumbelSoftmaxSampler:
            return None
        if type(self.scoring) is not MSACScore:
            return None
        k = self.sampler.num_samples
        if type(self.estimator) is EssentialMatrixEstimatorNister and k == 5 and not self.fmat:
            return "nister"
        if type(self.estimator) is EssentialMatrixEstimator and k == 5 and not self.fmat:
            return "stewenius"
        if type(self.estimator) is FundamentalMatrixEstimatorNew and k == 8 and self.fmat:
            return "f8"
        return None

    def __call__(self, matches, logits, K1, K2, gt_model, gumbels=None):
        """`gumbels` (optional, list of [B,N] tensors, one per batch) replaces the in-kernel noise: parity runs."""
        self._soft0 = None       # the weighted refit only ever uses soft weights drawn in THIS call
        solver = self._fused_solver() if self.fused else None
        if solver is not None and matches.is_cuda:
            cfg = (solver, self.ransac_batch_size, self.threshold, self.confidence, self.max_iterations, self.sampler.tau,
                   self.weighted, self.eps)
            cfg = cfg + (self.graph_hypotheses,)
            if self._fast_cfg != cfg:     # public attributes may change between calls (sweeps)
                self._fast, self._graphs, self._fast_cfg = None, collections.OrderedDict(), cfg
                self._graph_rounds = len(self._make_fast(solver, super_hypotheses=self.graph_hypotheses).plan())
            if (self.graph and gumbels is None and matches.dtype == torch.float32 and not self.weighted
                    and not torch.cuda.is_current_stream_capturing() and self._graph_rounds <= 8):
                # The reference calls this object one pair at a time (model_cl.py:488-490, test.py:38): ~25 launches of 5-30 us
                # per pair.  Round 5: the whole call -- threshold, every round, the adaptive stop taken on the device, refit --
                # is captured once per (point count, device) and replayed: one graph launch + one staging copy per pair.
                # Round 6: device rounds of `graph_hypotheses` hypotheses, whatever `ransac_batch_size` is.
                key = (matches.shape[0], matches.device.index)
                g = self._graphs.get(key)
                if g is None:
                    while len(self._graphs) >= max(1, self.max_graphs):      # bounded: variable-N inputs re-capture, the oldest goes
                        self._graphs.popitem(last=False)
                    g = self._graphs[key] = _GraphedCall(self._make_fast(solver, seed=self.sampler._next_seed(),
                                                                         super_hypotheses=self.graph_hypotheses),
                                                         matches.shape[0], matches.device)
                else:
                    self._graphs.move_to_end(key)
                return g(matches, logits, K1, K2)
            if self._fast is None:
                self._fast = self._make_fast(solver)
            self._fast.seed = self.sampler._next_seed()
            out = self._fast(matches.unsqueeze(0), logits.unsqueeze(0).to(matches.dtype), K1, K2,
                             gumbels=None if gumbels is None else [g.unsqueeze(0) for g in gumbels])
            return out["model"][0], out["mask"][0], out["score"][0], int(out["iterations"][0])
        iterations = 0
        best_score = 0
        point_number = matches.shape[0]
        best_mask, best_model = [], []
        models_out: Dict[int, torch.Tensor] = {}
        threshold = normalized_threshold(self.threshold, K1, K2, self.fmat)
        threshold = float(threshold) if not isinstance(threshold, float) else threshold
        max_iters = self.max_iterations
        batch = 0
        while iterations < max_iters:
            g = None
            if gumbels is not None:
                if batch >= len(gumbels):
                    break
                g = gumbels[batch]
            models, valid = self._hypotheses(matches, logits, g)
            batch += 1
            B, S = valid.shape
            if self.train:
                if S == 1:
                    chosen, keep = models[:, 0], valid[:, 0]
                elif self.sampler.num_samples == 8:
                    # ransac.py:82-83: with the 8-point sampler every estimated model is kept (no best-of-S); here that is
                    # every VERIFIED solution of every sample (the reference's ten slots include the non-real ones)
                    chosen, keep = models.reshape(B * S, 3, 3), valid.reshape(B * S)
                else:
                    chosen, which = ops.select_closest_autograd(models.unsqueeze(0), valid.unsqueeze(0),
                                                                gt_model.unsqueeze(0))
                    chosen, keep = chosen[0], which[0] >= 0
                models_out[iterations] = chosen[keep]
            else:
                flat = models.reshape(B * S, 3, 3)
                scores, masks = self.scoring.score(matches, flat, threshold)
                scores = torch.where(valid.reshape(-1) & ~torch.isnan(scores), scores, torch.full_like(scores, -1.0))
                best_idx = torch.argmax(scores)
                if scores[best_idx] > best_score or iterations == 0:
                    best_score = scores[best_idx]
                    best_mask = masks[best_idx]
                    best_model = flat[best_idx]
                    best_inlier_number = int(torch.sum(best_mask))
                    max_iters = min(self.max_iterations,
                                    adaptive_iteration_number(best_inlier_number, point_number, self.estimator.sample_size,
                                                              self.confidence, self.eps, self.max_iterations))
            iterations += self.ransac_batch_size

        if self.train:
            return models_out, best_mask, best_score, iterations

        # final refit on the inliers (ransac.py:148-195); not differentiable
        with torch.no_grad():
            inl = best_mask.nonzero(as_tuple=True)[0]
            slots = hasattr(self.estimator, "estimate_model_slots")
            cvalid = None
            rw = None
            if self.fmat:
                pts_ = matches[inl].unsqueeze(0) if inl.numel() >= 8 else None
                if self.weighted and pts_ is not None and getattr(self, "_soft0", None) is not None:
                    rw = self._soft0[inl].unsqueeze(0)      # ransac.py:151-153: soft_weights[0, inlier_indices[0]], last batch
            else:
                # pymagsac absent: Nister on ALL points in f64 as one sample (ransac.py:157-165 -> nister.py:64-65)
                pts_ = matches.unsqueeze(0).double()
            if pts_ is None:
                cand = None
            elif slots:      # fixed-shape slots + validity: the eye(3) fillers of failed solves must not compete
                cand, cvalid = self.estimator.estimate_model_slots(pts_, rw) if rw is not None else self.estimator.estimate_model_slots(pts_)
                cand, cvalid = cand.reshape(-1, 3, 3), cvalid.reshape(-1)
            else:
                cand = self.estimator.estimate_model(pts_, rw) if rw is not None else self.estimator.estimate_model(pts_)
            if cand is None or cand.shape[0] == 0:
                if not isinstance(best_model, torch.Tensor):
                    best_model = torch.eye(3, device=matches.device, dtype=matches.dtype)
            else:
                cand = cand.to(matches.dtype)
                scores, _ = self.scoring.score(matches, cand, threshold)
                scores = torch.where(torch.isnan(scores), torch.full_like(scores, -1.0), scores)
                if cvalid is not None:
                    scores = torch.where(cvalid, scores, torch.full_like(scores, -1.0))
                if scores.max() > best_score:
                    b = torch.argmax(scores)
                    best_model, best_score = cand[b], scores[b]
        return best_model, best_mask, best_score, iterations


class _GraphedCall(object):
    """One test-mode RANSAC call on one image pair as a replayed HIP graph (drop-in path of RANSAC.__call__).

    The captured call is BatchedRANSAC with P = 1, seeds advanced on the device (`device_seeds`) and the adaptive stop taken
    on the device (`device_termination`): every round is in the graph, the kernels of a round skip a pair that has terminated.
    Inputs are staged into the buffers the graph was captured on (one multi-tensor copy), results are handed out as fresh
    tensors (one more): the caller keeps them across calls, as `ret.append(Es)` of model_cl.py:492 does.  `iterations` is
    returned as a 0-dim int32 tensor -- reading it as an int would be the only host synchronisation of the call."""

    def __init__(self, driver, N, device):
        from .graphs import GraphedStep
        self.driver = driver.device_seeds(device)
        driver.device_termination = True
        self.matches = torch.zeros(1, N, 4, device=device)
        self.logits = torch.zeros(1, N, device=device)
        self.K1 = torch.eye(3, device=device).unsqueeze(0).clone()
        self.K2 = torch.eye(3, device=device).unsqueeze(0).clone()
        self.matches[0, :, :2] = torch.rand(N, 2, device=device)      # something solvable for the warm-up calls
        self.matches[0, :, 2:] = self.matches[0, :, :2] + 0.01 * torch.rand(N, 2, device=device)
        self._eye = None
        self.step = GraphedStep(self._run, warmup=2)

    def _run(self):
        out = self.driver(self.matches, self.logits, self.K1, self.K2)
        # one buffer for everything a call returns: model 36 B | score 4 | iterations 4 | mask N (handed out by ONE copy per call).
        # Round 6: the one-pair state of a device-terminated f32 call already lives in such a buffer (ops.RansacState.packed)
        if out.get("packed") is not None:
            return out["packed"]
        u8 = lambda t: t.contiguous().view(torch.uint8).reshape(-1)
        return torch.cat([u8(out["model"][0]), u8(out["score"][:1]), u8(out["iterations"][:1]), u8(out["mask"][0])])

    def __call__(self, matches, logits, K1, K2):
        src = [matches, logits.to(torch.float32)]
        dst = [self.matches[0], self.logits[0]]
        if K1 is not None and K2 is not None:
            src += [K1.to(device=matches.device, dtype=torch.float32).reshape(3, 3), K2.to(device=matches.device, dtype=torch.float32).reshape(3, 3)]
        else:
            # no intrinsics: the eager path hands None to dr_ransac_init (threshold used as is).  The captured call always normalises
            # with the staged K1 / K2, so identity goes in -- f = (1 + 1 + 1 + 1) / 4 = 1, threshold unchanged, exactly -- instead of
            # whatever the previous pair left in the buffers (round-5 advice)
            if self._eye is None:
                self._eye = torch.eye(3, device=matches.device)
            src += [self._eye, self._eye]
        dst += [self.K1[0], self.K2[0]]
        torch._foreach_copy_(dst, src)
        packed = self.step().clone()
        model = packed[:36].view(torch.float32).view(3, 3)
        score = packed[36:40].view(torch.float32)[0]
        iterations = packed[40:44].view(torch.int32)[0]
        mask = packed[44:].view(torch.bool)
        return model, mask, score, iterations


class RANSAC3D(object):
    """Drop-in for the reference's RANSAC3D (ransac.py:303-450).  Train mode as in the reference; test mode (dead code
    there) = arg-min of the residual sum."""

    def __init__(self, estimator, sampler, scoring, fmat=False, train=False, ransac_batch_size=64, sampler_id=0,
                 weighted=0, threshold=1e-3, confidence=0.999, max_iterations=5000, lo=0, lo_iters=64, eps=1e-5,
                 flag=True):
        self.estimator = estimator
        self.sampler = sampler
        self.scoring = scoring
        self.train = train
        self.ransac_batch_size = ransac_batch_size
        self.sampler_id = sampler_id
        self.threshold = threshold
        self.confidence = confidence
        self.max_iterations = max_iterations
        self.eps = eps
        self.flag = flag

    def adaptive_iteration_number(self, inlier_number, point_number, confidence):
        """ransac.py:452-465 (the same rule as RANSAC's)."""
        return adaptive_iteration_number(inlier_number, point_number, self.estimator.sample_size, confidence, self.eps,
                                         self.max_iterations)

    def __call__(self, matches, logits, gt_model, valid=False, gumbels=None):
        train = self.train and not valid
        B = self.ransac_batch_size
        iterations, batch = 0, 0
        models, residuals, mean_residuals = {}, {}, {}
        best_model, best_score = None, float("inf")
        while iterations < self.max_iterations:
            g = None
            if gumbels is not None:
                if batch >= len(gumbels):
                    break
                g = gumbels[batch].unsqueeze(0)
            batch += 1
            if _is_gumbel(self.sampler_id):
                # the sampler's own sample size: 3, or 8 for sampler id 3 (model_cl.py:185-207 builds an 8-point sampler and
                # the reference feeds its 8-point samples to the rigid solver, which accepts n >= 3)
                k = int(getattr(self.sampler, "num_samples", 3))
                if not 3 <= k <= 8:
                    raise ValueError("RANSAC3D needs a sampler with 3 <= num_samples <= 8")
                samples, _, _ = ops.SampleGather.apply(matches.unsqueeze(0), logits.unsqueeze(0).to(matches.dtype), B, k,
                                                       self.sampler.tau, g, self.sampler._next_seed())
                samples = samples[0]
            else:
                samples = matches[self.sampler.sample(matches.shape[0])]
            est, R, t, _ = self.estimator.estimate_model(samples, flag=self.flag)
            ok = self.estimator.last_valid
            res, mean_res, _ = self.estimator.squared_residual(matches[:, :3], matches[:, 3:],
                                                               est[:, :3, :].transpose(-1, -2))
            if train:
                models[iterations] = est[ok]
                residuals[iterations] = res
                mean_residuals[iterations] = mean_res
            else:
                b = torch.argmin(torch.where(ok, res, torch.full_like(res, float("inf"))))
                if float(res[b]) < best_score:
                    best_score, best_model = float(res[b]), est[b]
            iterations += B
        if train:
            return models, residuals, mean_residuals, 0, iterations
        return best_model, residuals, mean_residuals, best_score, iterations


class BatchedRANSAC(object):
    """The (pair x hypothesis) grid in one go: all P pairs sample, solve, score and select per round with a fixed
    number of launches (K1, K2, K3, K4, K6) and no [B,N] / [M,N] tensor in HBM unless `keep_masks` is set.

    solver: "nister" | "stewenius" | "f8" | "f7".  Test mode reproduces ransac.py:109-195 per pair (arg-max,
    adaptive stop with the per-pair inlier count evaluated on the device, final refit); train mode returns the
    chosen models [P, rounds*B, 3, 3] with autograd to `logits`.
    """

    _SOLVERS = {"nister": (5, 10), "stewenius": (5, 10), "f8": (8, 1), "f7": (7, 4)}

    def __init__(self, solver="nister", ransac_batch_size=1024, train=False, threshold=0.75, confidence=0.999,
                 max_iterations=5000, tau=1.0, seed=0, weighted=0, keep_masks=False, refit=True, eps=1e-5,
                 sampling="gumbel", num_samples=None):
        # num_samples: points per sample when it is not the solver's minimal count -- 8 with solver="nister" is the reference's
        # `-sam 3` (8-point Gumbel sampler) feeding the five-point estimator (ransac.py:82-83; nister.py:64-65 runs on all rows)
        # sampling: "gumbel" = the reference's sampler (noise for every point of every hypothesis, top-k);
        # "topdown" = the same index-set distribution drawn as k sequential soft-max draws without replacement
        # (ops.topdown_sample, O(B k log N)); test mode only, no soft weights (weighted=0), no explicit noise.
        # "uniform" = UniformSampler.batch_generate (samplers/uniform_sampler.py:15-19: randint(0, N - 1), with replacement,
        # the last point never drawn) for all pairs in one launch; index sets only, logits ignored.
        if sampling not in ("gumbel", "topdown", "uniform"):
            raise ValueError("sampling must be 'gumbel', 'topdown' or 'uniform'")
        if sampling in ("topdown", "uniform") and (train or weighted):
            raise ValueError(f"{sampling} sampling yields index sets only: train mode and weighted=1 need the Gumbel sampler")
        self.sampling = sampling
        self.pipeline = True     # test mode: issue round r+1's sampler/solver on a second stream while round r is scored
        # test mode, round 5: True = every round of a call is ISSUED and the adaptive stop of ransac.py:135-144 is taken on the
        # device (the kernels of a round skip the pairs whose counter has reached its bound, ops `gate=`): no read-back, so a
        # whole call -- however many rounds the data ask for -- is capturable in one HIP graph.  Costs a handful of empty
        # launches per unneeded round; meant for few rounds (max_iterations / ransac_batch_size <= 8), refused above 16.
        self.device_termination = False
        self.sync_every = None   # device rounds between termination read-backs; None = max(1, 256 // hypotheses per device round)
        self._pipe = None
        self.solver = solver
        self.k, self.S = self._SOLVERS[solver]
        if num_samples is not None and num_samples != self.k:
            if solver not in ("nister", "f8") or not self.k < num_samples <= 8:
                raise ValueError(f"solver {solver!r} takes {self.k} points per sample (non-minimal samples: 'nister' / 'f8', up to 8)")
            self.k = int(num_samples)
        self.B = ransac_batch_size
        self.train = train
        self.threshold = threshold
        self.confidence = confidence
        self.max_iterations = max_iterations
        self.tau = tau
        self.seed = seed
        self.calls = 0
        self.weighted = weighted
        self.keep_masks = keep_masks
        self.refit = refit
        self.eps = eps
        self.fmat = solver in ("f8", "f7")
        self._race_ws = None     # per-call weights of the one-logarithm sampler (written by dr_ransac_init, read by every round)
        self._side = None
        self._gap = None         # scratch word of the one-launch dispatch gap in front of the sampler (see __call__)
        self._dev_seed = None
        self._seed_queue = []
        # Super-rounds (round 6, test mode).  The loop of ransac.py:55-144 runs `ransac_batch_size` hypotheses per iteration of a
        # Python loop; with the reference's default of 64 a call is up to 79 iterations of four launches each.  A DEVICE round here is
        # R consecutive batches at once: the sampler draws row b of the round with the noise of row b % B of batch b // B (per-call
        # seeds are consecutive integers), solver and scoring see R * B hypotheses, and dr_ransac_update walks the R sub-batches IN
        # ORDER with the loop's own stop rule -- (model, mask, score, iterations) are those of the batch-by-batch loop, bit for bit,
        # whatever R is; hypotheses behind the stop are speculative work.  (h_0, h_1, ...) = hypotheses per device round, the last
        # entry repeated; None = automatic: rounds of 1024 hypotheses when ransac_batch_size is below 1024,
        # one batch per round otherwise; False = always one batch per round (the host loop of rounds 1-5).
        self.super_hypotheses = None

    def _next_seed(self):
        if self._dev_seed is not None:       # seeds advanced on the device (device_seeds(): graph-capturable steps)
            self.calls += 1
            if self._seed_queue:             # drawn ahead for every batch of this call by ONE launch (device_termination)
                blk, pos = self._seed_queue
                self._seed_queue = (blk, pos + 1) if pos + 1 < blk.shape[0] else []
                return blk[pos:pos + 1]
            return self._dev_seed.next()
        s = (self.seed * 0x9E3779B97F4A7C15 + self.calls) & (2 ** 64 - 1)
        self.calls += 1
        return s

    def _next_seeds(self, R):
        """the seed of the next batch of a device round of R batches; the R - 1 batches behind it take the consecutive seeds
        (consumed here: a batch-by-batch driver with the same base seed draws the same hypotheses)"""
        if R == 1:
            return self._next_seed()
        if self._dev_seed is not None:
            self.calls += R
            if self._seed_queue:
                blk, pos = self._seed_queue
                self._seed_queue = (blk, pos + R) if pos + R < blk.shape[0] else []
                return blk[pos:pos + 1]
            return self._dev_seed.next_block(R)[:1]
        s = (self.seed * 0x9E3779B97F4A7C15 + self.calls) & (2 ** 64 - 1)
        self.calls += R
        return s

    def plan(self, n_batches=None, dtype=torch.float32):
        """Batches per device round of a test-mode call (see `super_hypotheses`): a list summing to the number of batches the loop
        of ransac.py:55 can run at most, ceil(max_iterations / ransac_batch_size)."""
        if n_batches is None:
            n_batches = max(1, math.ceil(self.max_iterations / self.B))
        sh = self.super_hypotheses
        one = [1] * n_batches
        if sh is False or self.train or self.sampling != "gumbel" or self.keep_masks or dtype != torch.float32:
            return one
        if self.weighted:
            # weighted rows in the minimal solves (ransac.py:70-74) need the soft weights of every batch, and the weighted refit wants
            # the row-0 soft weights of the LAST batch a pair ran: batch by batch (a super-round samples index sets only)
            return one
        if sh is None:
            if self.B >= 1024:
                return one
            sh = (1024, 1024)
        # (dr_ransac_update walks at most 512 sub-batches per launch)
        sizes = [min(512, max(1, int(h) // self.B)) for h in sh]
        out, left = [], n_batches
        while left > 0:
            r = min(left, sizes[min(len(out), len(sizes) - 1)])
            out.append(r)
            left -= r
        return out

    def device_seeds(self, device):
        """From the next call on the per-call seed is computed on the device (ops.DeviceSeed) -- the same sequence of seeds,
        hence the same hypotheses, but nothing about a call depends on a host-side counter any more, so a call with
        max_iterations <= ransac_batch_size (one round, no read-back) can be captured in a HIP graph and replayed
        (differentiable_ransac_amd.graphs.GraphedStep)."""
        self._dev_seed = ops.DeviceSeed(self.seed, device, self.calls)
        return self

    def hypotheses(self, matches, logits, gumbels=None):
        """matches [P,N,4], logits [P,N] -> models [P,B,S,3,3], valid [P,B,S], idx [P,B,k] (differentiable w.r.t. logits)."""
        return self._hypotheses(matches, logits, gumbels)[:3]

    def _hypotheses(self, matches, logits, gumbels=None, gate=None, R=1):
        """hypotheses() + the (seed, noise) pair of the draw when the weighted refit will need row 0's soft weights again
        (returned, not stashed on self: two rounds are in flight on two streams when `pipeline` is on), else None.
        R > 1 (test mode, see `super_hypotheses`): R consecutive batches in one go -> models [P, R * B, S, 3, 3]; explicit noise is the
        R batches' tensors concatenated along the hypothesis axis."""
        if R > 1:
            Bq = R * self.B
            if gumbels is None and matches.dtype == torch.float32 and logits.dtype == torch.float32 and matches.shape[-1] == 4:
                idx, samples = ops.gumbel_topk_gather(matches, logits, Bq, self.k, self.tau, self._next_seeds(R), gate=gate, sub=self.B,
                                                      race_ws=self._race_ws)
            else:
                if gumbels is None:
                    raise ValueError("super-rounds with in-kernel noise serve f32 two-view correspondences (plan() says so)")
                idx = ops.gumbel_topk(logits, Bq, self.k, self.tau, gumbels, self._next_seeds(R), soft=False)["idx"]
                samples = ops.gather(matches, idx)
            if self.solver in ("nister", "stewenius"):
                if gate is not None and self.k == 5 and samples.dtype == torch.float32:
                    models, valid = ops.solve_essential_gated(samples, self.solver, gate)
                else:
                    models, valid = ops.solve_essential(samples, None, self.solver)
            elif self.solver == "f8":
                F, v = ops.solve_fundamental8(samples, None)
                models, valid = F.unsqueeze(2), v.unsqueeze(2)
            else:
                models, valid = ops.solve_f7(samples)
            return models, valid, idx, None
        if self.weighted and self.solver == "f8" and not self.train and self.refit:
            # the weighted LSQ refit (ransac.py:151-153) needs y_soft of hypothesis 0 of the LAST batch a pair ran: the seed is
            # handed back so that __call__ can re-draw that one row (ops.soft_weights_row0).  weighted=1 implies the Gumbel
            # sampler (__init__ refuses it with 'uniform' / 'topdown')
            seed = self._next_seed()
            samples, w, idx = ops.SampleGather.apply(matches, logits, self.B, self.k, self.tau, gumbels, seed)
            F, v = ops.solve_fundamental8(samples, w)
            return F.unsqueeze(2), v.unsqueeze(2), idx, (seed, gumbels)
        if (self.sampling == "uniform" and gumbels is None and self.solver == "f8" and self.k == 8
                and matches.dtype == torch.float32 and matches.shape[-1] == 4):
            # sampler + gather + 8-point solve in ONE launch (BASELINE configs[0] is launch-bound: six launches -> four)
            idx, F, v = ops.solve_f8_uniform(matches, self.B, self._next_seed())
            return F.unsqueeze(2), v.unsqueeze(2), idx, None
        if self.sampling == "uniform" and gumbels is None:
            idx = ops.uniform_sample(matches.shape[0], self.B, self.k, matches.shape[1], self._next_seed(), matches.device)
            samples, w = ops.gather(matches, idx), None
        elif self.sampling == "topdown" and gumbels is None:
            idx = ops.topdown_sample(logits, self.B, self.k, self._next_seed())
            samples, w = ops.gather(matches, idx), None
        elif not self.train and not self.weighted:
            # test mode consumes the index sets only (`points[samples != 0]`, ransac.py:65): no soft-max statistics, and
            # the samples are the points themselves (not points x a straight-through value of 1 +- 1 ulp)
            if gumbels is None and matches.dtype == torch.float32 and logits.dtype == torch.float32 and matches.shape[-1] == 4:
                idx, samples = ops.gumbel_topk_gather(matches, logits, self.B, self.k, self.tau, self._next_seed(), gate=gate,
                                                      race_ws=self._race_ws)   # one launch
                w = None
                if gate is not None and self.solver in ("nister", "stewenius") and self.k == 5:
                    models, valid = ops.solve_essential_gated(samples, self.solver, gate)
                    return models, valid, idx, None
            else:
                idx = ops.gumbel_topk(logits, self.B, self.k, self.tau, gumbels, self._next_seed(), soft=False)["idx"]
                samples, w = ops.gather(matches, idx), None
        else:
            samples, w, idx = ops.SampleGather.apply(matches, logits, self.B, self.k, self.tau, gumbels, self._next_seed())
        wts = w if self.weighted else None
        if self.solver in ("nister", "stewenius"):
            models, valid = ops.solve_essential(samples, wts, self.solver)
        elif self.solver == "f8":
            F, v = ops.solve_fundamental8(samples, wts)
            models, valid = F.unsqueeze(2), v.unsqueeze(2)
        else:
            models, valid = ops.solve_f7(samples)
        return models, valid, idx, None

    def __call__(self, matches, logits, K1=None, K2=None, gt_model=None, gumbels=None):
        P, N, _ = matches.shape
        dev, dt = matches.device, matches.dtype
        self._race_ws = None      # (weights of THIS call's logits only: set by the test-mode set-up below, dropped by _finish)
        rounds = max(1, math.ceil(self.max_iterations / self.B))
        if self.train:
            out = []
            for r in range(rounds):
                g = None if gumbels is None else gumbels[r]
                if (self.solver == "nister" and self.k == 5 and not self.weighted and matches.dtype == torch.float32
                        and logits.dtype == torch.float32 and logits.requires_grad):
                    # the training path proper (train mode implies the Gumbel sampler: __init__ refuses the index-only samplings):
                    # sampler + gather, then solver + best-of-ten as ONE autograd node whose backward
                    # takes the gradient of the chosen model in sparse form (ops.solve_select_essential)
                    samples, _, _ = ops.SampleGather.apply(matches, logits, self.B, self.k, self.tau, g, self._next_seed())
                    chosen, _, keep, _, _ = ops.solve_select_essential(samples, gt_model)
                    out.append((chosen, keep))
                    continue
                models, valid, _ = self.hypotheses(matches, logits, g)
                if self.S == 1:
                    chosen = models[:, :, 0]
                    keep = valid[:, :, 0]
                elif self.k == 8:      # ransac.py:82-83: the 8-point sampler keeps every model of every sample
                    chosen = models.reshape(P, self.B * self.S, 3, 3)
                    keep = valid.reshape(P, self.B * self.S)
                else:
                    chosen, _, keep = ops.select_closest_autograd(models, valid, gt_model, want_keep=True)
                out.append((chosen, keep))
            if len(out) == 1:            # one batch (train.py's max_iters = 100 with -rbs >= 100): nothing to concatenate
                return out[0]
            return torch.cat([c for c, _ in out], dim=1), torch.cat([k for _, k in out], dim=1)

        with torch.no_grad():
            # threshold normalisation (ransac.py:49-53) + per-pair state in one launch
            use_K = K1 is not None and not self.fmat
            all_masks = None
            matches = matches.contiguous()
            # The essential-matrix refit candidate (Nister on ALL points, ransac.py:157-165) depends on the matches only:
            # it is issued on a side stream and joins before the final scoring -- and it is issued FIRST, before the state set-up
            # (round 5): a refit block wants a whole SIMD's registers and 38.9 KB of LDS on its CU, and once the sampler's 32 768
            # light workgroups are in the queue it does not get them until the sampler's grid runs dry (measured at 128 pairs:
            # launched 6 us after the sampler it ran 15 -> 241 us for 52 us of work, and the solver behind it started 56 us late
            # on the SIMDs it held).
            pre = None
            # device rounds (see `super_hypotheses`): plan[i] batches of B hypotheses in round i
            n_batches = rounds if gumbels is None else min(rounds, len(gumbels))
            plan = self.plan(n_batches, dt) if n_batches > 0 else []
            first = [sum(plan[:i]) for i in range(len(plan))]         # index of the first batch of device round i
            rounds = len(plan)

            def issue_refit():
                if self._side is None:
                    self._side = torch.cuda.Stream(device=dev)
                self._side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(self._side):
                    out_ = ops.refit_essential(matches)
                    if not torch.cuda.is_current_stream_capturing():
                        matches.record_stream(self._side)
                return out_
            if self.refit and not self.fmat:
                pre = issue_refit()
            # (device termination with device seeds: the keys of all batches of the call come out of the set-up launch; one pair in
            #  f32: the state lives in one buffer, which is what the replayed drop-in call hands out)
            draw = self.device_termination and self._dev_seed is not None and n_batches > 1 and gumbels is None
            fold = _FOLD_SETUP and draw
            # the weights of the one-logarithm sampler, once per call, out of the same launch (when that form pays: ops.race_form_pays)
            race_lg = None
            if (_FOLD_SETUP and plan and gumbels is None and self.sampling == "gumbel" and not self.weighted and dt == torch.float32
                    and logits.dtype == torch.float32 and matches.shape[-1] == 4
                    and ops.race_form_pays(P, self.B * plan[0], N, self.tau)):
                race_lg = logits.contiguous()
            st, thr = ops.ransac_init(P, N, self.max_iterations, self.threshold, K1 if use_K else None,
                                      K2 if use_K else None, dev, dt, seeds=(self._dev_seed, n_batches) if fold else None,
                                      packed=_FOLD_SETUP and self.device_termination and P == 1, race_logits=race_lg)
            self._race_ws = st.race_ws
            if draw and not fold:
                st.seeds = self._dev_seed.next_block(n_batches)
            if pre is not None and plan and P * self.B * plan[0] >= 65536:
                # Dispatch order (round 5): a refit block wants a whole SIMD's registers and 38.9 KB of LDS on its CU; once the
                # sampler's 32 768 light workgroups are in the queue it does not get them until the sampler's grid runs dry.
                # The refit waits for this stream's earlier work through an event and lost that race by half a microsecond
                # (rocprofv3, 128 pairs: sampler dispatched at 5.4 us, refit at 5.9 -- it then ran 232 us for 57 us of work and the
                # solver behind it started 56 us late on the SIMDs it still held: step 0.947 -> 1.018 ms with the refit).  One
                # tiny launch in front of the sampler lets the refit's 128 blocks in first: refit 57 us, sampler 178 -> 191 us,
                # solver 146 us, step 0.996 ms.  (Set-up AND refit on the side stream with this stream waiting for the set-up:
                # refit first as well, but 57 us between two steps instead of 17: 1.043 ms.)
                if self._gap is None or self._gap.state.device != matches.device:
                    self._gap = ops.DeviceSeed(0, matches.device)
                self._gap.next()
            # Rounds are pipelined: the hypotheses of round r+1 (sampler + solver, latency-bound, independent of round r's
            # outcome) are issued on a second stream before round r is scored, so they run under K4/K6 and under the
            # host's "does any pair continue?" read-back.  If round r ends the loop they are simply dropped.
            main = torch.cuda.current_stream()

            def noise_of(r):
                if gumbels is None:
                    return None
                if plan[r] == 1:
                    return gumbels[first[r]]
                return torch.cat(list(gumbels[first[r]:first[r] + plan[r]]), dim=1)

            def have_round(r):
                return r < rounds

            def sub_of(r):      # dr_ransac_update's sub_models: the round's models are plan[r] batches of B x S
                return self.B * self.S if plan[r] > 1 else 0

            # (round 5, measured and dropped -- scratch/runs/r5_gpu_w.sh, 128 pairs: the refit issued right before the first
            #  scoring launch instead of up front: step with refit 1.015 -> 1.063 ms; the same on a high-priority stream: 1.069 ms.)
            if self.device_termination:
                if rounds > 16:
                    raise ValueError("device_termination issues every round: at most 16 device rounds per call (see super_hypotheses)")
                want_w = bool(self.weighted and self.solver == "f8" and self.refit)
                last_w = torch.zeros((P, N), device=dev, dtype=dt) if want_w else None
                if draw:
                    self._seed_queue = (st.seeds, 0)    # consecutive seeds, one per BATCH (drawn by dr_ransac_init)
                for r in range(rounds):
                    gate = st if r > 0 else None
                    models, valid, _, row0 = self._hypotheses(matches, logits, noise_of(r), gate=gate, R=plan[r])
                    if want_w:
                        w0 = ops.soft_weights_row0(logits, self.k, self.tau, row0[1], row0[0])
                        last_w = torch.where((st.iters.double() < st.max_iters)[:, None], w0, last_w)
                    flat = models.reshape(P, -1, 3, 3)
                    scores, masks = ops.msac_score(matches, flat, thr, want_masks=self.keep_masks, valid=valid.reshape(P, -1),
                                                   gate=gate)
                    if self.keep_masks:
                        all_masks = masks
                    ops.ransac_update(st, matches, flat, valid.reshape(P, -1), scores, thr, self.B, self.k, self.confidence,
                                      self.eps, sub_models=sub_of(r))
                self._seed_queue = []
                return self._finish(st, matches, thr, pre, last_w, all_masks)
            ahead = None
            if have_round(0):
                h = self._hypotheses(matches, logits, noise_of(0), R=plan[0])
                ahead = (h[0], h[1], h[3])
            r = 0
            want_w = bool(self.weighted and self.solver == "f8" and self.refit)   # (the 7-point solver takes no weights)
            last_w = torch.zeros((P, N), device=dev, dtype=dt) if want_w else None
            while ahead is not None:
                models, valid, row0 = ahead
                ahead = None
                if self.pipeline and have_round(r + 1):
                    if self._pipe is None:
                        self._pipe = torch.cuda.Stream(device=dev)
                    self._pipe.wait_stream(main)          # inputs (and, for explicit noise, the caller's tensors) are ready
                    with torch.cuda.stream(self._pipe):
                        h = self._hypotheses(matches, logits, noise_of(r + 1), R=plan[r + 1])
                        ahead = (h[0], h[1], h[3])
                if want_w:
                    # pairs still iterating in this round take this round's row-0 soft weights; terminated pairs keep theirs
                    # ("the last batch sampled", per pair)
                    w0 = ops.soft_weights_row0(logits, self.k, self.tau, row0[1], row0[0])
                    last_w = torch.where((st.iters.double() < st.max_iters)[:, None], w0, last_w)
                flat = models.reshape(P, -1, 3, 3)
                scores, masks = ops.msac_score(matches, flat, thr, want_masks=self.keep_masks, valid=valid.reshape(P, -1))
                if self.keep_masks:
                    all_masks = masks
                # K6: arg-max, "better?" test, best mask / inlier count and the adaptive stop of ransac.py:135-142, on the device
                ops.ransac_update(st, matches, flat, valid.reshape(P, -1), scores, thr, self.B, self.k, self.confidence,
                                  self.eps, sub_models=sub_of(r))
                r += 1
                if not have_round(r):
                    break
                # host read-back "does any pair continue?" only when another round could follow, and for small batches
                # only every few rounds (pairs that have terminated are frozen on the device by K6, so a round issued
                # after the last pair stopped changes nothing)
                sync_every = max(1, 256 // max(1, self.B * plan[r - 1])) if self.sync_every is None else self.sync_every
                if r % sync_every == 0 and not bool((st.iters.double() < st.max_iters).any()):
                    break
                if ahead is None:
                    h = self._hypotheses(matches, logits, noise_of(r), R=plan[r])
                    ahead = (h[0], h[1], h[3])
                else:
                    main.wait_stream(self._pipe)
                    for t_ in ahead[:2]:
                        t_.record_stream(main)
            if ahead is not None and self._pipe is not None:
                main.wait_stream(self._pipe)              # dropped speculative work: keep the allocator's stream order simple
            return self._finish(st, matches, thr, pre, last_w, all_masks)

    def _finish(self, st, matches, thr, pre, last_w, all_masks):
        """final refit on the inliers of the best model (ransac.py:148-195) and the result dictionary"""
        self._race_ws = None
        best_score, best_model, best_mask, best_inl, iters = (st.best_score, st.best_model, st.best_mask,
                                                              st.best_inliers, st.iters)
        if self.refit:
            if self.fmat:
                F, fvalid = ops.refit_fundamental(matches, best_mask, last_w)   # (weighted) LSQ on the inliers of the best mask
                cand, cvalid = F.unsqueeze(1), fvalid.unsqueeze(1)
            else:
                torch.cuda.current_stream().wait_stream(self._side)
                cand, cvalid = pre
                if not torch.cuda.is_current_stream_capturing():
                    cand.record_stream(torch.cuda.current_stream())
                    cvalid.record_stream(torch.cuda.current_stream())
            # score the candidates and keep the best one where it beats the RANSAC result: one launch, in place
            ops.refit_accept(matches, cand, cvalid, thr, best_score, best_model)
        return dict(model=best_model, mask=best_mask, score=best_score, iterations=iters, inliers=best_inl,
                    masks=all_masks, packed=st.packed)


class BatchedRANSAC3D(object):
    """RANSAC3D (ransac.py:303-450) over a batch of point-cloud pairs in one go: matches [P,N,6] = (p, q), logits [P,N].

    One round = K1 Gumbel top-k (k = 3) -> K2 gather -> K3r rigid SVD solver -> K4r squared residuals of every model
    against every point (+ inlier masks).  Train mode returns what the reference's train branch collects
    (models [P, rounds*B, 4, 4], keep, residual sums [P, rounds*B], mean residual per round) with autograd to the
    logits; test mode (dead code upstream) keeps the arg-min of the residual sum per pair."""

    def __init__(self, ransac_batch_size=2048, train=False, threshold=0.03, max_iterations=1000, tau=1.0, seed=0,
                 flag=True, keep_masks=False):
        self.B = ransac_batch_size
        self.train = train
        self.threshold = threshold
        self.max_iterations = max_iterations
        self.tau = tau
        self.seed = seed
        self.calls = 0
        self._dev_seed = None
        self.flag = flag
        self.keep_masks = keep_masks

    def _next_seed(self):
        if self._dev_seed is not None:
            self.calls += 1
            return self._dev_seed.next()
        s = (self.seed * 0x9E3779B97F4A7C15 + self.calls) & (2 ** 64 - 1)
        self.calls += 1
        return s

    def device_seeds(self, device):
        """See BatchedRANSAC.device_seeds."""
        self._dev_seed = ops.DeviceSeed(self.seed, device, self.calls)
        return self

    def __call__(self, matches, logits, gumbels=None):
        P, N, _ = matches.shape
        rounds = max(1, math.ceil(self.max_iterations / self.B))
        if gumbels is not None:
            rounds = min(rounds, len(gumbels))
        if self.train:
            out = []
            for r in range(rounds):
                g = None if gumbels is None else gumbels[r]
                samples, _, _ = ops.SampleGather.apply(matches, logits, self.B, 3, self.tau, g, self._next_seed())
                model, R, t, scale, valid = ops.solve_rigid_autograd(samples.reshape(P * self.B, 3, 6), None, self.flag)
                model = model.reshape(P, self.B, 4, 4)
                res, _ = ops.rigid_residual_autograd(matches, model, self.threshold)
                out.append((model, valid.reshape(P, self.B), res, res.sum(1) / (self.B * N)))
            return dict(models=torch.cat([o[0] for o in out], 1), keep=torch.cat([o[1] for o in out], 1),
                        residuals=torch.cat([o[2] for o in out], 1), mean_residuals=torch.stack([o[3] for o in out], 1))
        with torch.no_grad():
            matches = matches.contiguous()
            best, best_model = None, None      # "no state yet": the first update writes it (no fill / copy launches per call)
            best_mask = torch.empty((P, N), device=matches.device, dtype=torch.bool) if self.keep_masks else None
            masks = None
            for r in range(rounds):
                g = None if gumbels is None else gumbels[r]
                idx = ops.gumbel_topk(logits, self.B, 3, self.tau, g, self._next_seed(), soft=False)["idx"]
                if matches.dtype == torch.float32:
                    # K2 + K3r in one launch (samples read through the index sets), the round's residual sums cleared on the way
                    # the residual sums are ACCUMULATED (atomics: order-nondeterministic in the last bits) into a buffer that
                    # dr_solve_rigid_gather_f32 clears.  (Rounds 4-5 also zero-filled it outside a graph capture -- a 5 us torch fill
                    # launch per round in the eager step, round-5 review; an error between the two launches raises, so nobody reads
                    # the buffer then.)
                    res = torch.empty((P, self.B), device=matches.device, dtype=torch.float32)
                    model, valid = ops.solve_rigid_gather(matches, idx, self.flag, zero_sums=res)
                    res, masks = ops.rigid_residual(matches, model, self.threshold, self.keep_masks, res=res)
                else:
                    samples = ops.gather(matches, idx)
                    model, R, t, scale, valid = ops.solve_rigid(samples.reshape(P * self.B, 3, 6), None, self.flag)
                    model = model.reshape(P, self.B, 4, 4)
                    res, masks = ops.rigid_residual(matches, model, self.threshold, self.keep_masks)
                # K6 of the 3-D path on the device: arg-min over the valid models, strict "better" test, best model and mask
                # (one launch; was where / min / gather / where x3: ten torch kernels per round)
                best, best_model, _ = ops.ransac3d_update(matches, model, valid.reshape(P, self.B), res, self.threshold, best,
                                                          best_model, best_mask)
            return dict(model=best_model, residual=best, mask=best_mask, masks=masks)
