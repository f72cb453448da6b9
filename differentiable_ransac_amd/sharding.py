"""Multi-GPU sharding of the hot path: image pairs are independent RANSAC problems, so rank r of G
owns a contiguous block of pairs and runs the full (local pairs x hypotheses) grid with NO data-path collective.
The only exchanges are (a) the throughput reduction of the benchmark (MAX of elapsed time, SUM of hypotheses) and
(b) optional result gathering / the gradient all-reduce of the training step (which belongs to the caller's model),
and (c) when there are fewer pairs than GPUs, the hypothesis-split merge below (one tiny exchange per pair).
Backend: "nccl" (= RCCL over xGMI) on GPUs, "gloo" in the CPU tests.
"""
from __future__ import annotations

from typing import Dict, Tuple

import torch


def pair_range(total_pairs: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of pairs owned by `rank`; sizes differ by at most one."""
    base, extra = divmod(total_pairs, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_pairs(batch: Dict[str, torch.Tensor], rank: int, world: int) -> Dict[str, torch.Tensor]:
    """Slice every [P, ...] tensor of a batch down to this rank's pairs."""
    P = next(iter(batch.values())).shape[0]
    lo, hi = pair_range(P, rank, world)
    return {k: v[lo:hi] for k, v in batch.items()}


def job_throughput(local_hypotheses: int, local_seconds: float, dist=None, device=None) -> Tuple[float, float]:
    """Whole-job (hypotheses/s, seconds) = sum of hypotheses over ranks / max of elapsed time over ranks."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return local_hypotheses / local_seconds, local_seconds
    t = torch.tensor([local_seconds], dtype=torch.float64, device=device)
    h = torch.tensor([float(local_hypotheses)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(h, op=dist.ReduceOp.SUM)
    return float(h[0]) / float(t[0]), float(t[0])


def gather_results(local: torch.Tensor, total_pairs: int, dist=None) -> torch.Tensor:
    """all_gather of a per-pair result tensor [P_local, ...] into [total_pairs, ...] (evaluation only)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    world = dist.get_world_size()
    sizes = [pair_range(total_pairs, r, world) for r in range(world)]
    mx = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad)
    return torch.cat([o[: hi - lo] for o, (lo, hi) in zip(out, sizes)], dim=0)


def allreduce_mean_(grads, dist=None) -> None:
    """One flattened all-reduce (SUM -> mean) of a list of gradient tensors: the training step's only collective.
    622 616 f32 parameters = 2.49 MB for the reference's CLNet: latency-bound on xGMI, so a single bucket."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat /= dist.get_world_size()
    off = 0
    for g in grads:
        g.copy_(flat[off: off + g.numel()].view_as(g))
        off += g.numel()


class AsyncGradientBucket:
    """The training step's collective, issued asynchronously (what DDP does under train.py:150-175, per-pair loop
    model_cl.py:488): one flat gradient bucket, all-reduced SUM -> mean over the ranks.

    `launch()` -- called right after the backward of step i has been enqueued -- hands the bucket to the process group with
    `async_op=True` and returns at once: with the nccl backend (= RCCL) the collective runs on the communicator's own stream
    behind the compute stream's work, the compute stream is NOT blocked.  `wait()` -- called where the optimizer would consume
    the averaged gradient, i.e. AFTER step i + 1's forward + backward have been enqueued -- makes the compute stream (gloo: the
    host) wait for it, so the collective of step i runs under the kernels of step i + 1.  Two buffers alternate: step i + 1's
    backward may fill its bucket while step i's is still in flight.  `trace` records ("launch", i) / ("wait", i) in program
    order (OverlappedStep adds ("enqueued", i)); `exposed_events` holds (start, end) CUDA events around every wait on the
    compute stream: their elapsed time is the part of the collective the step could not hide."""

    def __init__(self, numel: int, device, dist=None, dtype=torch.float32, fill=None):
        self.dist = dist if (dist is not None and dist.is_initialized() and dist.get_world_size() > 1) else None
        self.buf = [torch.zeros(numel, device=device, dtype=dtype) for _ in range(2)]
        if fill is not None:
            for b in self.buf:
                b.copy_(fill)
        self.work = [None, None]
        self.issued = 0          # buckets handed to the process group so far
        self.waited = 0
        self.trace = []
        self.exposed_events = []
        self._cuda = self.buf[0].is_cuda
        self.on_overrun = None       # on_overrun(averaged_bucket_copy, step_index): see bucket()
        self.overrun_reduced = None

    def bucket(self) -> torch.Tensor:
        """The buffer the NEXT launch() will reduce (the backward of the current step writes its gradients here).
        With both buffers in flight the one about to be handed out is the OLDEST in-flight all_reduce's: it is waited for here,
        BEFORE the caller can write into it (round-5 advice: the guard used to sit in launch(), after the backward had already
        overwritten the live buffer); its averaged contents go to `on_overrun` (the optimizer's hook) and are kept as a copy in
        `overrun_reduced` -- the buffer itself is about to be refilled."""
        if self.issued - self.waited >= 2:
            i = self.waited
            reduced = self.wait()
            self.overrun_reduced = reduced.clone()
            if self.on_overrun is not None:
                self.on_overrun(self.overrun_reduced, i)
        return self.buf[self.issued % 2]

    def launch(self) -> None:
        i = self.issued
        if self.issued - self.waited >= 2:
            # bucket() waits before it hands out a live buffer; getting here means the caller filled a buffer it did not obtain
            # from bucket() while both were in flight: the reduced gradients of step i - 2 have been overwritten
            raise RuntimeError("AsyncGradientBucket.launch(): both buffers are in flight -- call bucket() (or wait()) "
                               "before writing the next step's gradients")
        if self.dist is not None:
            self.work[i % 2] = self.dist.all_reduce(self.buf[i % 2], op=self.dist.ReduceOp.SUM, async_op=True)
        self.trace.append(("launch", i))
        self.issued += 1

    def wait(self):
        """Wait for the oldest bucket in flight (no-op when none is); returns the averaged bucket or None."""
        if self.waited >= self.issued:
            return None
        i = self.waited
        ev = None
        if self._cuda:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        if self.work[i % 2] is not None:
            self.work[i % 2].wait()
            self.work[i % 2] = None
            self.buf[i % 2] /= self.dist.get_world_size()
        if ev is not None:
            ev[1].record()
            self.exposed_events.append(ev)
        self.trace.append(("wait", i))
        self.waited += 1
        return self.buf[i % 2]

    def drain(self) -> None:
        while self.waited < self.issued:
            self.wait()


class OverlappedStep:
    """step i = [enqueue forward + backward of step i] -> [wait for bucket i - 1: the optimizer's read] -> [launch bucket i].
    The first launch of step i + 1 is therefore enqueued BEFORE the wait on bucket i (tests/test_sharding_gloo.py asserts the
    order on `bucket.trace`)."""

    def __init__(self, step_fn, bucket: AsyncGradientBucket, on_reduced=None):
        # on_reduced(averaged_bucket, step_index): the optimizer's hook -- called with the averaged gradient of step i - 1 the
        # moment it has been waited for (round-4 advice: the waited bucket used to be dropped); also kept as `last_reduced`
        self.step_fn, self.bucket, self.n, self.on_reduced, self.last_reduced = step_fn, bucket, 0, on_reduced, None

    def __call__(self):
        out = self.step_fn()
        self.bucket.trace.append(("enqueued", self.n))
        reduced = self.bucket.wait()
        if reduced is not None:
            self.last_reduced = reduced
            if self.on_reduced is not None:
                self.on_reduced(reduced, self.n - 1)
        self.bucket.launch()
        self.n += 1
        return out


def hypothesis_seed(seed: int, rank: int) -> int:
    """Sampler seed of `rank` when the HYPOTHESES of one pair are split over ranks (P < G): every rank must draw a
    different stream.  The in-kernel Philox is keyed by the seed, so distinct seeds are independent streams."""
    return (int(seed) + 0x9E3779B97F4A7C15 * (int(rank) + 1)) & (2 ** 64 - 1) if rank else int(seed)


def merge_best(best_score: torch.Tensor, best_model: torch.Tensor, extras=(), dist=None):
    """Hypothesis split (SURVEY 8(e), P < G): every rank ran B/G hypotheses on the SAME pairs and holds a local best
    (score [P], model [P,3,3]); the job's answer per pair is the best over ranks (ties -> lowest rank, the analogue of
    the first arg-max of ransac.py:109).  One all_gather of the scores and one of the 9-float models (+ any per-pair
    `extras`, e.g. the inlier count); no collective touches the [P,M,N] data path.
    Returns (score [P], model [P,3,3], winner_rank [P] int64, extras...)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return (best_score, best_model, torch.zeros(best_score.shape[0], dtype=torch.int64, device=best_score.device),
                *extras)
    world = dist.get_world_size()

    def gathered(t):
        out = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(out, t.contiguous())
        return torch.stack(out, 0)

    sc = gathered(best_score)                                  # [G,P]
    key = torch.where(torch.isnan(sc), torch.full_like(sc, float("-inf")), sc)
    top = key.max(0).values
    ranks = torch.arange(world, device=sc.device)[:, None].expand_as(key)
    win = torch.where(key == top[None], ranks, torch.full_like(ranks, world)).min(0).values   # first rank at the max
    pick = lambda t: gathered(t)[win, torch.arange(t.shape[0], device=t.device)]
    return (pick(best_score), pick(best_model), win, *[pick(e) for e in extras])
