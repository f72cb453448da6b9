"""Multi-GPU sharding of the hot path: image pairs are independent RANSAC problems, so rank r of G
owns a contiguous block of pairs and runs the full (local pairs x hypotheses) grid with NO data-path collective.
The only exchanges are (a) the throughput reduction of the benchmark (MAX of elapsed time, SUM of hypotheses) and
(b) optional result gathering / the gradient all-reduce of the training step (which belongs to the caller's model).
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
