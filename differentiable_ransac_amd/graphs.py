"""HIP-graph replay of a whole RANSAC step (torch.cuda.graph = hipGraph on ROCm).

A step of the batched drivers is 6 (test mode) to ~25 (train mode, forward + backward) launches of 5-200 us each; issued
one by one from Python they cost 0.1 ms (test) to 0.25-0.5 ms (train: autograd) of host time per step, which is more
than the device needs for the small configurations (BASELINE config 1: 0.05 ms of device time) and about as much as it
needs for the train step.  Captured once and replayed, the same step costs the host one graph launch.

What makes a step capturable: (1) no host read-back inside it -- one round per call, i.e. max_iterations <=
ransac_batch_size, which is how the reference's training loop calls RANSAC; (2) the sampler's seed must not be a
by-value kernel argument (it would be frozen at capture time and every replay would draw the same hypotheses): drivers
switched to `device_seeds()` advance it on the device (`dr_seed_next`), so replay r draws what call r of an eager driver
with the same base seed draws -- bit for bit (tests/test_gpu_round2.py); (3) inputs live in fixed buffers: copy new data
into the tensors the step was captured on (`matches.copy_(...)`), outputs are overwritten by every replay.
"""
from __future__ import annotations

import torch


class GraphedStep:
    """step = GraphedStep(fn); out = step()  -- fn() is run `warmup` times eagerly on a side stream (allocator pools,
    first-call kernel attributes), captured once (the capture pass records the launches, it does not execute them: device
    state such as the seed counter is where the warm-up left it), and replayed by every call.  `fn` may contain an autograd backward
    (`loss.backward()`): gradients then live in static `.grad` buffers (leave `.grad` allocated between steps or set it
    to None INSIDE fn).  The value returned by fn during capture is returned by every call (tensors: static buffers)."""

    def __init__(self, fn, warmup: int = 3):
        self.graph = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        with torch.cuda.graph(self.graph):
            self.out = fn()
        self.replays = 0

    def __call__(self):
        self.graph.replay()
        self.replays += 1
        return self.out
