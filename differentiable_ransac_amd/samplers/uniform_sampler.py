"""Uniform sampler plugin -- interface of samplers/uniform_sampler.py:4-23.

The reference's `sample()` raises TypeError (it calls batch_generate() without the point count,
SURVEY Q1); here `sample(num_points)` works and keeps the randint(0, N-1) semantics: with
replacement, the last point is never drawn."""
import torch

from .. import ops


class UniformSampler(object):
    def __init__(self, batch_size, num_samples, num_points=None, device='cuda', seed=0):
        self.batch_size = batch_size
        self.num_samples = num_samples
        self.num_points = num_points
        self.device = device
        self.seed = seed
        self.calls = 0

    def unique_generate(self, num_points):
        """uniform_sampler.py:11-13: ONE sample of `num_samples` indices into the sequence `num_points` (the reference takes its
        len()), drawn with replacement from [0, len - 1]"""
        s = (self.seed * 0x9E3779B97F4A7C15 + self.calls) & (2 ** 64 - 1)
        self.calls += 1
        # dr_uniform_sample draws U{0..N-2} (randint(0, N - 1) of batch_generate): N = len + 1 covers [0, len - 1]
        return ops.uniform_sample(1, 1, self.num_samples, len(num_points) + 1, s, self.device)[0, 0].long()

    def batch_generate(self, num_points, pairs=None):
        s = (self.seed * 0x9E3779B97F4A7C15 + self.calls) & (2 ** 64 - 1)
        self.calls += 1
        idx = ops.uniform_sample(pairs or 1, self.batch_size, self.num_samples, num_points, s, self.device)
        return idx.long() if pairs else idx[0].long()

    def sample(self, num_points=None):
        n = num_points if num_points is not None else self.num_points
        if n is None:
            raise TypeError("UniformSampler.sample needs num_points (ctor or argument)")
        return self.batch_generate(n)
