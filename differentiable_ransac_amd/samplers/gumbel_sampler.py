"""Gumbel-softmax top-k sampler plugin -- interface of samplers/gumbel_sampler.py:9-42."""
import torch

from .. import ops


class _DenseSample(torch.autograd.Function):
    """API-faithful dense outputs (ret, y_soft) [B,N] of the reference's `sample()`.

    Forward runs the HIP kernel with the dense outputs enabled.  Backward is the soft-max Jacobian
    applied to (grad_ret + grad_y_soft): ret = y_hard - sg(y) + y, so d ret/d y = 1."""

    @staticmethod
    def forward(ctx, logits, B, k, tau, gumbel, seed):
        r = ops.gumbel_topk(logits.unsqueeze(0), B, k, tau, None if gumbel is None else gumbel.unsqueeze(0), seed,
                            dense=True)
        ctx.save_for_backward(r["y_soft"][0])
        ctx.tau = tau
        ctx.mark_non_differentiable(r["idx"])
        return r["ret"][0], r["y_soft"][0], r["idx"][0]

    @staticmethod
    def backward(ctx, g_ret, g_soft, _):
        (y,) = ctx.saved_tensors
        a = g_ret + g_soft
        gg = y * (a - (y * a).sum(-1, keepdim=True))
        return gg.sum(0) / ctx.tau, None, None, None, None, None


class GumbelSoftmaxSampler():
    """sample(logits [N]) -> (ret [B,N], y_soft [B,N]) exactly as the reference (gumbel_sampler.py:25-42):
    `ret != 0` marks the k selected points of every hypothesis, non-selected entries are exactly 0.

    Extras used by the batched driver (no [B,N] tensors): `sample_indices(logits [P,N])`.
    Noise: in-kernel Philox keyed by (seed, call counter), or explicit via `gumbels=`.
    """

    def __init__(self, batch_size, num_samples, tau=1., device='cuda', data_type=torch.float32, seed=0):
        self.batch_size = batch_size
        self.num_samples = num_samples
        self.device = device
        self.dtype = data_type
        self.tau = tau
        self.seed = seed
        self.calls = 0
        self.last_indices = None

    def _next_seed(self):
        s = (self.seed * 0x9E3779B97F4A7C15 + self.calls) & (2 ** 64 - 1)
        self.calls += 1
        return s

    def sample(self, logits=None, num_points=2000, selected=None, gumbels=None):
        if logits is None:
            logits = torch.ones(num_points, device=self.device, dtype=self.dtype, requires_grad=True)
        else:
            logits = logits.to(self.dtype).to(self.device)
        ret, y_soft, idx = _DenseSample.apply(logits, self.batch_size, self.num_samples, self.tau, gumbels,
                                              self._next_seed())
        self.last_indices = idx
        return ret, y_soft

    def sample_indices(self, logits, gumbels=None, want_noise=False):
        """logits [P,N] -> dict(idx [P,B,k] int32 ascending, y_sel, lse[, gumbel])."""
        return ops.gumbel_topk(logits.to(self.dtype), self.batch_size, self.num_samples, self.tau, gumbels,
                               self._next_seed(), want_noise=want_noise)
