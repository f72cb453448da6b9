"""Deterministic synthetic correspondences (SURVEY.md 8(d) recipe).

Pure torch-CPU generators (then `.to(device)`), so the same seeded tensors are fed to the
HIP path, to the oracle and to the golden-vector generator.
"""
from __future__ import annotations

import math
from typing import Dict

import torch


def _rodrigues(axis: torch.Tensor, angle: float) -> torch.Tensor:
    a = axis / axis.norm()
    K = torch.tensor([[0.0, -a[2], a[1]], [a[2], 0.0, -a[0]], [-a[1], a[0], 0.0]], dtype=axis.dtype)
    return torch.eye(3, dtype=axis.dtype) + math.sin(angle) * K + (1 - math.cos(angle)) * (K @ K)


def two_view_pair(seed: int, n_points: int = 2000, inlier_ratio: float = 0.5, noise: float = 1e-3,
                  dtype=torch.float32, pixel: bool = False) -> Dict[str, torch.Tensor]:
    """One synthetic image pair.

    Returns matches [N,4] (x1,y1,x2,y2; normalised camera coordinates, or pixels when
    `pixel`), logits [N], gt_E [3,3] (unit Frobenius norm; satisfies x2^T E x1 = 0),
    gt_F [3,3], K1, K2, inlier mask [N], the ground-truth pose R [3,3], t [3] (x2 ~ R x1 + t).  The first floor(N*(1-rho)) points are outliers.
    """
    g = torch.Generator().manual_seed(int(seed))
    f64 = torch.float64
    axis = torch.randn(3, generator=g, dtype=f64)
    R = _rodrigues(axis, 0.3)
    t = torch.randn(3, generator=g, dtype=f64)
    t = t / t.norm()
    X = torch.rand(n_points, 3, generator=g, dtype=f64) * 2 - 1
    X[:, 2] += 4.0
    X2 = X @ R.T + t
    x1 = X[:, :2] / X[:, 2:3]
    x2 = X2[:, :2] / X2[:, 2:3]
    x1 = x1 + noise * torch.randn(n_points, 2, generator=g, dtype=f64)
    x2 = x2 + noise * torch.randn(n_points, 2, generator=g, dtype=f64)
    n_out = int(math.floor(n_points * (1 - inlier_ratio)))
    x2[:n_out] = torch.rand(n_out, 2, generator=g, dtype=f64) * 0.5 - 0.25
    inl = torch.zeros(n_points, dtype=torch.bool)
    inl[n_out:] = True
    logits = torch.randn(n_points, generator=g, dtype=f64) + 3.0 * inl.to(f64)
    tx = torch.tensor([[0.0, -t[2], t[1]], [t[2], 0.0, -t[0]], [-t[1], t[0], 0.0]], dtype=f64)
    E = tx @ R
    E = E / E.norm()
    K = torch.tensor([[1000.0, 0.0, 500.0], [0.0, 1000.0, 500.0], [0.0, 0.0, 1.0]], dtype=f64)
    Kinv = torch.linalg.inv(K)
    F = Kinv.T @ E @ Kinv
    F = F / F.norm()
    matches = torch.cat((x1, x2), dim=1)
    if pixel:
        one = torch.ones(n_points, 1, dtype=f64)
        p1 = torch.cat((x1, one), 1) @ K.T
        p2 = torch.cat((x2, one), 1) @ K.T
        matches = torch.cat((p1[:, :2], p2[:, :2]), dim=1)
    return dict(matches=matches.to(dtype), logits=logits.to(dtype), gt_E=E.to(dtype), gt_F=F.to(dtype),
                K1=K.to(dtype), K2=K.clone().to(dtype), inliers=inl, R=R.to(dtype), t=t.to(dtype))


def rigid_pair(seed: int, n_points: int = 50000, inlier_ratio: float = 0.5, noise: float = 0.01,
               dtype=torch.float32) -> Dict[str, torch.Tensor]:
    """3-D registration pair: matches [N,6] = (p, q), q = p R^T + t + noise; 50 % of q replaced."""
    g = torch.Generator().manual_seed(int(seed))
    f64 = torch.float64
    axis = torch.randn(3, generator=g, dtype=f64)
    R = _rodrigues(axis, 0.7)
    t = torch.randn(3, generator=g, dtype=f64) * 0.3
    P = torch.rand(n_points, 3, generator=g, dtype=f64)
    Q = P @ R.T + t + noise * torch.randn(n_points, 3, generator=g, dtype=f64)
    n_out = int(math.floor(n_points * (1 - inlier_ratio)))
    Q[:n_out] = torch.rand(n_out, 3, generator=g, dtype=f64)
    inl = torch.zeros(n_points, dtype=torch.bool)
    inl[n_out:] = True
    logits = torch.randn(n_points, generator=g, dtype=f64) + 3.0 * inl.to(f64)
    T = torch.eye(4, dtype=f64)
    T[:3, :3] = R
    T[:3, 3] = t
    return dict(matches=torch.cat((P, Q), 1).to(dtype), logits=logits.to(dtype), gt_T=T.to(dtype), inliers=inl)


def batch_two_view(pairs: int, n_points: int, seed0: int = 0, dtype=torch.float32, pixel: bool = False,
                   inlier_ratio: float = 0.5) -> Dict[str, torch.Tensor]:
    """Stack `pairs` synthetic pairs (seeds seed0 .. seed0+pairs-1) on a leading dimension."""
    items = [two_view_pair(seed0 + p, n_points, inlier_ratio, dtype=dtype, pixel=pixel) for p in range(pairs)]
    return {k: torch.stack([it[k] for it in items]) for k in items[0]}


def gumbel_noise(shape, seed: int, dtype=torch.float32) -> torch.Tensor:
    """Explicit Gumbel(0,1) noise, replaying torch.distributions.Gumbel from torch.rand
    (u = tiny + rand*(1-eps-tiny); g = -log(-log u))."""
    g = torch.Generator().manual_seed(int(seed))
    fi = torch.finfo(dtype)
    u = torch.rand(shape, generator=g, dtype=dtype) * ((1 - fi.eps) - fi.tiny) + fi.tiny
    return -torch.log(-torch.log(u))
