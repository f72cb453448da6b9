"""CPU oracle for the differentiable-RANSAC hot path (TEST INFRASTRUCTURE ONLY).

This module is a torch-CPU *restatement* -- written from the maths in SURVEY.md
Appendix B, vectorised over the hypothesis batch, with every source of randomness
taken as an explicit input -- of the reference functions on the hot path:

    K1   GumbelSoftmaxSampler.sample          samplers/gumbel_sampler.py:25-42
    K1u  UniformSampler.batch_generate        samplers/uniform_sampler.py:15-19
    K2   straight-through gather              ransac.py:58-65
    K3n  Nister 5-pt                          estimators/essential_matrix_estimator_nister.py:69-408
    K3s  Stewenius 5-pt                       estimators/essential_matrix_estimator_stewenius.py:20-172
    K3f8 8-pt / LSQ fundamental               estimators/fundamental_matrix_estimator.py:177-260
    K3f7 7-pt fundamental (correct maths; the reference's is degenerate)
    K3r  rigid SVD solver                     estimators/rigid_transformation_SVD_based_solver.py:11-74
    K4   MSAC / Sampson scoring               scorings/msac_score.py:12-55
    K4r  rigid squared residual               estimators/rigid_transformation_SVD_based_solver.py:76-89
    K5   train-mode best-of-S selection       ransac.py:78-108
    K6   test-mode arg-max + adaptive stop    ransac.py:109-144, 202-215
    K7   final refit                          ransac.py:148-195
    L    MatchLoss residual (8(f) rank 2)     loss.py:107-153, cv_utils.py:680-695
    Lp   PoseLoss pose error (8(f) rank 3)    loss.py:11-68, cv_utils.py:48-80,118-189,361-380,503-525
         (Horn decomposition + R/t error pinned by golden vectors; the triangulation inside the cheirality vote is
          cv2.triangulatePoints, absent here: restated from OpenCV's published DLT, PARITY UNPINNED for that primitive)

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.
The product package (differentiable_ransac_amd) never does: its ops fail loudly when
the HIP library is missing.

Pinned against the reference: tests/golden/*.npz were produced by
tests/golden/gen_golden.py, which imports /root/reference (with empty cv2/h5py
stubs) in the build container; tests/test_oracle_golden.py checks every function
here against those vectors.

Unlike the reference the functions return fixed-shape outputs plus validity
masks (the reference drops failed samples and so returns ragged tensors);
`compact_*` helpers produce the reference's shapes.
"""
from __future__ import annotations

import math
from typing import Optional

import torch

# --------------------------------------------------------------------------- #
# K1 / K1u / K2 : samplers and gather
# --------------------------------------------------------------------------- #


def gumbel_from_uniform(rand: torch.Tensor) -> torch.Tensor:
    """Replays torch.distributions.Gumbel(0,1).sample() from torch.rand output.

    Gumbel(0,1) is Uniform(tiny, 1-eps) pushed through -log(-log u)
    (gumbel_sampler.py:20-22 builds the distribution; torch/distributions/gumbel.py).
    """
    fi = torch.finfo(rand.dtype)
    u = rand * ((1 - fi.eps) - fi.tiny) + fi.tiny
    return -torch.log(-torch.log(u))


def gumbel_topk(logits: torch.Tensor, gumbels: torch.Tensor, tau: float, k: int):
    """K1, gumbel_sampler.py:25-42 with the noise made explicit.

    logits [N], gumbels [B,N] -> idx [B,k] int64 ascending by point index,
    ret [B,N] (straight-through one-hot), y_soft [B,N].
    """
    g = (logits.unsqueeze(0) + gumbels) / tau
    y_soft = g.softmax(-1)
    top = torch.topk(g, k, dim=-1).indices
    y_hard = torch.zeros_like(g).scatter_(-1, top, 1.0)
    ret = y_hard - y_soft.detach() + y_soft
    idx = torch.sort(top, dim=-1).values
    return idx, ret, y_soft


def uniform_sample(batch_size: int, k: int, num_points: int, generator=None):
    """K1u, uniform_sampler.py:15-19: randint(0, N-1) -- with replacement, and the last
    point is never drawn (torch.randint's high is exclusive)."""
    return torch.randint(0, num_points - 1, (batch_size, k), generator=generator)


def gather_samples(matches: torch.Tensor, ret: torch.Tensor, soft: Optional[torch.Tensor] = None):
    """K2, ransac.py:64-65 (+ :73 for the weighted variant).

    matches [N,c], ret [B,N] -> minimal samples [B,k,c] (= coordinates times the
    straight-through value at the selected entries, ascending point index)."""
    B = ret.shape[0]
    pts = matches.unsqueeze(0) * ret.unsqueeze(-1)
    sel = ret != 0
    out = pts[sel].view(B, -1, matches.shape[-1])
    if soft is None:
        return out
    return out, soft[sel].view(B, -1)


# --------------------------------------------------------------------------- #
# multivariate polynomial bookkeeping (x, y, z), degree <= 3
# --------------------------------------------------------------------------- #

_E1 = [(1, 0, 0), (0, 1, 0), (0, 0, 1), (0, 0, 0)]  # (x, y, z, 1)

# Nister's orders (nister.py:410-430)
_N2 = [(2, 0, 0), (1, 1, 0), (1, 0, 1), (1, 0, 0), (0, 2, 0), (0, 1, 1), (0, 1, 0), (0, 0, 2), (0, 0, 1), (0, 0, 0)]
_N3 = [(3, 0, 0), (0, 3, 0), (2, 1, 0), (1, 2, 0), (2, 0, 1), (2, 0, 0), (0, 2, 1), (0, 2, 0), (1, 1, 1), (1, 1, 0),
       (1, 0, 2), (1, 0, 1), (1, 0, 0), (0, 1, 2), (0, 1, 1), (0, 1, 0), (0, 0, 3), (0, 0, 2), (0, 0, 1), (0, 0, 0)]
# GrevLex orders used by the Stewenius solver (stewenius.py:134-172)
_G2 = [(2, 0, 0), (1, 1, 0), (0, 2, 0), (1, 0, 1), (0, 1, 1), (0, 0, 2), (1, 0, 0), (0, 1, 0), (0, 0, 1), (0, 0, 0)]
_G3 = [(3, 0, 0), (2, 1, 0), (1, 2, 0), (0, 3, 0), (2, 0, 1), (1, 1, 1), (0, 2, 1), (1, 0, 2), (0, 1, 2), (0, 0, 3),
       (2, 0, 0), (1, 1, 0), (0, 2, 0), (1, 0, 1), (0, 1, 1), (0, 0, 2), (1, 0, 0), (0, 1, 0), (0, 0, 1), (0, 0, 0)]


def _mul_table(ea, eb, eo):
    pos = {e: i for i, e in enumerate(eo)}
    return [(i, j, pos[(a[0] + b[0], a[1] + b[1], a[2] + b[2])]) for i, a in enumerate(ea) for j, b in enumerate(eb)]


def _polymul(a, b, table, n_out):
    out = torch.zeros(a.shape[:-1] + (n_out,), dtype=a.dtype)
    for i, j, o in table:
        out[..., o] = out[..., o] + a[..., i] * b[..., j]
    return out


_T_N11 = _mul_table(_E1, _E1, _N2)
_T_N21 = _mul_table(_N2, _E1, _N3)
_T_G11 = _mul_table(_E1, _E1, _G2)
_T_G21 = _mul_table(_G2, _E1, _G3)


def _epipolar_rows_5pt(pts, weights=None):
    """rows (x1x2, x1y2, x1, y1x2, y1y2, y1, x2, y2, 1)  (nister.py:87-115, stewenius.py:38-42)"""
    x1, y1, x2, y2 = pts[..., 0], pts[..., 1], pts[..., 2], pts[..., 3]
    A = torch.stack((x1 * x2, x1 * y2, x1, y1 * x2, y1 * y2, y1, x2, y2, torch.ones_like(x1)), dim=-1)
    if weights is not None:
        A = weights.unsqueeze(-1) * A
    return A


def _constraints(basis, order2_tab, order3_tab, double_eet: bool):
    """Ten cubic constraints on E(x,y,z) = x*B0 + y*B1 + z*B2 + B3.

    basis [B,3,3,4]: entry polynomial (i,j) in (x,y,z,1).  Rows 0-8: entries of
    EE^T E - 1/2 tr(EE^T) E (row-major i,j); row 9: det E.
    """
    Bn = basis.shape[0]
    e = lambda i, j: basis[:, i, j]
    m11 = lambda a, b: _polymul(a, b, order2_tab, 10)
    m21 = lambda a, b: _polymul(a, b, order3_tab, 20)
    eet = [[m11(e(i, 0), e(j, 0)) + m11(e(i, 1), e(j, 1)) + m11(e(i, 2), e(j, 2)) for j in range(3)] for i in range(3)]
    if double_eet:
        eet = [[2 * v for v in r] for r in eet]
    tr = eet[0][0] + eet[1][1] + eet[2][2]
    rows = []
    for i in range(3):
        for j in range(3):
            rows.append(m21(eet[i][0], e(0, j)) + m21(eet[i][1], e(1, j)) + m21(eet[i][2], e(2, j))
                        - 0.5 * m21(tr, e(i, j)))
    det = (m21(m11(e(0, 1), e(1, 2)) - m11(e(0, 2), e(1, 1)), e(2, 0))
           + m21(m11(e(0, 2), e(1, 0)) - m11(e(0, 0), e(1, 2)), e(2, 1))
           + m21(m11(e(0, 0), e(1, 1)) - m11(e(0, 1), e(1, 0)), e(2, 2)))
    rows.append(det)
    return torch.stack(rows, dim=1)  # [B,10,20]


def _pmul_z(a, b):
    """product of two univariate polynomials, coefficients ascending, batched."""
    na, nb = a.shape[-1], b.shape[-1]
    out = torch.zeros(a.shape[:-1] + (na + nb - 1,), dtype=a.dtype)
    for i in range(na):
        out[..., i:i + nb] = out[..., i:i + nb] + a[..., i:i + 1] * b
    return out


# --------------------------------------------------------------------------- #
# K3n : Nister five-point
# --------------------------------------------------------------------------- #


def nister_poly_system(pts, weights=None):
    """First half of nister.py:69-348: null space -> 10x20 -> Gauss-Jordan -> B(z) -> cs.

    Returns dict with null [B,4,9] (rows N0..N3), coeffs [B,10,20], ok [B] (rank test
    :154-157), Bz [B,3,13] (the `A` of :165-176, highest degree first per column
    block), cs [B,11] ascending."""
    Bn = pts.shape[0]
    A = _epipolar_rows_5pt(pts, weights)
    _, _, vh = torch.linalg.svd(A.transpose(-1, -2) @ A)
    null = vh[:, -4:, :]  # [B,4,9]
    # entry polynomial (i,j) = coefficients at flat index 3*j+i  (:123)
    basis = null.transpose(-1, -2).reshape(Bn, 3, 3, 4).transpose(1, 2)
    rows = _constraints(basis, _T_N11, _T_N21, double_eet=False)
    coeffs = rows  # rows 0..8 trace rows (:145-152), row 9 det (:126-128)
    left = coeffs[:, :, :10]
    rk_left = torch.linalg.matrix_rank(left)
    rk_all = torch.linalg.matrix_rank(coeffs)
    ok = rk_left >= torch.maximum(rk_all, torch.full_like(rk_left, 10))
    safe_left = torch.where(ok[:, None, None], left, torch.eye(10, dtype=pts.dtype).expand_as(left))
    elim = torch.linalg.solve(safe_left, coeffs[:, :, 10:])  # [B,10,10]
    # rows e..j = 4..9 ; k = e - z f ; l = g - z h ; m = i - z j     (:165-176)
    Bz = torch.zeros(Bn, 3, 13, dtype=pts.dtype)
    for r in range(3):
        hi, lo = elim[:, 4 + 2 * r], elim[:, 5 + 2 * r]
        Bz[:, r, 1:4] = hi[:, 0:3]
        Bz[:, r, 0:3] -= lo[:, 0:3]
        Bz[:, r, 5:8] = hi[:, 3:6]
        Bz[:, r, 4:7] -= lo[:, 3:6]
        Bz[:, r, 9:13] = hi[:, 6:10]
        Bz[:, r, 8:12] -= lo[:, 6:10]
    # det B(z) with columns of degree 3,3,4 ; ascending coefficients
    c0 = Bz[:, :, 0:4].flip(-1)
    c1 = Bz[:, :, 4:8].flip(-1)
    c2 = Bz[:, :, 8:13].flip(-1)
    minor = lambda a, b: _pmul_z(c0[:, a], c1[:, b]) - _pmul_z(c0[:, b], c1[:, a])  # rows a,b of cols 0,1
    cs = (_pmul_z(minor(1, 2), c2[:, 0]) - _pmul_z(minor(0, 2), c2[:, 1]) + _pmul_z(minor(0, 1), c2[:, 2]))
    return dict(null=null, coeffs=coeffs, ok=ok, Bz=Bz, cs=cs)


def _polyval_desc(c, z):
    """Horner, c [..., d+1] highest first, z [..., R] -> [..., R]"""
    out = torch.zeros_like(z) + c[..., 0:1]
    for i in range(1, c.shape[-1]):
        out = out * z + c[..., i:i + 1]
    return out


def nister_5pt(pts, weights=None, real_tol: float = 1e-8):
    """K3n, nister.py:69-408.  pts [B,5,4] (or [B,n>5,4]: the non-minimal fallback of
    :64-65 runs the same code on all points).

    Returns E [B,10,3,3], sample_ok [B] (survived the rank / NaN filters :154-157,
    :365-366), is_real [B,10] (root had |imag| <= real_tol*(1+|root|); the reference keeps
    the real part of complex roots too -- Q10 -- those slots are "don't care")."""
    s = nister_poly_system(pts, weights)
    Bn = pts.shape[0]
    cs, Bz, null = s["cs"], s["Bz"], s["null"]
    comp = torch.zeros(Bn, 10, 10, dtype=pts.dtype)
    comp[:, :-1, 1:] = torch.eye(9, dtype=pts.dtype)
    comp[:, -1, :] = -cs[:, :-1] / cs[:, -1:]
    finite = torch.isfinite(comp).all(-1).all(-1)
    ok = s["ok"] & finite
    comp = torch.where(ok[:, None, None], comp, torch.eye(10, dtype=pts.dtype).expand_as(comp))
    ev = torch.linalg.eigvals(comp)
    roots = ev.real
    is_real = ev.imag.abs() <= real_tol * (1 + ev.real.abs())
    # B(z) (x, y, 1)^T = 0 -> [x,y] from rows 0-1, LSQ over 3 rows if row 2 disagrees (:379-392)
    bx = torch.stack([_polyval_desc(Bz[:, r, 0:4], roots) for r in range(3)], dim=-1)  # [B,10,3]
    by = torch.stack([_polyval_desc(Bz[:, r, 4:8], roots) for r in range(3)], dim=-1)
    b1 = torch.stack([_polyval_desc(Bz[:, r, 8:13], roots) for r in range(3)], dim=-1)
    M = torch.stack((bx, by), dim=-1)  # [B,10,3,2]
    rhs = b1.unsqueeze(-1)  # [B,10,3,1]
    M2 = M[:, :, :2, :]
    det2 = M2[..., 0, 0] * M2[..., 1, 1] - M2[..., 0, 1] * M2[..., 1, 0]
    inv2 = torch.stack((torch.stack((M2[..., 1, 1], -M2[..., 0, 1]), -1),
                        torch.stack((-M2[..., 1, 0], M2[..., 0, 0]), -1)), -2) / det2[..., None, None]
    xz = inv2 @ rhs[:, :, :2]
    bad = ((M[:, :, 2:3] @ xz - rhs[:, :, 2:3]).abs() > 1e-3).flatten(1)
    if bad.any():
        q, r = torch.linalg.qr(M[bad])
        xz[bad] = torch.linalg.solve(r, q.transpose(-1, -2) @ rhs[bad])
    x, y = -xz[..., 0, 0], -xz[..., 1, 0]
    Ef = (x[..., None] * null[:, None, 0] + y[..., None] * null[:, None, 1]
          + roots[..., None] * null[:, None, 2] + null[:, None, 3])
    Ef = Ef / torch.sqrt(x * x + y * y + roots * roots + 1.0)[..., None]
    E = Ef.view(Bn, 10, 3, 3).transpose(-1, -2)
    return E, ok, is_real


def compact_models(E, sample_ok):
    """Reference output shape: drop failed samples, flatten (nister.py:404-407)."""
    out = E[sample_ok].reshape(-1, 3, 3)
    if out.shape[0] == 0:
        return torch.eye(3, dtype=E.dtype).unsqueeze(0)
    return out


# --------------------------------------------------------------------------- #
# K3s : Stewenius five-point
# --------------------------------------------------------------------------- #


def stewenius_5pt(pts, real_tol: float = 1e-8):
    """K3s, stewenius.py:20-80.  Returns E [B,10,3,3] (LAPACK eigenvector scale, no
    normalisation), is_real [B,10], lam [B,10] complex eigenvalues."""
    Bn = pts.shape[0]
    A = _epipolar_rows_5pt(pts)
    _, _, vh = torch.linalg.svd(A)  # full: [B,9,9]
    null = vh[:, -4:, :].transpose(-1, -2)  # [B,9,4]
    basis = null.reshape(Bn, 3, 3, 4).transpose(1, 2)  # (:53)
    C = _constraints(basis, _T_G11, _T_G21, double_eet=True)
    # reference scales the trace rows by 2 (2EE^T E - tr(EE^T)E with 0.5 applied to tr(2EE^T)); det row unscaled:
    # _constraints(double_eet=True) doubled EE^T inside the det-free rows only.
    G = torch.linalg.solve(C[:, :, :10], C[:, :, 10:])
    act = torch.zeros(Bn, 10, 10, dtype=pts.dtype)
    act[:, 0:3] = G[:, 0:3]
    act[:, 3] = G[:, 4]
    act[:, 4] = G[:, 5]
    act[:, 5] = G[:, 7]
    act[:, 6, 0] = -1
    act[:, 7, 1] = -1
    act[:, 8, 3] = -1
    act[:, 9, 6] = -1
    lam, vec = torch.linalg.eig(act)
    is_real = lam.imag.abs() <= real_tol * (1 + lam.real.abs())
    Ef = null @ vec.real[:, -4:]  # [B,9,10]
    E = Ef.transpose(-1, -2).reshape(Bn, 10, 3, 3).transpose(-1, -2)
    return E, is_real, lam


# --------------------------------------------------------------------------- #
# K3f8 / K3f7 : fundamental matrix
# --------------------------------------------------------------------------- #


def hartley_normalize(m):
    """fundamental_matrix_estimator.py:177-217.  m [B,n,4] -> normalised, T1, T2^T."""
    mass = m.mean(dim=1)
    c = m - mass.unsqueeze(1)
    d1 = torch.linalg.norm(c[:, :, :2], dim=2).mean(dim=1)
    d2 = torch.linalg.norm(c[:, :, 2:], dim=2).mean(dim=1)
    r1 = math.sqrt(2) / d1
    r2 = math.sqrt(2) / d2
    n = torch.cat((c[:, :, :2] * r1[:, None, None], c[:, :, 2:] * r2[:, None, None]), dim=2)
    T1 = torch.zeros(m.shape[0], 3, 3, dtype=m.dtype)
    T2t = torch.zeros(m.shape[0], 3, 3, dtype=m.dtype)
    T1[:, 0, 0] = T1[:, 1, 1] = r1
    T2t[:, 0, 0] = T2t[:, 1, 1] = r2
    T1[:, 2, 2] = T2t[:, 2, 2] = 1
    T1[:, 0, 2] = -r1 * mass[:, 0]
    T1[:, 1, 2] = -r1 * mass[:, 1]
    T2t[:, 2, 0] = -r2 * mass[:, 2]
    T2t[:, 2, 1] = -r2 * mass[:, 3]
    return n, T1, T2t


def _f_rows(pts, weights=None):
    """rows (x1x2, x2y1, x2, y2x1, y2y1, y2, x1, y1, 1)  (fundamental…:243-246)"""
    x1, y1, x2, y2 = pts[..., 0], pts[..., 1], pts[..., 2], pts[..., 3]
    A = torch.stack((x1 * x2, x2 * y1, x2, y2 * x1, y2 * y1, y2, x1, y1, torch.ones_like(x1)), dim=-1)
    if weights is not None:
        A = weights.unsqueeze(-1) * A
    return A


def fundamental_8pt(pts, weights=None):
    """K3f8, fundamental…:172-174 + :230-260.  pts [B,n>=8,4] -> F [B,3,3]
    (un-normalised, full rank, sign = LAPACK's)."""
    n, T1, T2t = hartley_normalize(pts)
    A = _f_rows(n, weights)
    _, _, vh = torch.linalg.svd(A.transpose(-1, -2) @ A)
    F = vh[:, -1, :].reshape(-1, 3, 3)
    return T2t @ F @ T1


def fundamental_7pt(pts):
    """K3f7 with the CORRECT maths (the reference's two versions are
    numerically wrong, Q7/Q8, so there is no reference vector for this function).

    pts [B,7,4] -> F [B,4,3,3] (unit Frobenius norm; unused slots = eye(3)), valid [B,4].
    Real roots ascending in slots 0.., slot 3 is always unused (a cubic has <= 3 roots;
    the 4-slot contract is FundamentalMatrixEstimatorNew's, fundamental…:303-308)."""
    Bn = pts.shape[0]
    A = _f_rows(pts)
    _, _, vh = torch.linalg.svd(A)
    F1 = vh[:, -1, :].reshape(Bn, 3, 3)
    F2 = vh[:, -2, :].reshape(Bn, 3, 3)
    p = lambda a: torch.linalg.det(a * F1 + (1 - a) * F2)
    p0, p1, pm1, p2, pm2 = p(0.0), p(1.0), p(-1.0), p(2.0), p(-2.0)
    c0 = p0
    c2 = (p1 + pm1) / 2 - p0
    c1 = 2 * (p1 - pm1) / 3 - (p2 - pm2) / 12
    c3 = (p2 - pm2) / 12 - (p1 - pm1) / 6
    comp = torch.zeros(Bn, 3, 3, dtype=pts.dtype)
    comp[:, 1, 0] = comp[:, 2, 1] = 1
    comp[:, 0, 2] = -c0 / c3
    comp[:, 1, 2] = -c1 / c3
    comp[:, 2, 2] = -c2 / c3
    ev = torch.linalg.eigvals(comp)
    real = ev.imag.abs() <= 1e-9 * (1 + ev.real.abs())
    lam = torch.where(real, ev.real, torch.full_like(ev.real, float("inf")))
    lam, order = torch.sort(lam, dim=-1)
    real = torch.gather(real, 1, order)
    F = lam[:, :, None, None].nan_to_num(posinf=0.0) * F1[:, None] + (1 - lam.nan_to_num(posinf=0.0))[:, :, None, None] * F2[:, None]
    F = F / torch.linalg.norm(F, dim=(-1, -2), keepdim=True)
    eye = torch.eye(3, dtype=pts.dtype)
    F = torch.where(real[:, :, None, None], F, eye.expand_as(F))
    F = torch.cat((F, eye.expand(Bn, 1, 3, 3)), dim=1)
    valid = torch.cat((real, torch.zeros(Bn, 1, dtype=torch.bool)), dim=1)
    return F, valid


# --------------------------------------------------------------------------- #
# K3r / K4r : rigid transformation
# --------------------------------------------------------------------------- #


def rigid_svd(data, weights=None, flag: bool = True):
    """K3r, rigid…:11-74.  data [B,n>=3,6] -> model [B,4,4], R, t, scale, ok [B].

    flag=True (reference default) decomposes cov^T cov, so R ~ I (Q9); t is the
    row-sum formula of :66."""
    n = data.shape[1]
    c = data.mean(dim=1)
    d = data - c[:, None, :]
    a0 = torch.sqrt((d[:, :, 0:3] ** 2).sum(-1)).sum(-1) / n
    a1 = torch.sqrt((d[:, :, 3:6] ** 2).sum(-1)).sum(-1) / n
    dt = d.transpose(-1, -2)
    if weights is not None:
        dt = dt * weights
    s3 = torch.sqrt(torch.tensor(3.0))
    X0 = dt[:, 0:3, :] * (s3 / a0)[:, None, None]
    X1 = dt[:, 3:6, :] * (s3 / a1)[:, None, None]
    cov = X0 @ X1.transpose(-1, -2)
    ok = ~torch.isnan(cov).any(-1).any(-1)
    cov_s = torch.where(ok[:, None, None], cov, torch.eye(3, dtype=cov.dtype).expand_as(cov))
    tgt = cov_s.transpose(-1, -2) @ cov_s if flag else cov_s.transpose(-1, -2)
    u, _, vh = torch.linalg.svd(tgt)
    v = vh.transpose(-1, -2).clone()
    R = v @ u.transpose(-1, -2)
    neg = torch.linalg.det(R) < 0
    v[neg, :, 2] = -v[neg, :, 2]
    R = v @ u.transpose(-1, -2)
    scale = a1 / a0
    t = (R * (-c[:, None, 0:3])).sum(dim=1) + c[:, 3:6]
    model = torch.zeros(data.shape[0], 4, 4, dtype=R.dtype)
    model[:, :3, :3] = R
    model[:, :3, 3] = t
    model[:, 3, 3] = 1
    return model, R, t, scale, ok


def rigid_squared_residual(pts1, pts2, descriptor, threshold: float = 0.03):
    """K4r, rigid…:76-89.  descriptor [B,4,3] = model[:, :3, :]^T.
    -> (sum_n d2 [B], mean d2 scalar, mask [B,N])"""
    h = torch.cat((pts1, torch.ones(pts1.shape[0], 1, dtype=pts1.dtype)), dim=1)
    t = h @ descriptor
    d2 = ((pts2[None] - t) ** 2).sum(-1)
    return d2.sum(-1), d2.mean(), d2 < threshold


# --------------------------------------------------------------------------- #
# K4 : MSAC
# --------------------------------------------------------------------------- #


def msac_score(matches, models, threshold: float = 0.75, chunk: int = 0):
    """K4, msac_score.py:12-55.  matches [N,4], models [M,3,3] -> scores [M], masks [M,N] bool."""
    if chunk and models.shape[0] > chunk:
        parts = [msac_score(matches, models[i:i + chunk], threshold) for i in range(0, models.shape[0], chunk)]
        return torch.cat([p[0] for p in parts]), torch.cat([p[1] for p in parts])
    thr2 = (3 / 2 * threshold) ** 2
    n = matches.shape[0]
    one = torch.ones(n, 1, dtype=matches.dtype)
    h1 = torch.cat((matches[:, 0:2], one), dim=-1)
    h2 = torch.cat((matches[:, 2:4], one), dim=-1)
    Mx1 = models @ h1.T  # [M,3,N]
    Mtx2 = models.transpose(-1, -2) @ h2.T
    jj = Mx1[:, 0] ** 2 + Mx1[:, 1] ** 2 + Mtx2[:, 0] ** 2 + Mtx2[:, 1] ** 2
    r = (h1.T.unsqueeze(0) * Mtx2).sum(-2)
    d2 = r.square() / jj
    masks = d2 < thr2
    scores = torch.clamp(1 - d2 / thr2, min=0.0).sum(-1)
    return scores, masks


# --------------------------------------------------------------------------- #
# K5 / K6 / K7 and the drivers
# --------------------------------------------------------------------------- #


def select_closest(models, gt, S: int):
    """K5, ransac.py:87-96.  models [B,S,3,3], gt [3,3] -> chosen [B,3,3], which [B]."""
    dist = torch.linalg.norm(models - gt, dim=(-2, -1))
    which = dist.argmin(dim=-1)
    chosen = models[torch.arange(models.shape[0]), which]
    return chosen, which


def adaptive_iteration_number(inliers: int, n_points: int, sample_size: int, confidence: float = 0.999,
                              eps: float = 1e-5, max_iterations: int = 5000) -> float:
    """K6, ransac.py:202-215."""
    ratio = inliers / n_points
    prob = 1.0 - ratio ** sample_size
    if prob >= 1.0 - eps:
        return max_iterations
    return max(0.0, math.log10(1.0 - confidence) / math.log10(1 - ratio ** sample_size + eps))


def normalized_threshold(threshold, K1, K2, fmat: bool):
    """ransac.py:49-53 (sic: K1[0,0] twice, Q3)."""
    if fmat:
        return threshold
    return threshold / float((K1[0, 0] + K1[1, 1] + K1[0, 0] + K2[1, 1]) / 4)


def ransac_train_batch(matches, logits, gumbels, gt, solver: str, tau: float = 1.0, weighted: bool = False):
    """One train-mode batch of RANSAC.__call__ (ransac.py:55-108) for Gumbel samplers.

    solver in {"nister", "f8"}.  Returns chosen models [B',3,3] (NaN-filtered), idx [B,k]."""
    k = {"nister": 5, "f8": 8}[solver]
    idx, ret, soft = gumbel_topk(logits, gumbels, tau, k)
    if weighted:
        samples, w = gather_samples(matches, ret, soft)
    else:
        samples, w = gather_samples(matches, ret), None
    if solver == "nister":
        E, ok, _ = nister_5pt(samples, w)
        chosen, _ = select_closest(E[ok], gt, 10)
    else:
        chosen = fundamental_8pt(samples, w)
    keep = ~torch.isnan(chosen).any(-1).any(-1)
    return chosen[keep], idx


def ransac_test(matches, logits, gumbel_batches, K1, K2, solver: str, threshold: float = 0.75,
                max_iterations: int = 5000, confidence: float = 0.999, tau: float = 1.0, refit: bool = True,
                weighted: bool = False):
    """Test-mode RANSAC.__call__ (ransac.py:55-200, lo=0) with explicit noise per batch.

    gumbel_batches: list of [B,N] noise tensors, consumed one per iteration; the loop stops
    at the adaptive bound or when the list is exhausted.
    Returns best_model [3,3], best_mask [N], best_score, iterations."""
    fmat = solver == "f8"
    k = 8 if fmat else 5
    thr = normalized_threshold(threshold, K1, K2, fmat)
    N = matches.shape[0]
    it, best_score, best_mask, best_model = 0, 0.0, None, None
    max_iters = max_iterations
    for g in gumbel_batches:
        if it >= max_iters:
            break
        B = g.shape[0]
        idx, ret, y_soft = gumbel_topk(logits, g, tau, k)
        if weighted:                          # ransac.py:70-74: the soft weights of the selected points scale the rows
            samples, wts = gather_samples(matches, ret, y_soft)
        else:
            samples, wts = gather_samples(matches, ret), None
        if fmat:
            models = fundamental_8pt(samples, wts)
        elif weighted:
            E, ok, _ = nister_5pt(samples, wts)
            models = compact_models(E, ok)
        else:
            E, ok, _ = nister_5pt(samples)
            models = compact_models(E, ok)
        scores, masks = msac_score(matches, models, thr)
        b = int(torch.argmax(scores))
        if float(scores[b]) > best_score or it == 0:
            best_score, best_mask, best_model = float(scores[b]), masks[b], models[b]
            max_iters = min(max_iterations, adaptive_iteration_number(int(best_mask.sum()), N, k, confidence,
                                                                      max_iterations=max_iterations))
        it += B
    if refit:
        inl = best_mask.nonzero(as_tuple=True)[0]
        if fmat and weighted:
            # ransac.py:151-153: `soft_weights[0, inlier_indices[0]]` -- hypothesis 0 of the LAST batch sampled
            cand = fundamental_8pt(matches[inl].unsqueeze(0), y_soft[0, inl].unsqueeze(0))
        elif fmat:
            cand = fundamental_8pt(matches[inl].unsqueeze(0))
        else:
            # pymagsac absent => Nister on ALL points in f64 as one sample (ransac.py:157-165, nister.py:64-65)
            E, ok, _ = nister_5pt(matches.unsqueeze(0).double())
            cand = compact_models(E, ok).to(matches.dtype)
        scores, _ = msac_score(matches, cand, thr)
        if float(scores.max()) > best_score:
            b = int(torch.argmax(scores))
            best_model, best_score = cand[b], float(scores[b])
    return best_model, best_mask, best_score, it


def ransac3d_train_batch(matches, logits, gumbels, tau: float = 1.0, flag: bool = True):
    """One train-mode batch of RANSAC3D.__call__ (ransac.py:355-382)."""
    idx, ret, _ = gumbel_topk(logits, gumbels, tau, 3)
    samples = gather_samples(matches, ret)
    model, R, t, scale, ok = rigid_svd(samples, flag=flag)
    res, mean_res, mask = rigid_squared_residual(matches[:, :3], matches[:, 3:], model[:, :3, :].transpose(-1, -2))
    return model[ok], res, mean_res, mask, idx


def episym(x1, x2, F):
    """batch_episym, cv_utils.py:680-695: x1, x2 [n,2], F [M,3,3] -> ys [M,n] (symmetric epipolar error)."""
    one = torch.ones(x1.shape[0], 1, dtype=x1.dtype)
    h1, h2 = torch.cat((x1, one), 1), torch.cat((x2, one), 1)
    Fx1 = torch.einsum("mij,nj->mni", F, h1)
    Ftx2 = torch.einsum("mji,nj->mni", F, h2)
    r = (h2[None] * Fx1).sum(-1)
    return r ** 2 * (1.0 / (Fx1[..., 0] ** 2 + Fx1[..., 1] ** 2 + 1e-15) + 1.0 / (Ftx2[..., 0] ** 2 + Ftx2[..., 1] ** 2 + 1e-15))


def match_loss(models, matches, gt_mask):
    """MatchLoss.forward for one pair given the GT-inlier mask (loss.py:137-153): mean of min(ys, 1)."""
    ys = episym(matches[gt_mask, :2], matches[gt_mask, 2:], models)
    return torch.clamp(ys, max=1.0).mean()


# --------------------------------------------------------------------------- #
# pose error of an essential matrix (PoseLoss, loss.py:11-68)
# --------------------------------------------------------------------------- #


def cofactor3(E):
    """Cofactor matrix of [...,3,3] by cross products of the rows.  The reference computes inv(E).T * det(E)
    (matrix_cofactor_tensor, cv_utils.py:163-175), which is the same matrix for det != 0 and raises for det == 0."""
    r0, r1, r2 = E[..., 0, :], E[..., 1, :], E[..., 2, :]
    return torch.stack((torch.linalg.cross(r1, r2), torch.linalg.cross(r2, r0), torch.linalg.cross(r0, r1)), dim=-2)


def _skew(b):
    z = torch.zeros_like(b[..., 0])
    return torch.stack((torch.stack((z, -b[..., 2], b[..., 1]), -1), torch.stack((b[..., 2], z, -b[..., 0]), -1),
                        torch.stack((-b[..., 1], b[..., 0], z), -1)), -2)


def horn_decompose(E):
    """new_decompose_E, cv_utils.py:118-161 (Horn 1990), batched over [...,3,3]: returns R1, R2 [...,3,3], t [...,3].
    b = sqrt(tr(E E^T)/2) * (largest pairwise cross product of the COLUMNS of E, normalised); R1,2 = (cof(E) -+ [b]x E)
    / (b.b); t = b / |b|.  As in the reference the skew matrix is built from detached values (torch.tensor(...) at
    :144-148), i.e. it is a constant for autograd."""
    e1, e2, e3 = E[..., :, 0], E[..., :, 1], E[..., :, 2]
    crosses = torch.stack((torch.linalg.cross(e1, e2), torch.linalg.cross(e2, e3), torch.linalg.cross(e3, e1)), dim=-2)
    norms = torch.linalg.norm(crosses, dim=-1)
    largest = norms.argmax(dim=-1)
    pick = torch.gather(crosses, -2, largest[..., None, None].expand(largest.shape + (1, 3))).squeeze(-2)
    scale = torch.sqrt(0.5 * (E * E).sum((-1, -2)))
    b1 = scale[..., None] * pick / torch.linalg.norm(pick, dim=-1, keepdim=True)
    B1 = _skew(b1.detach())
    bb = (b1 * b1).sum(-1)[..., None, None]
    cof = cofactor3(E)
    R1 = (cof - B1 @ E) / bb
    R2 = (cof + B1 @ E) / bb
    return R1, R2, b1 / torch.linalg.norm(b1, dim=-1, keepdim=True)


def svd_decompose(E):
    """decompose_E, cv_utils.py:83-116, batched over [...,3,3]: R1 = U_ W V_^T, R2 = U_ W^T V_^T (U_, V_^T negated when
    their determinant is negative), t = U[:, -1]."""
    U, _, Vh = torch.linalg.svd(E)
    W = torch.tensor([[0.0, -1.0, 0.0], [1.0, 0.0, 0.0], [0.0, 0.0, 1.0]], dtype=E.dtype)
    U_ = torch.where((torch.linalg.det(U) < 0)[..., None, None], -U, U)
    Vh_ = torch.where((torch.linalg.det(Vh) < 0)[..., None, None], -Vh, Vh)
    return U_ @ W @ Vh_, U_ @ W.T @ Vh_, U[..., :, -1]


def triangulate_dlt(P0, P1, x1, x2):
    """cv2.triangulatePoints (OpenCV calib3d triangulate.cpp, the call at cv_utils.py:182): per point the 4x4 system
    [x P[2] - P[0]; y P[2] - P[1]] of both cameras, solution = right singular vector of the smallest singular value.
    OpenCV is absent from the build container: PARITY UNPINNED for this primitive (restated from the published
    algorithm); every test the reference applies to the result is invariant to its sign and scale.
    P0, P1 [...,3,4]; x1, x2 [n,2] -> X [...,n,4]."""
    def rows(P, x):
        return torch.stack((x[:, 0, None] * P[..., None, 2, :] - P[..., None, 0, :],
                            x[:, 1, None] * P[..., None, 2, :] - P[..., None, 1, :]), dim=-2)
    A = torch.cat((rows(P0, x1), rows(P1, x2)), dim=-2)          # [..., n, 4, 4]
    return torch.linalg.svd(A)[2][..., 3, :]


def cheirality_votes(R1, R2, t, x1, x2, distance_threshold: float = 50.0):
    """recoverPose + cheirality_check, cv_utils.py:48-80,177-189: number of points in front of both cameras (and
    closer than the threshold) for the candidates (R1,t), (R2,t), (R1,-t), (R2,-t).  R [...,3,3], t [...,3] -> [...,4]."""
    P0 = torch.eye(3, 4, dtype=R1.dtype).expand(R1.shape[:-2] + (3, 4))
    votes = []
    for R, tt in ((R1, t), (R2, t), (R1, -t), (R2, -t)):
        P = torch.cat((R, tt[..., None]), dim=-1)
        Q = triangulate_dlt(P0, P, x1, x2)                       # [..., n, 4]
        Qh = Q / Q[..., 3:4]
        d2 = (P[..., None, 2, :] * Qh).sum(-1)
        m = (Q[..., 2] * Q[..., 3] > 0) & (Qh[..., 2] < distance_threshold) & (d2 > 0) & (d2 < distance_threshold)
        votes.append(m.sum(-1))
    return torch.stack(votes, dim=-1)


def recover_pose_mask(E, matches, distance_threshold: float = 50.0):
    """The inlier mask cv2.recoverPose returns (loss.py:99,134 use it as the ground-truth inlier mask): the points that
    pass cheirality_check (cv_utils.py:177-189) for the winning candidate of recoverPose (cv_utils.py:48-80).
    E [3,3] -> (mask [n] bool, winning candidate)."""
    R1, R2, t = horn_decompose(E[None])
    x1, x2 = matches[:, :2], matches[:, 2:]
    P0 = torch.eye(3, 4, dtype=E.dtype)
    masks = []
    for R, tt in ((R1[0], t[0]), (R2[0], t[0]), (R1[0], -t[0]), (R2[0], -t[0])):
        P = torch.cat((R, tt[:, None]), dim=-1)
        Q = triangulate_dlt(P0, P, x1, x2)
        Qh = Q / Q[:, 3:4]
        d2 = (P[2] * Qh).sum(-1)
        masks.append((Q[:, 2] * Q[:, 3] > 0) & (Qh[:, 2] < distance_threshold) & (d2 > 0) & (d2 < distance_threshold))
    votes = torch.stack([m.sum() for m in masks])
    best = int(votes.argmax())
    return masks[best], best


def rotation_translation_error(R_gt, t_gt, R, t):
    """evaluate_R_t_tensor, cv_utils.py:361-380, batched over the leading dims of R/t: radians."""
    eps = 1e-8
    c = ((R * R_gt).sum((-1, -2)) - 1.0) * 0.5                   # trace(R R_gt^T) = sum(R * R_gt)
    err_q = torch.arccos(torch.clamp(c, -1.0, 1.0))
    tg = t_gt / (torch.linalg.norm(t_gt) + eps)
    loss_t = torch.clamp(1.0 - (t * tg).sum(-1) ** 2, min=eps)
    err_t = torch.arccos(torch.sqrt(1.0 - loss_t + eps))
    return err_q, err_t


def pose_error(E, matches, R_gt, t_gt, distance_threshold: float = 50.0, svd: bool = False):
    """eval_essential_matrix, cv_utils.py:503-525, for models E [M,3,3] of one pair: (err_R, err_t) in
    degrees [M], chosen candidate [M] (first arg-max of the votes, as torch.argmax at cv_utils.py:69).  svd selects
    decompose_E (cv_utils.py:83-116) instead of Horn's new_decompose_E (:118-161)."""
    R1, R2, t = svd_decompose(E) if svd else horn_decompose(E)
    with torch.no_grad():
        votes = cheirality_votes(R1, R2, t, matches[:, :2], matches[:, 2:], distance_threshold)
        which = votes.argmax(dim=-1)
    R = torch.where((which % 2 == 0)[:, None, None], R1, R2)
    tt = torch.where((which < 2)[:, None], t, -t)
    eq, et = rotation_translation_error(R_gt, t_gt, R, tt)
    return eq * (180.0 / math.pi), et * (180.0 / math.pi), which


def pose_loss(models_per_pair, matches, R_gt, t_gt):
    """PoseLoss.forward_average, loss.py:17-68 (essential-matrix branch): mean over pairs of the mean over the pair's
    models of (err_R + err_t) / 2 in degrees."""
    total = 0.0
    for b, E in enumerate(models_per_pair):
        eq, et, _ = pose_error(E, matches[b], R_gt[b], t_gt[b])
        total = total + ((eq + et) / 2).sum() / E.shape[0]
    return total / len(models_per_pair)


# --------------------------------------------------------------------------- #
# canonical forms used by the parity tests
# --------------------------------------------------------------------------- #


def canonical(M):
    """unit Frobenius norm, sign such that the largest-magnitude entry is positive."""
    flat = M.reshape(M.shape[:-2] + (9,))
    flat = flat / torch.linalg.norm(flat, dim=-1, keepdim=True)
    j = flat.abs().argmax(dim=-1, keepdim=True)
    s = torch.sign(torch.gather(flat, -1, j))
    return (flat * s).reshape(M.shape)


def match_solution_sets(A, a_valid, Bm, b_valid):
    """For every valid canonical model in A [S,3,3] the distance to the nearest valid one in Bm.
    Returns tensor [n_valid_A] of max-abs distances (inf if Bm has none)."""
    Ac = canonical(A[a_valid]).reshape(-1, 9)
    Bc = canonical(Bm[b_valid]).reshape(-1, 9)
    if Ac.shape[0] == 0:
        return torch.zeros(0, dtype=A.dtype)
    if Bc.shape[0] == 0:
        return torch.full((Ac.shape[0],), float("inf"), dtype=A.dtype)
    d = (Ac[:, None, :] - Bc[None, :, :]).abs().amax(-1)
    return d.min(dim=1).values
