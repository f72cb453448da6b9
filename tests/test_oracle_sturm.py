"""CPU: the root-isolation algorithm of the round-3 five-point kernels (oracle/sturm_ref.py restates it in numpy) against
numpy's companion-matrix eigenvalues -- what the reference itself uses (essential_matrix_estimator_nister.py:361-370) -- on the
degree-10 polynomials of RANSAC-like five-point samples and on polynomials with known roots."""
import numpy as np
import torch

from oracle import cpu_ref as O
from oracle import sturm_ref as S


def _real_roots_numpy(c, lo=-1.0, hi=1.0):
    r = np.roots(c[::-1])
    r = r[np.isfinite(r)]
    real = r[np.abs(r.imag) <= 1e-8 * (1 + np.abs(r.real))].real
    return np.sort(real[(real > lo) & (real <= hi)])


def _check(c):
    iso, evals = S.isolate(c)
    found = missed = phantom = 0
    for i in range(c.shape[0]):
        real = _real_roots_numpy(c[i])
        used = np.zeros(len(real), bool)
        for (l, h) in iso[i]:
            inside = np.nonzero((real > l - 1e-12) & (real <= h + 1e-12) & ~used)[0]
            if len(inside):
                used[inside[0]] = True
            else:
                phantom += 1
        found += int(used.sum())
        missed += int((~used).sum())
    return found, missed, phantom, evals


def test_sturm_isolation_on_five_point_polynomials():
    from differentiable_ransac_amd import synth
    rng = np.random.default_rng(3)
    pairs = [synth.two_view_pair(500 + i, 400, inlier_ratio=0.5, noise=1e-3, dtype=torch.float64) for i in range(4)]
    smp = torch.stack([pairs[i % 4]["matches"][rng.choice(400, 5, replace=False)] for i in range(1536)])
    s = O.nister_poly_system(smp)
    cs = s["cs"].numpy()[s["ok"].numpy()]
    for c in (cs, cs[:, ::-1].copy()):            # |z| <= 1, and |z| > 1 through the reversed polynomial
        found, missed, phantom, evals = _check(c)
        assert found > 2000 and missed <= 1e-3 * found + 1 and phantom <= 1e-3 * found + 1, (found, missed, phantom)
        assert evals.mean() < 6 and evals.max() <= 64     # 4.4 evaluations of the chain per polynomial on such samples


def test_sturm_isolation_on_polynomials_with_known_roots():
    rng = np.random.default_rng(4)
    coefs, truth = [], []
    for _ in range(300):
        nreal = int(rng.choice([0, 2, 4, 6, 8, 10]))
        real = np.sort(rng.uniform(-0.95, 0.95, nreal))
        while nreal >= 2 and np.diff(real).min() < 2e-2:
            real = np.sort(rng.uniform(-0.95, 0.95, nreal))
        c = np.array([1.0])
        for r in real:
            c = np.convolve(c, [-r, 1.0])
        for _ in range((10 - nreal) // 2):
            a, b = rng.uniform(-2, 2), rng.uniform(0.3, 2.0)
            c = np.convolve(c, [a * a + b * b, -2 * a, 1.0])
        coefs.append(c * rng.uniform(0.5, 2) * rng.choice([-1, 1]))
        truth.append(real)
    iso, _ = S.isolate(np.stack(coefs))
    for ivs, real in zip(iso, truth):
        assert len(ivs) == len(real)
        for (l, h), r in zip(ivs, real):
            assert l < r <= h
    # degenerate inputs produce no interval and no exception
    deg = np.zeros((3, 11))
    deg[1, :] = np.nan
    deg[2, 0] = 1.0
    iso, _ = S.isolate(deg)
    assert iso == [[], [], []]
