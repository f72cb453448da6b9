"""The real-root search of the five-point kernels (Sturm-sequence isolation + bracketed refinement; it replaces
torch.linalg.eigvals of the companion matrix, nister.py:361-370, and torch.linalg.eig, stewenius.py:74) on polynomials with
KNOWN roots, through the test hook dr_debug_real_roots10, for both the round-3 method and the derivative chain it replaced."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _poly_from_roots(real, cpairs, lead):
    """ascending coefficients of lead * prod (z - r) * prod ((z - a)^2 + b^2), f64 via exact convolution"""
    c = np.array([1.0])
    for r in real:
        c = np.convolve(c, np.array([-r, 1.0]))
    for a, b in cpairs:
        c = np.convolve(c, np.array([a * a + b * b, -2 * a, 1.0]))
    return lead * c


def _make(rng, n, nreal_choices, spread=2.5, min_sep=1e-2):
    coefs, truth = [], []
    for _ in range(n):
        nreal = int(rng.choice(nreal_choices))
        while True:
            real = np.sort(rng.uniform(-spread, spread, nreal))
            mags = np.abs(np.abs(real) - 1.0)
            if (nreal < 2 or np.diff(real).min() > min_sep) and (nreal == 0 or (mags.min() > 1e-3 and np.abs(real).min() > 1e-3)):
                break
        cp = [(rng.uniform(-2, 2), rng.uniform(0.3, 2.0)) for _ in range((10 - nreal) // 2)]
        c = _poly_from_roots(real, cp, rng.uniform(0.5, 2.0) * rng.choice([-1, 1]))
        assert c.shape[0] == 11
        coefs.append(c)
        truth.append(real)
    return np.stack(coefs), truth


def _found(roots, counts, i):
    return np.sort(np.concatenate([roots[i, 0, : counts[i, 0]], roots[i, 1, : counts[i, 1]]]))


@pytest.mark.parametrize("method", [1, 0])
def test_real_roots_of_polynomials_with_known_roots(dev, method):
    from differentiable_ransac_amd import ops
    rng = np.random.default_rng(5)
    coef, truth = _make(rng, 4096, [0, 2, 2, 4, 4, 6, 8, 10])
    roots, counts = ops.debug_real_roots10(torch.from_numpy(coef).to(dev), method)
    roots, counts = roots.cpu().numpy(), counts.cpu().numpy()
    miss = extra = 0
    worst = 0.0
    for i, real in enumerate(truth):
        got = _found(roots, counts, i)
        if len(got) != len(real):
            miss += max(0, len(real) - len(got))
            extra += max(0, len(got) - len(real))
            continue
        if len(real):
            worst = max(worst, float(np.max(np.abs(got - real) / (1 + np.abs(real)))))
    total = sum(len(t) for t in truth)
    assert extra == 0 and miss <= (0 if method == 1 else 2e-3 * total), (method, miss, extra, total)
    assert worst < 1e-7, worst          # degree-10 coefficients from a product: the roots are conditioned ~1e-10 .. 1e-8
    # halves: |z| <= 1 in slot 0 ascending, |z| > 1 in slot 1
    for i in range(0, 4096, 97):
        a, b = roots[i, 0, : counts[i, 0]], roots[i, 1, : counts[i, 1]]
        assert np.all(np.abs(a) <= 1 + 1e-12) and np.all(np.abs(b) >= 1 - 1e-12) and np.all(np.diff(a) > 0)


def test_sturm_separates_close_root_pairs_and_survives_degenerate_polynomials(dev):
    from differentiable_ransac_amd import ops
    rng = np.random.default_rng(9)
    coefs, truth = [], []
    for sep in (1e-3, 1e-5, 1e-7):
        for _ in range(256):
            x0 = rng.uniform(-0.8, 0.8) if rng.random() < 0.5 else rng.choice([-1, 1]) * rng.uniform(1.3, 2.5)
            others = rng.uniform(-2.5, 2.5, 2)
            while np.min(np.abs(others - x0)) < 0.2 or abs(others[0] - others[1]) < 0.2 or np.min(np.abs(np.abs(others) - 1)) < 1e-2:
                others = rng.uniform(-2.5, 2.5, 2)
            real = np.sort(np.array([x0 - sep / 2, x0 + sep / 2, *others]))
            cp = [(rng.uniform(-2, 2), rng.uniform(0.5, 2.0)) for _ in range(3)]
            coefs.append(_poly_from_roots(real, cp, 1.0))
            truth.append((sep, real))
    coef = np.stack(coefs)
    roots, counts = ops.debug_real_roots10(torch.from_numpy(coef).to(dev), 1)
    roots, counts = roots.cpu().numpy(), counts.cpu().numpy()
    for sep in (1e-3, 1e-5):
        ok = 0
        rows = [i for i, t in enumerate(truth) if t[0] == sep]
        for i in rows:
            got = _found(roots, counts, i)
            ok += len(got) == 4 and np.max(np.abs(got - truth[i][1])) < 1e-2 * sep + 1e-9
        # a pair 1e-5 apart sits at the edge of what f64 coefficients of a degree-10 product resolve: most, not all
        assert ok >= (0.99 if sep == 1e-3 else 0.9) * len(rows), (sep, ok, len(rows))
    for i, t in enumerate(truth):           # never a phantom: every reported root is a root
        for r in _found(roots, counts, i):
            assert abs(np.polyval(coef[i][::-1], r)) <= 1e-6 * np.sum(np.abs(coef[i]) * np.abs(r) ** np.arange(11)) + 1e-12
    # degenerate inputs: zero polynomial, NaN, vanishing leading coefficient (degree 9), constant, even / odd symmetric
    deg = np.zeros((8, 11))
    deg[1, :] = np.nan
    deg[2] = np.concatenate([_poly_from_roots([-0.5, 0.25, 2.0], [(0.3, 1.0)] * 3, 1.0), [0.0]])      # degree 9
    deg[3, 0] = 3.0
    deg[4] = _poly_from_roots([-2.0, -0.5, 0.5, 2.0], [(0.0, 1.0)] * 3, 1.0)                            # even polynomial
    deg[5] = _poly_from_roots([-1.5, 0.0, 0.3, 1.5], [(0.4, 0.7), (-0.3, 1.9), (0.1, 1.2)], 1.0)       # a root exactly at the first split point
    deg[6] = _poly_from_roots([-0.7, 0.5, 0.5, 1.8], [(1.0, 1.0), (0.2, 0.9), (-0.6, 1.1)], 1.0)       # a double root
    deg[7] = 1e-300 * deg[4]
    roots, counts = ops.debug_real_roots10(torch.from_numpy(deg).to(dev), 1)
    roots, counts = roots.cpu().numpy(), counts.cpu().numpy()
    assert counts[0].sum() == 0 and counts[1].sum() == 0 and counts[3].sum() == 0
    assert np.allclose(_found(roots, counts, 2), [-0.5, 0.25, 2.0], atol=1e-9)
    assert np.allclose(_found(roots, counts, 4), [-2.0, -0.5, 0.5, 2.0], atol=1e-9)
    assert np.allclose(_found(roots, counts, 7), [-2.0, -0.5, 0.5, 2.0], atol=1e-9)
    got5 = _found(roots, counts, 5)                                   # the root AT a split point may be counted on either side or lost
    assert all(np.any(np.abs(got5 - r) < 1e-9) for r in (-1.5, 0.3, 1.5))
    got6 = _found(roots, counts, 6)                                   # the simple roots are found; the double root may or may not be
    assert np.any(np.abs(got6 + 0.7) < 1e-9) and np.any(np.abs(got6 - 1.8) < 1e-9) and np.all(np.isfinite(roots))


def test_hip_isolation_counts_equal_the_cpu_restatement(dev):
    """The kernel's Sturm isolation against its numpy restatement (oracle/sturm_ref.py) on the degree-10 polynomials of
    RANSAC-like five-point samples: the same number of roots per half for (almost) every polynomial, and every root the kernel
    returns lies in one of the restatement's isolating intervals."""
    from differentiable_ransac_amd import ops, synth
    from oracle import cpu_ref as O
    from oracle import sturm_ref as S
    rng = np.random.default_rng(12)
    pairs = [synth.two_view_pair(700 + i, 400, inlier_ratio=0.5, noise=1e-3, dtype=torch.float64) for i in range(4)]
    smp = torch.stack([pairs[i % 4]["matches"][rng.choice(400, 5, replace=False)] for i in range(2048)])
    s = O.nister_poly_system(smp)
    cs = s["cs"].numpy()[s["ok"].numpy()]
    roots, counts = ops.debug_real_roots10(torch.from_numpy(cs).to(dev), 1)
    roots, counts = roots.cpu().numpy(), counts.cpu().numpy()
    inner, _ = S.isolate(cs)
    outer, _ = S.isolate(cs[:, ::-1].copy())
    differ = outside = total = 0
    for i in range(cs.shape[0]):
        for half, ivs in ((0, inner[i]), (1, outer[i])):
            got = roots[i, half, : counts[i, half]]
            w = got if half == 0 else 1.0 / got          # the search variable of the outer half is w = 1 / z
            ivs = [iv for iv in ivs if half == 0 or iv[1] > -1 + 1e-12]
            differ += len(w) != len(ivs)
            for x in w:
                total += 1
                outside += not any(l - 1e-9 <= x <= h + 1e-9 for (l, h) in ivs)
    assert total > 5000 and differ <= 2e-3 * 2 * cs.shape[0] + 1 and outside <= 1e-3 * total + 1, (differ, outside, total)


def test_sturm_fallback_degree_drops_and_even_roots(dev):
    """Round 5 (review of rounds 3-4: "the Sturm chain still drops roots on a zero leading coefficient"): polynomials whose chain
    loses a degree -- p of degree 9 or 8 stored as a degree-10 polynomial, p with a DOUBLE root (p and p' share a factor), p with a
    root at 0 -- go through the derivative chain instead; every real root numpy's companion-matrix eigenvalues report (what the
    reference calls, nister.py:361-370) is found, double roots once, nothing is invented"""
    from differentiable_ransac_amd import ops
    rng = np.random.default_rng(21)
    coefs, truth, kind = [], [], []

    def distinct(n, lo=-2.5, hi=2.5, sep=0.15):
        while True:
            r = np.sort(rng.uniform(lo, hi, n))
            if (n < 2 or np.diff(r).min() > sep) and np.abs(np.abs(r) - 1).min() > 2e-2 and np.abs(r).min() > 2e-2:
                return r
    for drop in (1, 2):                       # vanishing leading coefficients
        for _ in range(200):
            nreal = int(rng.choice([2, 4, 6])) if drop == 2 else int(rng.choice([1, 3, 5, 7]))
            real = distinct(nreal)
            cp = [(rng.uniform(-2, 2), rng.uniform(0.4, 2.0)) for _ in range((10 - drop - nreal) // 2)]
            c = _poly_from_roots(real, cp, rng.uniform(0.5, 2.0))
            coefs.append(np.concatenate([c, np.zeros(11 - c.shape[0])])); truth.append(real); kind.append("drop%d" % drop)
    for _ in range(300):                      # one double root + simple roots
        real = distinct(int(rng.choice([1, 3, 5])))
        x0 = real[0]
        cp = [(rng.uniform(-2, 2), rng.uniform(0.4, 2.0)) for _ in range((10 - 1 - len(real)) // 2)]
        c = _poly_from_roots(np.concatenate([real, [x0]]), cp, 1.0)
        coefs.append(c); truth.append(real); kind.append("double")
    for _ in range(100):                      # a root at 0 (the reversed polynomial then has a vanishing leading coefficient)
        real = distinct(3)
        cp = [(rng.uniform(-2, 2), rng.uniform(0.4, 2.0)) for _ in range(3)]
        c = _poly_from_roots(np.concatenate([real, [0.0]]), cp, 1.0)
        coefs.append(c); truth.append(np.sort(np.concatenate([real, [0.0]]))); kind.append("zero")
    coef = np.stack(coefs)
    roots, counts = ops.debug_real_roots10(torch.from_numpy(coef).to(dev), 1)
    roots, counts = roots.cpu().numpy(), counts.cpu().numpy()
    lost = {k: 0 for k in set(kind)}
    extra = 0
    for i, real in enumerate(truth):
        got = _found(roots, counts, i)
        tol = 2e-5 if kind[i] == "double" else 1e-7      # a double root is conditioned like sqrt(eps)
        for r in real:
            if len(got) == 0 or np.min(np.abs(got - r) / (1 + abs(r))) > tol:
                lost[kind[i]] += 1
        for g in got:
            if np.min(np.abs(real - g) / (1 + np.abs(real))) > tol:
                extra += 1
    # a double root of ROUNDED coefficients is a pair ~1e-8 apart (real or complex): the chain sees the common factor when its
    # remainder is at rounding level, which most but not all of them reach
    assert lost["drop1"] == 0 and lost["drop2"] == 0 and lost["zero"] == 0 and extra == 0, (lost, extra)
    assert lost["double"] <= 0.15 * 300, lost
