"""CPU restatement (numpy, f64) of the real-root ISOLATION the five-point kernels run since round 3
(differentiable_ransac_amd/csrc/solver_common.hpp: real_roots_half_sturm) -- test infrastructure only, like the rest of oracle/.

The reference finds the roots of the degree-10 polynomial with torch.linalg.eigvals of the companion matrix
(essential_matrix_estimator_nister.py:361-370) and keeps the real parts of ALL eigenvalues (Q10); the kernels keep the real
roots only.  This module restates HOW the kernels isolate them, so that the algorithm can be pinned on the CPU against numpy's
eigenvalues (tests/test_oracle_sturm.py) independently of the GPU:

  chain      f0 = p, f1 = p', f_{k+1} = -rem(f_{k-1}, f_k) as division-free pseudo-remainders with positive multipliers
             b^2 A - (a_n b x + a_{n-1} b - a_n b_{n-2}) B, each renormalised to max |coef| = 1;
  V(x)       number of sign changes of f0(x) .. f10(x) (a zero counts as positive, as the kernel's sign bit does);
  isolate    bisection of (-1, 1] on V, splitting a hair off the centre, until every interval holds one root;
             an interval is kept when p changes sign over it (the kernel then refines it by bisection + Newton).
"""
import numpy as np

SPLIT = 0.49999952316284180      # the kernel's split point: l + SPLIT * (h - l)


def horner(c, x):
    """c [B, n+1] ascending, x [B] -> p(x) [B]"""
    r = c[:, -1].copy()
    for i in range(c.shape[1] - 2, -1, -1):
        r = r * x + c[:, i]
    return r


def sturm_chain(c):
    """c [B, 11] ascending (max |coef| = 1) -> list of 11 arrays, F[k] [B, 11 - k] ascending"""
    F = [c.copy()]
    d = c[:, 1:] * np.arange(1, 11)
    mx = np.abs(d).max(1, keepdims=True)
    mx[mx == 0] = 1
    F.append(d / mx)
    for k in range(1, 10):
        A, Bp = F[k - 1], F[k]
        n = A.shape[1] - 1
        a_n, a_n1, b = A[:, n], A[:, n - 1], Bp[:, n - 1]
        b2 = Bp[:, n - 2]
        q1, q0, bb = a_n * b, a_n1 * b - a_n * b2, b * b
        R = q0[:, None] * Bp[:, :n - 1] - bb[:, None] * A[:, :n - 1]
        R[:, 1:] += q1[:, None] * Bp[:, :n - 2]
        mx = np.abs(R).max(1, keepdims=True)
        mx[(mx == 0) | ~np.isfinite(mx)] = 1
        F.append(R / mx)
    return F


def variations(F, x):
    """sign changes of the chain at x [B] (sign bit semantics: +0 is positive) and the sign bit of p(x)"""
    neg = np.stack([np.signbit(horner(f, x)) for f in F], 1)
    return (neg[:, 1:] != neg[:, :-1]).sum(1), neg[:, 0]


def isolate(c, max_steps=640):
    """c [B, 11] ascending, any scale -> per polynomial the list of isolating intervals (lo, hi) of its real roots in (-1, 1]
    over which p changes sign, in ascending order; plus the number of chain evaluations per polynomial"""
    c = np.asarray(c, dtype=np.float64)
    B = c.shape[0]
    cmax = np.abs(c).max(1, keepdims=True)
    ok = np.isfinite(cmax[:, 0]) & (cmax[:, 0] > 0)
    cn = np.where(ok[:, None], c / np.where(cmax > 0, cmax, 1), 0.0)
    cn[~ok, 0] = 1.0
    F = sturm_chain(cn)
    vm, sm = variations(F, -np.ones(B))
    vp, sp = variations(F, np.ones(B))
    out = [[] for _ in range(B)]
    evals = np.full(B, 2)
    for i in range(B):
        if not ok[i] or vm[i] - vp[i] < 1:
            continue
        Fi = [f[i:i + 1] for f in F]
        stack = [(-1.0, 1.0, int(vm[i]), bool(sm[i]), int(vp[i]), bool(sp[i]))]
        steps = 0
        while stack and steps < max_steps:
            steps += 1
            l, h, vl, sl, vh, sh = stack.pop()
            if vl - vh == 1:
                if sl != sh:
                    out[i].append((l, h))
                continue
            mid = l + SPLIT * (h - l)
            if not (l < mid < h and h - l > 1e-12):
                continue
            v, s = variations(Fi, np.array([mid]))
            v, s = int(v[0]), bool(s[0])
            evals[i] += 1
            if v - vh >= 1:
                stack.append((mid, h, v, s, vh, sh))
            if vl - v >= 1:
                stack.append((l, mid, vl, sl, v, s))     # left part on top: ascending output
    return out, evals
