"""CPU prototype: Sturm-sequence isolation of the real roots of the five-point degree-10 polynomial, against numpy's roots.
Polynomials come from the oracle's nister_poly_system on RANSAC-like samples (mixed inliers / outliers)."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from oracle import cpu_ref as O
from differentiable_ransac_amd import synth

rng = np.random.default_rng(0)
Bn = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
IR = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
NZ = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-3
pairs = [synth.two_view_pair(100 + i, 500, inlier_ratio=IR, noise=NZ, dtype=torch.float64) for i in range(8)]
smp = []
for i in range(Bn):
    m = pairs[i % 8]["matches"]
    idx = rng.choice(m.shape[0], 5, replace=False)
    smp.append(m[idx])
smp = torch.stack(smp)
s = O.nister_poly_system(smp)
cs = s["cs"].numpy()            # ascending, [B, 11]
ok = s["ok"].numpy()
cs = cs[ok]
Bn = cs.shape[0]
cs = cs / np.abs(cs).max(1, keepdims=True)
print("polynomials", Bn)

def horner(c, x):       # c [B, n+1] ascending, x [B]
    r = c[:, -1].copy()
    for i in range(c.shape[1] - 2, -1, -1):
        r = r * x + c[:, i]
    return r

def sturm_chain(c):     # generic degrees 10, 9, ..., 0 ; pseudo-remainders with positive multipliers, renormalised
    B = c.shape[0]
    F = [c.copy()]
    d = c[:, 1:] * np.arange(1, 11)
    d = d / np.abs(d).max(1, keepdims=True)
    F.append(d)
    for k in range(1, 10):
        A, Bp = F[k - 1], F[k]          # deg n, n-1
        n = A.shape[1] - 1
        a_n, a_n1 = A[:, n], A[:, n - 1]
        b = Bp[:, n - 1]
        b2 = Bp[:, n - 2] if n >= 2 else np.zeros(B)
        # b^2 A - (a_n b x + (a_{n-1} b - a_n b_{n-2})) Bp
        q1 = a_n * b
        q0 = a_n1 * b - a_n * b2
        R = (b * b)[:, None] * A[:, :n - 1]
        R = R - q0[:, None] * Bp[:, :n - 1]
        R[:, 1:] -= q1[:, None] * Bp[:, :n - 2]
        R = -R
        mx = np.abs(R).max(1, keepdims=True)
        mx[mx == 0] = 1
        F.append(R / mx)
    return F

def variations(F, x):
    vals = [horner(f, x) for f in F]
    V = np.zeros(x.shape[0], dtype=np.int64)
    prev = np.sign(vals[0])
    for v in vals[1:]:
        sg = np.sign(v)
        ch = (sg != 0) & (prev != 0) & (sg != prev)
        V += ch
        prev = np.where(sg != 0, sg, prev)
    return V

def isolate(c):
    """returns list per polynomial of isolating brackets in [-1, 1], evaluation counts"""
    B = c.shape[0]
    F = sturm_chain(c)
    one = np.ones(B)
    Vm, Vp = variations(F, -one), variations(F, one)
    nroots = np.maximum(Vm - Vp, 0)
    evals = np.full(B, 2)
    brackets = [[] for _ in range(B)]
    # breadth-first subdivision, vectorised over a flat work list
    work = [(i, -1.0, 1.0, int(Vm[i]), int(Vp[i])) for i in range(B) if Vm[i] - Vp[i] >= 1]
    rounds = 0
    while work:
        rounds += 1
        idx = np.array([w[0] for w in work]); lo = np.array([w[1] for w in work]); hi = np.array([w[2] for w in work])
        vlo = np.array([w[3] for w in work]); vhi = np.array([w[4] for w in work])
        single = (vlo - vhi) == 1
        nxt = []
        for j in np.nonzero(single)[0]:
            brackets[idx[j]].append((lo[j], hi[j]))
        m = ~single
        if m.any():
            mid = 0.5 * (lo[m] + hi[m])
            Fm = [f[idx[m]] for f in F]
            vm = variations(Fm, mid)
            np.add.at(evals, idx[m], 1)
            for j, (i_, l_, h_, a_, b_, v_, md) in enumerate(zip(idx[m], lo[m], hi[m], vlo[m], vhi[m], vm, mid)):
                if h_ - l_ < 1e-12:
                    continue
                if a_ - v_ >= 1:
                    nxt.append((i_, l_, md, a_, int(v_)))
                if v_ - b_ >= 1:
                    nxt.append((i_, md, h_, int(v_), b_))
        work = nxt
    return brackets, nroots, evals

def check(c, label):
    br, nroots, evals = isolate(c)
    found = miss = phantom = nosign = 0
    B = c.shape[0]
    comp = np.zeros((B, 10, 10))
    comp[:, np.arange(9), np.arange(1, 10)] = 1.0
    lead = c[:, 10:11]
    lead = np.where(lead == 0, 1e-300, lead)
    comp[:, 9, :] = -c[:, :10] / lead
    allr = np.linalg.eigvals(comp)
    for i in range(B):
        r = allr[i]
        r = r[np.isfinite(r)]
        real = r[np.abs(r.imag) <= 1e-8 * (1 + np.abs(r.real))].real
        real = real[(real > -1) & (real <= 1)]
        got = 0
        used = np.zeros(len(real), bool)
        for (l, h) in br[i]:
            pl = np.polyval(c[i][::-1], l); ph = np.polyval(c[i][::-1], h)
            if pl * ph > 0:
                nosign += 1
                continue
            inside = np.nonzero((real > l - 1e-12) & (real <= h + 1e-12) & ~used)[0]
            if len(inside):
                used[inside[0]] = True
                got += 1
            else:
                phantom += 1
        found += got
        miss += len(real) - got
    ev = evals
    w = ev[: (B // 64) * 64].reshape(-1, 64)
    print(f"{label}: true roots {found + miss}, found {found}, missed {miss} ({miss / max(1, found + miss):.5f}), phantom {phantom}, "
          f"brackets without a sign change {nosign};  evaluations per polynomial mean {ev.mean():.2f} max {ev.max()}, "
          f"mean of per-wave max {w.max(1).mean():.1f}, per-wave total/64 {w.sum(1).mean() / 64:.2f}")

check(cs, "inner  |z| <= 1")
rev = cs[:, ::-1].copy()
check(rev, "outer  (reversed)")


def derivative_chain_roots(c):
    """the tree's method, idealised (every bracket refined to convergence): roots of p^(10-d) bracket those of p^(9-d)"""
    B = c.shape[0]
    from math import factorial
    pts = np.full((B, 1), -1.0)                      # breakpoints so far (left end), right end = 1
    for d in range(1, 11):
        k = 10 - d
        q = np.stack([c[:, i + k] * (factorial(i + k) / factorial(i)) for i in range(d + 1)], 1)   # p^(k), ascending, degree d
        ends = np.concatenate([pts, np.ones((B, 1))], 1)        # [B, d+1]
        fv = np.stack([horner(q, ends[:, i]) for i in range(d + 1)], 1)
        new = [np.full(B, -1.0)]
        for i in range(d):
            lo, hi = ends[:, i].copy(), ends[:, i + 1].copy()
            has = ((fv[:, i] < 0) != (fv[:, i + 1] < 0)) & (hi > lo)
            neg = fv[:, i] < 0
            a, b = lo.copy(), hi.copy()
            for _ in range(60):
                m = 0.5 * (a + b)
                left = (horner(q, m) < 0) == neg
                a = np.where(left, m, a)
                b = np.where(left, b, m)
            new.append(np.where(has, 0.5 * (a + b), hi))
            if d == 10:
                pass
        pts = np.stack(new, 1)
        if d == 10:
            hasl = [((fv[:, i] < 0) != (fv[:, i + 1] < 0)) & (ends[:, i + 1] > ends[:, i]) for i in range(10)]
            return pts[:, 1:], np.stack(hasl, 1)


def check_chain(c, label):
    B = c.shape[0]
    r, has = derivative_chain_roots(c)
    comp = np.zeros((B, 10, 10))
    comp[:, np.arange(9), np.arange(1, 10)] = 1.0
    lead = np.where(c[:, 10:11] == 0, 1e-300, c[:, 10:11])
    comp[:, 9, :] = -c[:, :10] / lead
    allr = np.linalg.eigvals(comp)
    found = miss = phantom = 0
    for i in range(B):
        t = allr[i]
        t = t[np.isfinite(t)]
        real = t[np.abs(t.imag) <= 1e-8 * (1 + np.abs(t.real))].real
        real = real[(real > -1) & (real <= 1)]
        mine = r[i][has[i]]
        used = np.zeros(len(real), bool)
        for x in mine:
            dd = np.abs(real - x)
            dd[used] = 9
            if len(dd) and dd.min() < 1e-6:
                used[dd.argmin()] = True
            else:
                phantom += 1
        found += used.sum()
        miss += (~used).sum()
    print(f"{label} [derivative chain]: true roots {found + miss}, found {found}, missed {miss} ({miss / max(1, found + miss):.5f}), phantom {phantom}")


if len(sys.argv) > 4:
    check_chain(cs, "inner")
    check_chain(rev, "outer")
