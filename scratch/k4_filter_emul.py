"""CPU emulation of the K4 candidate filter (csrc/msac_filter.hip): the f16-split matrix-core evaluation of
r = x2^T M x1 and of J as a quadratic form, the conservative inequality, and the claim it must satisfy:

    every (model, point) the exact f32 chain calls an inlier  =>  the filter calls it a candidate.

Run on the CPU box:  python scratch/k4_filter_emul.py
Prints, per data set, the number of exact inliers, of candidates, of MISSED inliers (must be 0) and the smallest
margin (Jhat - rt^2) / Jhat over the exact inliers.
"""
import sys

import numpy as np

sys.path.insert(0, '.')
import torch
from differentiable_ransac_amd import synth

EPS = 1.0 / 16.0
KAPPA_R = 5e-6      # relative error bound of rt (split representation + f32 accumulation + the exact chain's own error)
E_ABS = 4e-3        # absolute slack (scaled units)
KAPPA_J = 4e-6


def f16(x):
    return np.asarray(x, np.float32).astype(np.float16).astype(np.float32)


def split2(x):
    x = np.asarray(x, np.float32)
    h = f16(x)
    l = f16(x - h)
    return h, l


def exact_chain(m, x1, y1, x2, y2, inv_thr2):
    """the production f32 arithmetic (sampson_s), evaluated with float32 numpy fma emulation by float64 + rounding"""
    f = np.float32

    def fma(a, b, c):
        return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)

    m = [np.full_like(x1, f(v)) for v in m]
    a0 = fma(x2, m[0], fma(y2, m[3], m[6]))
    a1 = fma(x2, m[1], fma(y2, m[4], m[7]))
    a2 = fma(x2, m[2], fma(y2, m[5], m[8]))
    b0 = fma(x1, m[0], fma(y1, m[1], m[2]))
    b1 = fma(x1, m[3], fma(y1, m[4], m[5]))
    r = fma(x1, a0, fma(y1, a1, a2))
    jj = fma(a0, a0, fma(a1, a1, fma(b0, b0, (b1 * b1).astype(np.float32))))
    with np.errstate(all='ignore'):
        rc = (f(1) / jj).astype(np.float32)
        sv = fma((r * r).astype(np.float32) * rc, np.full_like(x1, f(inv_thr2)), np.full_like(x1, f(-1)))
    return sv


DEG = np.array([2, 2, 1, 2, 2, 1, 1, 1, 0])
JR = [0, 3, 6, 1, 4, 7, 2, 5]   # coefficient of the features x1x2, x1y2, x1, y1x2, y1y2, y1, x2, y2


def filter_pair(matches, models, thr, rng=None):
    """matches [N,4] f32, models [M,9] f32 -> (exact inlier mask [M,N], candidate mask [M,N])"""
    matches = np.asarray(matches, np.float32)
    models = np.asarray(models, np.float32)
    theta = np.float32(1.5) * np.float32(thr)
    inv_thr2 = np.float32(1.0) / (theta * theta)
    fin = np.isfinite(matches)
    cmax = np.max(np.abs(np.where(fin, matches, 0)))
    s = 0 if cmax == 0 else int(np.floor(np.log2(cmax))) + 1
    thp = float(theta) * 2.0 ** -s
    ok_mode = np.isfinite(thp) and 2.0 ** -14 < thp < 16.0 and abs(s) <= 16
    g = int(np.clip(-np.floor(np.log2(thp)), -15, 15)) if ok_mode else 0
    Theta = np.float32(2.0 ** g * thp)
    A = np.float32((1.0 + EPS + 1e-3)) * Theta * Theta
    x1, y1, x2, y2 = [np.ldexp(matches[:, i], -s).astype(np.float32) for i in range(4)]
    one = np.ones_like(x1)
    F = np.stack([x1 * x2, x1 * y2, x1, y1 * x2, y1 * y2, y1, x2, y2]).astype(np.float32)        # [8,N]
    G = np.stack([x2 * x2, x2 * y2, y2 * y2, x2, y2, x1 * x1, x1 * y1, y1 * y1, x1, y1]).astype(np.float32)
    Fsum = np.abs(F).sum(0) + 1
    Gsum = np.abs(G).sum(0) + 1
    Ep = KAPPA_R * 2.0 ** g * Fsum + E_ABS
    H = (1 + 1 / EPS) * Ep * Ep + KAPPA_J * 4 * A * Gsum + 1e-5
    H16 = f16(H * (1 + 2.0 ** -9) + 1e-7)
    H16 = np.where(H16 < H, np.nextafter(H16.astype(np.float16), np.float16(np.inf)).astype(np.float32), H16)
    Fh, Fl = split2(F)
    Gh, Gl = split2(G)
    Mx, N = models.shape[0], matches.shape[0]
    exact = np.zeros((Mx, N), bool)
    cand = np.zeros((Mx, N), bool)
    for i in range(Mx):
        m = models[i]
        sv = exact_chain(m, matches[:, 0], matches[:, 1], matches[:, 2], matches[:, 3], inv_thr2)
        exact[i] = sv < 0
        if not np.all(np.isfinite(m)) or not np.any(m != 0):
            continue   # dead column: never a candidate (the exact chain yields no inlier for it either)
        mp = np.ldexp(m.astype(np.float64), s * DEG)
        mx = np.max(np.abs(mp))
        e = -(int(np.floor(np.log2(mx))) + 1)
        if not ok_mode or not np.isfinite(mx) or abs(e) > 30:
            cand[i] = True     # all-candidates mode
            continue
        mpp = np.ldexp(mp, e).astype(np.float32)          # max in [0.5, 1)
        chat = np.ldexp(mpp, g).astype(np.float32)
        ch, cl = split2(chat[JR])
        rt = (ch[:, None] * Fh + ch[:, None] * Fl + cl[:, None] * Fh + cl[:, None] * Fl).astype(np.float64).sum(0) + chat[8]
        rt = rt.astype(np.float32)
        if rng is not None:   # emulate a pessimistic f32 accumulation error inside the matrix core
            rt = (rt + rng.uniform(-1, 1, N) * 32 * 2.0 ** -24 * (np.abs(chat[JR])[:, None] * np.abs(F)).sum(0)).astype(np.float32)
        q = np.array([mpp[0] ** 2 + mpp[1] ** 2, 2 * (mpp[0] * mpp[3] + mpp[1] * mpp[4]), mpp[3] ** 2 + mpp[4] ** 2,
                      2 * (mpp[0] * mpp[6] + mpp[1] * mpp[7]), 2 * (mpp[3] * mpp[6] + mpp[4] * mpp[7]),
                      mpp[0] ** 2 + mpp[3] ** 2, 2 * (mpp[0] * mpp[1] + mpp[3] * mpp[4]), mpp[1] ** 2 + mpp[4] ** 2,
                      2 * (mpp[0] * mpp[2] + mpp[3] * mpp[5]), 2 * (mpp[1] * mpp[2] + mpp[4] * mpp[5])], np.float32)
        qc = np.float32(mpp[6] ** 2 + mpp[7] ** 2 + mpp[2] ** 2 + mpp[5] ** 2)
        qh, ql = split2(A * q)
        Jt = (qh[:, None] * Gh + qh[:, None] * Gl + ql[:, None] * Gh).astype(np.float64).sum(0) + A * qc + H16
        Jt = Jt.astype(np.float32)
        cand[i] = (rt * rt).astype(np.float32) <= Jt
    return exact, cand


def report(tag, matches, models, thr, rng=None):
    ex, ca = filter_pair(matches, models, thr, rng)
    missed = int((ex & ~ca).sum())
    print(f'{tag:34s} evals {ex.size:9d}  exact inliers {int(ex.sum()):8d}  candidates {int(ca.sum()):8d} '
          f'(x{ca.sum() / max(1, ex.sum()):.3f})  groups-of-4 {int(ca.reshape(ca.shape[0], -1, 4).any(-1).sum()):7d}  MISSED {missed}')
    return missed


def main():
    rng = np.random.default_rng(0)
    bad = 0
    for seed in range(3):
        pair = synth.two_view_pair(seed, 2000)
        mt = pair['matches'].numpy()
        E = pair['gt_E'].numpy().reshape(9)
        gen = torch.Generator().manual_seed(seed)
        models = np.concatenate([E[None], E[None] * 37.5, E[None] * 1e-3,
                                 E[None] + 0.002 * torch.randn(40, 9, generator=gen).numpy(),
                                 E[None] + 0.05 * torch.randn(120, 9, generator=gen).numpy(),
                                 torch.randn(40, 9, generator=gen).numpy()]).astype(np.float32)
        bad += report(f'E, normalised coords, seed {seed}', mt, models, 7.5e-4, rng)
    # F matrices on pixel coordinates
    pair = synth.two_view_pair(7, 2000, pixel=True)
    F = pair['gt_F'].numpy().reshape(9)
    gen = torch.Generator().manual_seed(7)
    scale = np.abs(F) + 1e-9
    models = np.concatenate([F[None], F[None] * (1 + 0.001 * torch.randn(60, 9, generator=gen).numpy()),
                             F[None] + scale * 0.05 * torch.randn(100, 9, generator=gen).numpy()]).astype(np.float32)
    bad += report('F, pixel coords, thr 0.75', pair['matches'].numpy(), models, 0.75, rng)
    bad += report('F, pixel coords, thr 3', pair['matches'].numpy(), models, 3.0, rng)
    # degenerate inputs: points at the origin / on the axes, tiny and huge coefficients, tiny thresholds
    pair = synth.two_view_pair(11, 512)
    mt = pair['matches'].numpy().copy()
    mt[:50, 0] = 0; mt[50:100, 1] = 0; mt[100:150, 2:] = 0; mt[150:160] = 0; mt[160:170] = 1e-30
    E = pair['gt_E'].numpy().reshape(9)
    gen = torch.Generator().manual_seed(11)
    models = np.concatenate([E[None], E[None] * 1e-20, E[None] * 1e20, np.zeros((1, 9)), np.eye(3).reshape(1, 9),
                             E[None] + 0.01 * torch.randn(50, 9, generator=gen).numpy(),
                             np.array([[1e-8, 0, 0, 0, 1e-8, 0, 0, 0, 1.0]]), np.array([[0, 0, 1, 0, 0, 0, 0, 0, 0.]]),
                             np.array([[0, 0, 0, 0, 0, 1, 0, -1, 0.]])]).astype(np.float32)
    for thr in (7.5e-4, 1e-6, 1e-9, 0.5, 100.0, 0.0):
        bad += report(f'degenerate inputs, thr {thr:g}', mt, models, thr, rng)
    print('TOTAL MISSED', bad)
    return bad


if __name__ == '__main__':
    sys.exit(1 if main() else 0)
