"""Numerics of the linear forms of K4 (a = M^T x2, b = M x1) computed as bf16 x bf16 -> f32 products of THREE-way
bf16 splits (hi + mid + lo = the f32 value exactly), six products per coefficient x coordinate term, accumulated in f32 --
what a v_mfma_f32_32x32x16_bf16 does -- against the f32 FMA chain the VALU kernel uses and an f64 reference.
CPU only (numpy); run: python scratch/bf16_split_emul.py"""
import numpy as np

rng = np.random.default_rng(0)


def bf16_trunc(x):
    """round-toward-zero to bf16, returned as f32 (top 16 bits kept)"""
    u = x.astype(np.float32).view(np.uint32) & np.uint32(0xFFFF0000)
    return u.view(np.float32)


def split3(x):
    h = bf16_trunc(x)
    r1 = (x - h).astype(np.float32)           # exact: the low 16 bits
    m = bf16_trunc(r1)
    r2 = (r1 - m).astype(np.float32)          # exact
    l = bf16_trunc(r2)
    return h, m, l, (r2 - l).astype(np.float32)


def dot_split(coefs, coords, const):
    """sum_i coef_i * coord_i + const with six bf16 products per term, f32 accumulation in MFMA k-order"""
    acc = np.zeros(np.broadcast(coefs[0], coords[0]).shape, np.float32)
    for c, x in zip(coefs, coords):
        ch, cm, cl, _ = split3(c)
        xh, xm, xl, _ = split3(x)
        for a, b in ((xh, ch), (xh, cm), (xm, ch), (xh, cl), (xl, ch), (xm, cm)):
            acc = (acc + (a.astype(np.float32) * b.astype(np.float32)).astype(np.float32)).astype(np.float32)
    kh, km, kl, _ = split3(const)
    for k in (kh, km, kl):
        acc = (acc + k).astype(np.float32)
    return acc


def fma_chain(coefs, coords, const):
    """x*c0 + (y*c1 + const) as two f32 FMAs (emulated through f64: one rounding per fma)"""
    inner = (coords[1].astype(np.float64) * coefs[1].astype(np.float64) + const.astype(np.float64)).astype(np.float32)
    return (coords[0].astype(np.float64) * coefs[0].astype(np.float64) + inner.astype(np.float64)).astype(np.float32)


M, N = 4096, 2000
E = rng.standard_normal((M, 1, 3)).astype(np.float32)
E /= np.linalg.norm(E, axis=-1, keepdims=True)
pts = (rng.random((1, N, 2)).astype(np.float32) - 0.5) * 1.2     # normalised image coordinates
x, y = pts[..., 0], pts[..., 1]
c0, c1, c2 = E[..., 0], E[..., 1], E[..., 2]
ref = x.astype(np.float64) * c0 + y.astype(np.float64) * c1 + c2.astype(np.float64)
for name, val in (("f32 fma chain", fma_chain((c0, c1), (x, y), c2)), ("bf16 3-way split, f32 acc", dot_split((c0, c1), (x, y), c2))):
    err = np.abs(val.astype(np.float64) - ref)
    print(f"{name:28s}: max abs err {err.max():.3e}  mean {err.mean():.3e}   (|terms| <= ~1; f32 ulp at 1 = 1.2e-7)")
h, m, l, rest = split3(x)
print("split exact:", float(np.abs(rest).max()) == 0.0, " dropped products bound (m*l, l*m, l*l):", 3 * 2.0 ** -8 * 2.0 ** -16)
