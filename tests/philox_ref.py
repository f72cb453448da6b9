"""numpy restatement of the counter-based RNG used in-kernel (Philox4x32-7, Salmon et al. SC'11: seven rounds, the smallest
Crush-resistant Philox4x32; ten until round 2) -- test infrastructure, mirrors dr::Philox in
differentiable_ransac_amd/csrc/dr_common.hpp."""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)


ROUNDS = 7


def philox4x32(seed: int, c0, c1, c2, c3, rounds: int = ROUNDS):
    c0, c1, c2, c3 = [np.asarray(c, dtype=np.uint32) for c in np.broadcast_arrays(c0, c1, c2, c3)]
    k0 = np.uint32(seed & 0xFFFFFFFF)
    k1 = np.uint32((seed >> 32) & 0xFFFFFFFF)
    with np.errstate(over="ignore"):
        for _ in range(rounds):
            p0 = M0 * c0.astype(np.uint64)
            p1 = M1 * c2.astype(np.uint64)
            h0, l0 = (p0 >> np.uint64(32)).astype(np.uint32), p0.astype(np.uint32)
            h1, l1 = (p1 >> np.uint64(32)).astype(np.uint32), p1.astype(np.uint32)
            c0, c1, c2, c3 = h1 ^ c1 ^ k0, l1, h0 ^ c3 ^ k1, l0
            k0 = np.uint32(k0 + W0)
            k1 = np.uint32(k1 + W1)
    return c0, c1, c2, c3


def philox4x32_10(seed: int, c0, c1, c2, c3):
    """The ten-round generator (known-answer test against the Random123 vectors only)."""
    return philox4x32(seed, c0, c1, c2, c3, rounds=10)


def gumbel_noise_f32(seed: int, P: int, B: int, N: int):
    """noise [P,B,N] as the f32 kernel generates it: counter (n/4, b, p, 0), word n%4."""
    q = np.arange((N + 3) // 4, dtype=np.uint32)[None, None, :]
    b = np.arange(B, dtype=np.uint32)[None, :, None]
    p = np.arange(P, dtype=np.uint32)[:, None, None]
    w = np.stack(philox4x32(seed, q, b, p, np.uint32(0)), axis=-1).reshape(P, B, -1)[:, :, :N]
    # r = the word rounded to 24 significant bits (what v_cvt_f32_u32 does) * 2^-32, u = fl(r * (1 - eps - tiny) + tiny)
    tiny, eps = np.float32(1.17549435e-38), np.float32(1.1920928955078125e-07)
    r = w.astype(np.float32).astype(np.float64) * 2.0 ** -32
    u = (r * np.float64(np.float32(1.0) - eps - tiny) + np.float64(tiny)).astype(np.float32)
    return -np.log(-np.log(u.astype(np.float64))).astype(np.float32)


def uniform_indices(seed: int, P: int, B: int, k: int, N: int):
    j = np.arange(k, dtype=np.uint32)[None, None, :]
    b = np.arange(B, dtype=np.uint32)[None, :, None]
    p = np.arange(P, dtype=np.uint32)[:, None, None]
    w0 = philox4x32(seed, j, b, p, np.uint32(1))[0]
    return ((w0.astype(np.uint64) * np.uint64(max(N - 1, 1))) >> np.uint64(32)).astype(np.int64)
