"""numpy emulation of the closed-form DLT triangulation used in pose_error.hip (char. polynomial + Newton + adjugate)."""
import numpy as np, torch, sys
sys.path.insert(0, '.')
from oracle import cpu_ref as O

def tri(R, t, x1, y1, x2, y2, newton=12):
    a = [x2 * R[2, k] - R[0, k] for k in range(3)] + [x2 * t[2] - t[0]]
    c = [y2 * R[2, k] - R[1, k] for k in range(3)] + [y2 * t[2] - t[1]]
    g = [[None] * 4 for _ in range(4)]
    for i in range(4):
        for j in range(4):
            g[i][j] = a[i] * a[j] + c[i] * c[j]
    g[0][0] = g[0][0] + 1.0; g[1][1] = g[1][1] + 1.0
    g[0][2] = g[0][2] - x1; g[2][0] = g[0][2]
    g[1][2] = g[1][2] - y1; g[2][1] = g[1][2]
    g[2][2] = g[2][2] + x1 * x1 + y1 * y1
    def minors(b):
        pr = [(0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 3)]
        m01 = {p: b[0][p[0]] * b[1][p[1]] - b[0][p[1]] * b[1][p[0]] for p in pr}
        m23 = {p: b[2][p[0]] * b[3][p[1]] - b[2][p[1]] * b[3][p[0]] for p in pr}
        return m01, m23
    m01, m23 = minors(g)
    c3 = g[0][0] + g[1][1] + g[2][2] + g[3][3]
    c2 = sum(g[i][i] * g[j][j] - g[i][j] ** 2 for i in range(4) for j in range(i + 1, 4))
    M3 = g[2][0] * m01[(1, 2)] - g[2][1] * m01[(0, 2)] + g[2][2] * m01[(0, 1)]
    M2 = g[3][0] * m01[(1, 3)] - g[3][1] * m01[(0, 3)] + g[3][3] * m01[(0, 1)]
    M1 = g[0][0] * m23[(2, 3)] - g[0][2] * m23[(0, 3)] + g[0][3] * m23[(0, 2)]
    M0 = g[1][1] * m23[(2, 3)] - g[1][2] * m23[(1, 3)] + g[1][3] * m23[(1, 2)]
    c1 = M0 + M1 + M2 + M3
    c0 = (m01[(0, 1)] * m23[(2, 3)] - m01[(0, 2)] * m23[(1, 3)] + m01[(0, 3)] * m23[(1, 2)] + m01[(1, 2)] * m23[(0, 3)]
          - m01[(1, 3)] * m23[(0, 2)] + m01[(2, 3)] * m23[(0, 1)])
    lam = np.zeros_like(c0)
    for _ in range(newton):
        p = (((lam - c3) * lam + c2) * lam - c1) * lam + c0
        dp = ((4 * lam - 3 * c3) * lam + 2 * c2) * lam - c1
        lam = lam - np.where(dp != 0, p / np.where(dp != 0, dp, 1), 0)
    b = [[g[i][j] - (lam if i == j else 0) for j in range(4)] for i in range(4)]
    m01, m23 = minors(b)
    C = [[None] * 4 for _ in range(4)]
    C[0][0] = b[1][1] * m23[(2, 3)] - b[1][2] * m23[(1, 3)] + b[1][3] * m23[(1, 2)]
    C[0][1] = -(b[1][0] * m23[(2, 3)] - b[1][2] * m23[(0, 3)] + b[1][3] * m23[(0, 2)])
    C[0][2] = b[1][0] * m23[(1, 3)] - b[1][1] * m23[(0, 3)] + b[1][3] * m23[(0, 1)]
    C[0][3] = -(b[1][0] * m23[(1, 2)] - b[1][1] * m23[(0, 2)] + b[1][2] * m23[(0, 1)])
    C[1][1] = b[0][0] * m23[(2, 3)] - b[0][2] * m23[(0, 3)] + b[0][3] * m23[(0, 2)]
    C[1][2] = -(b[0][0] * m23[(1, 3)] - b[0][1] * m23[(0, 3)] + b[0][3] * m23[(0, 1)])
    C[1][3] = b[0][0] * m23[(1, 2)] - b[0][1] * m23[(0, 2)] + b[0][2] * m23[(0, 1)]
    C[2][2] = b[3][0] * m01[(1, 3)] - b[3][1] * m01[(0, 3)] + b[3][3] * m01[(0, 1)]
    C[2][3] = -(b[3][0] * m01[(1, 2)] - b[3][1] * m01[(0, 2)] + b[3][2] * m01[(0, 1)])
    C[3][3] = b[2][0] * m01[(1, 2)] - b[2][1] * m01[(0, 2)] + b[2][2] * m01[(0, 1)]
    for i in range(4):
        for j in range(i):
            C[i][j] = C[j][i]
    d = np.stack([np.abs(C[k][k]) for k in range(4)])
    k = d.argmax(0)
    X = np.stack([np.choose(k, [C[kk][i] for kk in range(4)]) for i in range(4)], -1)
    return X, lam

if __name__ == '__main__':
    g = {k: torch.from_numpy(v) for k, v in np.load('tests/golden/pose_error.npz').items()}
    E = g['models']; R1, R2, t = O.horn_decompose(E)
    x = g['matches'].numpy()
    ov = O.cheirality_votes(R1, R2, t, g['matches'][:, :2], g['matches'][:, 2:]).numpy()
    bad = 0; worst = 0
    for m in range(E.shape[0]):
        v = np.zeros(4, dtype=int)
        for r, R in enumerate((R1[m].numpy(), R2[m].numpy())):
            X, lam = tri(R, t[m].numpy(), x[:, 0], x[:, 1], x[:, 2], x[:, 3])
            P = np.concatenate((R, t[m].numpy()[:, None]), 1)
            Q = O.triangulate_dlt(torch.eye(3, 4, dtype=torch.float64), torch.from_numpy(P), g['matches'][:, :2], g['matches'][:, 2:]).numpy()
            cosv = np.abs((X * Q).sum(-1)) / np.linalg.norm(X, axis=-1)
            worst = max(worst, (1 - cosv).max())
            z = X[:, 2] / X[:, 3]; d = (X[:, :3] @ R[2] + t[m].numpy()[2] * X[:, 3]) / X[:, 3]
            v[r] = ((z > 0) & (z < 50) & (d > 0) & (d < 50)).sum(); v[2 + r] = ((z < 0) & (-z < 50) & (d < 0) & (-d < 50)).sum()
        if not (v == ov[m]).all(): bad += 1; print(m, v, ov[m])
    print('models with different votes:', bad, ' worst 1-|cos| vs SVD:', worst)
