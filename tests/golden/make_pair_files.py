"""Writes small synthetic files in the two on-disk formats of the reference (NG-RANSAC `.npy` pairs, 3DMatch `.npz`)
-- shared by the golden-vector generator and the reader tests, so both see byte-identical inputs."""
import os

import numpy as np


def write_ngransac_pair(path, seed, n):
    r = np.random.RandomState(seed)
    h1, w1, h2, w2 = 480, 640, 600, 800
    pts1 = (r.rand(1, n, 2) * [w1, h1]).astype(np.float32)
    pts2 = (r.rand(1, n, 2) * [w2, h2]).astype(np.float32)
    ratios = r.rand(1, n, 1).astype(np.float32)
    K1 = np.array([[520.0, 0, 320.5], [0, 515.0, 241.0], [0, 0, 1]], dtype=np.float32)
    K2 = np.array([[710.0, 0, 401.0], [0, 705.0, 299.0], [0, 0, 1]], dtype=np.float32)
    a = 0.3
    R = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]], dtype=np.float32)
    t = (r.randn(3, 1) / 3).astype(np.float32)
    fs1 = (1 + r.rand(1, n, 1) * 5).astype(np.float32)
    fs2 = (1 + r.rand(1, n, 1) * 5).astype(np.float32)
    a1 = (r.rand(1, n, 1) * 360).astype(np.float32)
    a2 = (r.rand(1, n, 1) * 360).astype(np.float32)
    arr = np.empty(13, dtype=object)
    for i, v in enumerate([pts1, pts2, ratios, (h1, w1), (h2, w2), K1, K2, R, t, fs1, a1, fs2, a2]):
        arr[i] = v
    np.save(path, arr, allow_pickle=True)


def write_3dmatch_pair(path, seed, n):
    r = np.random.RandomState(seed)
    T = np.eye(4, dtype=np.float32)
    T[:3, 3] = r.randn(3)
    np.savez(path, transform=T, corr_scores=r.rand(n).astype(np.float32),
             src_corr_points=r.rand(n, 3).astype(np.float32), ref_corr_points=r.rand(n, 3).astype(np.float32))


def make_tree(root):
    """root/ng/{a,b}.npy (many / few matches), root/m3d/{a,b}.npz; returns the two folder names (with trailing slash,
    as the reference concatenates folder + file name)."""
    ng, m3 = os.path.join(root, "ng") + os.sep, os.path.join(root, "m3d") + os.sep
    os.makedirs(ng, exist_ok=True), os.makedirs(m3, exist_ok=True)
    write_ngransac_pair(ng + "a.npy", 1, 900)
    write_ngransac_pair(ng + "b.npy", 2, 60)
    write_3dmatch_pair(m3 + "a.npz", 3, 500)
    write_3dmatch_pair(m3 + "b.npz", 4, 40)
    return ng, m3
