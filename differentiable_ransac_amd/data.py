"""Readers for the two on-disk formats the reference trains and tests on, host side, numpy/torch
only.  They return what the reference's `Dataset.__getitem__` / `Dataset3D.__getitem__` return (same keys, shapes,
dtypes, same use of the global torch RNG for the resampling), plus `collate` helpers that produce the `[P, N, ...]`
batches `BatchedRANSAC` / `RANSAC3D` consume.

  NGRansacPairs   datasets.py:16-129   one `.npy` object array per image pair (NG-RANSAC pre-processed SIFT matches):
                  [pts1 (1,n,2), pts2, ratios (1,n,1), im_size1 (h,w), im_size2, K1, K2, gt_R, gt_t (3,1),
                   f_size1, ang1, f_size2, ang2, ...]
  Match3DPairs    datasets.py:311-352  one `.npz` per point-cloud pair (3DMatch correspondences):
                  transform (4,4), corr_scores (n,), src_corr_points (n,3), ref_corr_points (n,3)

The reference normalises the essential-matrix branch with `cv2.undistortPoints(pts, K, None)` (datasets.py:85-86);
without distortion coefficients that is the pinhole inverse (u - cx) / fx, (v - cy) / fy, which is what is done here.
"""
import math
import os

import numpy as np
import torch


def _resample(x: torch.Tensor, dim: int, n: int) -> torch.Tensor:
    """exactly n entries along `dim`: random subset if there are more, random repetitions if fewer
    (datasets.py:104-117 / :333-349; consumes torch.randperm from the global generator in the same order)."""
    m = x.size(dim)
    if m > n:
        rnd = torch.randperm(m)
        return x.index_select(dim, rnd).narrow(dim, 0, n)
    if m < n:
        result = x
        for _ in range(0, math.ceil(n / m - 1)):
            rnd = torch.randperm(m)
            result = torch.cat((result, x.index_select(dim, rnd)), dim=dim)
        return result.narrow(dim, 0, n)
    return x


class NGRansacPairs(torch.utils.data.Dataset):
    """Drop-in for the reference's `Dataset` (datasets.py:16-129)."""

    def __init__(self, folders, ratiothreshold=0.8, nfeatures=2000, fmat=False):
        self.nfeatures = nfeatures
        self.ratiothreshold = ratiothreshold
        self.fmat = fmat
        self.minset = 7 if fmat else 5
        self.files = []
        for folder in folders:
            self.files += [folder + f for f in os.listdir(folder)]

    def __len__(self):
        return len(self.files)

    def __getitem__(self, index):
        data = np.load(self.files[index], allow_pickle=True, encoding="latin1")
        pts1, pts2, ratios = data[0], data[1], data[2]
        im_size1, im_size2 = torch.from_numpy(np.asarray(data[3])), torch.from_numpy(np.asarray(data[4]))
        K1, K2 = torch.from_numpy(data[5]), torch.from_numpy(data[6])
        gt_R, gt_t = torch.from_numpy(data[7]), torch.from_numpy(data[8])
        f_size1, f_size2 = np.asarray(data[9]), np.asarray(data[11])
        ang1, ang2 = np.asarray(data[10]), np.asarray(data[12])

        ratio_filter = ratios[0, :, 0] < self.ratiothreshold            # Lowe's ratio criterion
        if ratio_filter.sum() >= self.minset:                           # else: keep everything (the reference warns)
            pts1, pts2, ratios = pts1[:, ratio_filter, :], pts2[:, ratio_filter, :], ratios[:, ratio_filter, :]
            f_size1, f_size2 = f_size1[:, ratio_filter, :], f_size2[:, ratio_filter, :]
            ang1, ang2 = ang1[:, ratio_filter, :], ang2[:, ratio_filter, :]
        scale_ratio = f_size2 / f_size1
        ang = ((ang2 - ang1) % 180) * (3.141592653 / 180)

        if self.fmat:   # image coordinates normalised by the image size (datasets.py:72-81)
            pts1, pts2 = pts1.copy(), pts2.copy()
            pts1[0, :, 0] -= float(im_size1[1]) / 2
            pts1[0, :, 1] -= float(im_size1[0]) / 2
            pts1 /= float(max(im_size1))
            pts2[0, :, 0] -= float(im_size2[1]) / 2
            pts2[0, :, 1] -= float(im_size2[0]) / 2
            pts2 /= float(max(im_size2))
        else:           # calibrated coordinates (cv2.undistortPoints without distortion, datasets.py:85-86)
            def pinhole_inverse(p, K):
                K = K.numpy()
                out = np.empty_like(p)
                out[..., 0] = (p[..., 0] - K[0, 2]) / K[0, 0]
                out[..., 1] = (p[..., 1] - K[1, 2]) / K[1, 1]
                return out
            pts1, pts2 = pinhole_inverse(pts1, K1), pinhole_inverse(pts2, K2)
        corr = np.concatenate((pts1, pts2, ratios, scale_ratio, ang), axis=2)      # [1, n, 7]
        corr = torch.from_numpy(np.transpose(corr))                                 # [7, n, 1]
        if self.nfeatures > 0:
            corr = _resample(corr, 1, self.nfeatures)

        t = gt_t.reshape(-1).to(torch.float32)
        tx = torch.tensor([[0.0, -float(t[2]), float(t[1])], [float(t[2]), 0.0, -float(t[0])],
                           [-float(t[1]), float(t[0]), 0.0]], dtype=torch.float32)
        gt_E = tx.mm(gt_R)
        gt_F = K2.inverse().transpose(0, 1).mm(gt_E).mm(K1.inverse())
        return {"correspondences": corr.float(), "gt_F": gt_F, "gt_E": gt_E, "gt_R": gt_R, "gt_t": gt_t, "K1": K1,
                "K2": K2, "im_size1": im_size1, "im_size2": im_size2, "files": self.files[index]}


class Match3DPairs(torch.utils.data.Dataset):
    """Drop-in for the reference's `Dataset3D` (datasets.py:311-352)."""

    def __init__(self, folders, num=4000):
        self.files = []
        for folder in folders:
            self.files += [folder + f for f in os.listdir(folder)]
        self.num = num

    def __len__(self):
        return len(self.files)

    def __getitem__(self, index):
        data = np.load(self.files[index])
        gt_pose = data["transform"]
        corr = np.concatenate((data["src_corr_points"], data["ref_corr_points"], np.expand_dims(data["corr_scores"], -1)),
                              axis=-1)
        corr = torch.from_numpy(corr)
        if self.num > 0:
            if corr.shape[0] > self.num:
                gt_pose = gt_pose[:self.num]     # sic (datasets.py:338): a no-op on the 4x4 pose
            corr = _resample(corr, 0, self.num)
        return {"correspondences": corr, "gt_pose": gt_pose}


def collate_two_view(items, device=None):
    """List of NGRansacPairs items -> the batch BatchedRANSAC takes: matches [P,N,4], side [P,N,3] (ratio, scale, angle),
    K1, K2 [P,3,3], gt_E, gt_F, gt_R [P,3,3], gt_t [P,3]."""
    corr = torch.stack([it["correspondences"][:, :, 0].t() for it in items])      # [P, N, 7]
    out = {"matches": corr[..., :4].contiguous(), "side": corr[..., 4:].contiguous()}
    for k in ("K1", "K2", "gt_E", "gt_F", "gt_R"):
        out[k] = torch.stack([it[k].to(torch.float32) for it in items])
    out["gt_t"] = torch.stack([it["gt_t"].reshape(3).to(torch.float32) for it in items])
    return {k: v.to(device) for k, v in out.items()} if device is not None else out


def collate_3d(items, device=None):
    """List of Match3DPairs items -> matches [P,N,6] (src xyz, ref xyz), scores [P,N], gt_pose [P,4,4]."""
    corr = torch.stack([it["correspondences"] for it in items]).to(torch.float32)
    out = {"matches": corr[..., :6].contiguous(), "scores": corr[..., 6].contiguous(),
           "gt_pose": torch.stack([torch.as_tensor(it["gt_pose"], dtype=torch.float32) for it in items])}
    return {k: v.to(device) for k, v in out.items()} if device is not None else out
