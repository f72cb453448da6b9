"""Realistic importance scores for the benchmark (second half) -- runs ONLY in the build container.

Pipeline: synthetic two-view geometry written in the NG-RANSAC `.npy` pair format (matches in pixels + the SIFT side
information the network was trained on: ratio-test value, scale ratio, orientation difference, with inlier-like statistics
for the true matches) -> this package's reader `NGRansacPairs` (= the reference's Dataset, pinned by data_readers.npz) ->
the REFERENCE's scores network (model_cl.DeepRansac_CLNet.ds_0, imported from /root/reference with empty cv2 / h5py stubs)
with the reference's shipped weights pretrained_models/saved_model_5PC_l_epi/model.net, on the CPU -> its three outputs
(`-p 0/1/2`: normalised weights, weights, log-probabilities).  Stored: inputs (matches, side information, K, ground
truth) and the network outputs only -- no reference source, no weights.

    python tests/golden/gen_clnet_logits.py        ->  tests/golden/clnet_logits.npz
"""
import argparse
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
for name in ("cv2", "h5py"):
    sys.modules.setdefault(name, types.ModuleType(name))
sys.path.insert(0, REF)
sys.path.insert(1, REPO)

from differentiable_ransac_amd import synth  # noqa: E402
from differentiable_ransac_amd.data import NGRansacPairs, collate_two_view  # noqa: E402


def write_pair(path, seed, n, inlier_ratio=0.5):
    """one geometric pair in the NG-RANSAC container layout (datasets.py:33-60 reads entries 0-12)"""
    pair = synth.two_view_pair(seed, n, inlier_ratio=inlier_ratio, pixel=True, dtype=torch.float64)
    r = np.random.RandomState(seed)
    inl = pair["inliers"].numpy()
    m = pair["matches"].numpy()
    pts1 = m[None, :, 0:2].astype(np.float32)
    pts2 = m[None, :, 2:4].astype(np.float32)
    # SIFT-like side information: true matches pass the ratio test comfortably and keep scale / orientation consistent
    ratios = np.where(inl, r.beta(2, 5, n) * 0.8, 0.55 + 0.45 * r.rand(n)).astype(np.float32)[None, :, None]
    fs1 = (1.5 + 4 * r.rand(n)).astype(np.float32)
    fs2 = np.where(inl, fs1 * np.exp(0.08 * r.randn(n)), 1.5 + 4 * r.rand(n)).astype(np.float32)
    a1 = (360 * r.rand(n)).astype(np.float32)
    a2 = np.where(inl, a1 + 17.0 + 4 * r.randn(n), 360 * r.rand(n)).astype(np.float32) % 360
    K = pair["K1"].numpy().astype(np.float32)
    arr = np.empty(13, dtype=object)
    for i, v in enumerate([pts1, pts2, ratios, (1000, 1000), (1000, 1000), K, K.copy(), pair["R"].numpy().astype(np.float32),
                           pair["t"].numpy().astype(np.float32).reshape(3, 1), fs1[None, :, None], a1[None, :, None],
                           fs2[None, :, None], a2[None, :, None]]):
        arr[i] = v
    np.save(path, arr, allow_pickle=True)
    return inl


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=4)
    ap.add_argument("--nfeatures", type=int, default=2000)
    a = ap.parse_args()
    torch.manual_seed(0)
    torch.set_num_threads(8)
    with tempfile.TemporaryDirectory() as root:
        folder = root + os.sep
        for p in range(a.pairs):
            write_pair(folder + f"pair{p}.npy", 700 + p, 2600)     # 2600 raw matches; the 0.8 ratio test keeps ~ 2000+
        ds = NGRansacPairs([folder], 0.8, a.nfeatures, False)
        ds.files.sort()
        items = [ds[i] for i in range(len(ds))]
    batch = collate_two_view(items)
    corr = torch.stack([it["correspondences"] for it in items])          # [P, 7, N, 1], what train.py / test.py feed the model
    from model_cl import DeepRansac_CLNet                                # the reference's network definition
    opt = types.SimpleNamespace(fmat=0, sampler=2, precision=1, device="cpu", ransac_batch_size=64, tr=0, weighted=0,
                                threshold=0.75)
    model = DeepRansac_CLNet(opt)
    state = torch.load(os.path.join(REF, "pretrained_models", "saved_model_5PC_l_epi", "model.net"), map_location="cpu")
    model.load_state_dict(state)
    model.eval()
    outs = {}
    with torch.no_grad():
        for p_type, key in ((0, "weights_normalized"), (1, "weights"), (2, "log_probs")):
            w, _ = model(corr, batch["K1"], batch["K2"], None, None, prob_type=p_type, predict=False)
            outs[key] = w.float().numpy()
    # geometric ground truth of the resampled points: Sampson distance to gt_E below the test-time threshold
    m = batch["matches"].double()
    E = batch["gt_E"].double()
    h1 = torch.cat((m[..., :2], torch.ones_like(m[..., :1])), -1)
    h2 = torch.cat((m[..., 2:], torch.ones_like(m[..., :1])), -1)
    Ex1 = h1 @ E.transpose(-1, -2)
    Etx2 = h2 @ E
    r = (h2 * Ex1).sum(-1)
    d2 = r ** 2 / (Ex1[..., 0] ** 2 + Ex1[..., 1] ** 2 + Etx2[..., 0] ** 2 + Etx2[..., 1] ** 2)
    inl = d2 < (1.5 * 0.75 / 1000.0) ** 2
    sep = [(float(outs["log_probs"][p][inl[p].numpy()].mean()), float(outs["log_probs"][p][~inl[p].numpy()].mean()))
           for p in range(a.pairs)]
    print("mean log-probability of geometric inliers / outliers per pair:", sep)
    np.savez_compressed(os.path.join(HERE, "clnet_logits.npz"), matches=batch["matches"].numpy(), side=batch["side"].numpy(),
                        K1=batch["K1"].numpy(), K2=batch["K2"].numpy(), gt_E=batch["gt_E"].numpy(),
                        geometric_inliers=inl.numpy(), **outs)
    print("wrote clnet_logits.npz", {k: v.shape for k, v in outs.items()}, "inlier fraction", float(inl.double().mean()))


if __name__ == "__main__":
    main()
