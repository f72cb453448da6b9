"""the on-disk format readers against the reference's own Dataset / Dataset3D (golden vectors made by
tests/golden/gen_golden.py from the same synthetic files, written by tests/golden/make_pair_files.py)."""
import os
import sys

import torch

from tests.conftest import load_golden

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
from make_pair_files import make_tree  # noqa: E402


def test_readers_match_reference(tmp_path):
    from differentiable_ransac_amd.data import Match3DPairs, NGRansacPairs, collate_3d, collate_two_view
    g = load_golden("data_readers")
    ng, m3 = make_tree(str(tmp_path))
    for fmat in (False, True):
        ds = NGRansacPairs([ng], ratiothreshold=0.8, nfeatures=200, fmat=fmat)
        order = sorted(range(len(ds)), key=lambda i: ds.files[i])
        assert len(order) == 2
        items = []
        for j, i in enumerate(order):
            torch.manual_seed(100 + j)
            it = ds[i]
            items.append(it)
            tag = f"{'F' if fmat else 'E'}{j}"
            for k in ("correspondences", "gt_F", "gt_E", "gt_R", "gt_t", "K1", "K2", "im_size1", "im_size2"):
                want = g[f"{tag}_{k}"]
                got = it[k]
                assert tuple(got.shape) == tuple(want.shape), (tag, k)
                assert torch.allclose(got.double(), want.double(), rtol=1e-6, atol=1e-7), (tag, k)
            assert it["correspondences"].dtype == torch.float32
        batch = collate_two_view(items)
        assert batch["matches"].shape == (2, 200, 4) and batch["side"].shape == (2, 200, 3) and batch["gt_t"].shape == (2, 3)
        assert torch.equal(batch["matches"][0, :, 0], items[0]["correspondences"][0, :, 0])
    d3 = Match3DPairs([m3], num=100)
    order = sorted(range(len(d3)), key=lambda i: d3.files[i])
    items = []
    for j, i in enumerate(order):
        torch.manual_seed(200 + j)
        it = d3[i]
        items.append(it)
        assert torch.allclose(it["correspondences"].double(), g[f"M{j}_correspondences"].double(), atol=1e-7)
        assert torch.allclose(torch.as_tensor(it["gt_pose"]).double(), g[f"M{j}_gt_pose"].double())
    b3 = collate_3d(items)
    assert b3["matches"].shape == (2, 100, 6) and b3["scores"].shape == (2, 100) and b3["gt_pose"].shape == (2, 4, 4)


def test_few_matches_are_repeated_and_many_subsampled(tmp_path):
    from differentiable_ransac_amd.data import NGRansacPairs
    ng, _ = make_tree(str(tmp_path))
    ds = NGRansacPairs([ng], ratiothreshold=0.8, nfeatures=300)
    for i in range(len(ds)):
        c = ds[i]["correspondences"]
        assert c.shape == (7, 300, 1) and bool(torch.isfinite(c).all())
        assert (c[4] < 0.8).all()          # Lowe ratio filter applied before the resampling
