"""The duck-typed plugin boundary, pinned like the numbers are: tests/golden/plugin_signatures.json holds
inspect.signature of every public method of every plugin class of the reference (written by tests/golden/gen_golden.py in the
build container); every one of them must exist here with the reference's parameters, in the reference's order, as a prefix of
its own -- anything the package adds must be optional."""
import inspect
import json
import os

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))

# reference members that are NOT part of the contract, with the reason (nothing outside the defining class calls them upstream)
NOT_CONTRACT = {
    ("EssentialMatrixEstimator", "get_constraint_mat"): "private maths of the reference's Stewenius implementation (stewenius.py:82-172)",
    ("EssentialMatrixEstimator", "multiply_deg_one_poly"): "same",
    ("EssentialMatrixEstimator", "multiply_two_deg_one_poly"): "same",
    ("EssentialMatrixEstimatorNister", "o1"): "private polynomial products of the reference's Nister implementation (nister.py:410-440)",
    ("EssentialMatrixEstimatorNister", "o2"): "same",
    ("FundamentalMatrixEstimatorNew", "coeff"): "cubic of the reference's 7-point branch, degenerate upstream (SURVEY Q7/Q8)",
    ("RANSAC", "localOptimization"): "LO is out of scope (never ran upstream: SURVEY Q2); the constructor refuses lo != 0",
    ("RANSAC3D", "localOptimization"): "same",
}
# reference parameters whose default differs here ON PURPOSE
DEFAULT_DIFFERS = {
    ("UniformSampler", "sample"): "Q1: the reference's sample() raises TypeError; here it takes the point count (optional)",
}


def _classes():
    from differentiable_ransac_amd import estimators, ransac, samplers, scorings
    return {
        "GumbelSoftmaxSampler": samplers.GumbelSoftmaxSampler, "UniformSampler": samplers.UniformSampler,
        "MSACScore": scorings.MSACScore, "EssentialMatrixEstimatorNister": estimators.EssentialMatrixEstimatorNister,
        "EssentialMatrixEstimator": estimators.EssentialMatrixEstimator,
        "FundamentalMatrixEstimatorNew": estimators.FundamentalMatrixEstimatorNew,
        "RigidTransformationSVDBasedSolver": estimators.RigidTransformationSVDBasedSolver,
        "RANSAC": ransac.RANSAC, "RANSAC3D": ransac.RANSAC3D,
    }


def test_every_reference_plugin_method_exists_with_the_reference_signature_as_a_prefix():
    with open(os.path.join(HERE, "golden", "plugin_signatures.json")) as f:
        ref = json.load(f)
    mine = _classes()
    assert set(ref) == set(mine)
    checked = 0
    for cname, methods in ref.items():
        for mname, params in methods.items():
            if (cname, mname) in NOT_CONTRACT:
                continue
            fn = getattr(mine[cname], mname, None)
            assert fn is not None, f"{cname}.{mname} is missing"
            got = list(inspect.signature(fn).parameters.values())
            assert len(got) >= len(params), f"{cname}.{mname}: fewer parameters than the reference"
            for (rname, rkind, rdef), g in zip(params, got):
                assert g.name == rname and g.kind.name == rkind, f"{cname}.{mname}: {g.name} vs {rname}"
                if rdef is None:
                    continue           # required upstream: required or optional here
                assert g.default is not inspect.Parameter.empty, f"{cname}.{mname}({rname}) lost its default"
                if (cname, mname) not in DEFAULT_DIFFERS:
                    mydef = repr(g.default)
                    assert mydef == rdef or (rdef == "'torch.float32'" and g.default is torch.float32), \
                        f"{cname}.{mname}({rname}): default {mydef} vs {rdef}"
            for g in got[len(params):]:
                assert g.default is not inspect.Parameter.empty or g.kind.name.startswith("VAR"), \
                    f"{cname}.{mname}: extra parameter {g.name} must be optional"
            checked += 1
    assert checked >= 25


def test_normalize_helper_against_the_reference_vector():
    """FundamentalMatrixEstimatorNew.normalize (fundamental_matrix_estimator.py:177-228): plain tensor ops -> runs on any device"""
    from differentiable_ransac_amd.estimators import FundamentalMatrixEstimatorNew
    g = np.load(os.path.join(HERE, "golden", "f8_helpers.npz"))
    est = FundamentalMatrixEstimatorNew(device="cpu")
    n, T1, T2t = est.normalize(torch.from_numpy(g["samples"]))
    for got, key in ((n, "normalized"), (T1, "T1"), (T2t, "T2t")):
        assert np.abs(got.numpy() - g[key]).max() < 1e-12, key


def test_adaptive_iteration_number_method_is_the_module_function():
    from differentiable_ransac_amd import ransac
    from differentiable_ransac_amd.estimators import EssentialMatrixEstimatorNister
    r = ransac.RANSAC(EssentialMatrixEstimatorNister(device="cpu"), None, None, max_iterations=5000)
    assert r.adaptive_iteration_number(0, 2000, 0.999) == 5000           # probability >= 1 - eps: the cap (ransac.py:205-206)
    v = r.adaptive_iteration_number(1000, 2000, 0.999)
    assert abs(v - ransac.adaptive_iteration_number(1000, 2000, 5, 0.999, 1e-5, 5000)) == 0 and 215 < v < 220


@pytest.mark.gpu
def test_estimate_non_minimal_model_helper_against_the_reference_vector():
    """normalize -> estimate_non_minimal_model is what the reference's estimate_model chains for n > 7
    (fundamental_matrix_estimator.py:172-174); models compared up to sign"""
    from differentiable_ransac_amd.estimators import FundamentalMatrixEstimatorNew
    g = np.load(os.path.join(HERE, "golden", "f8_helpers.npz"))
    dev = torch.device("cuda:0")
    est = FundamentalMatrixEstimatorNew(device=dev)
    s, w = torch.from_numpy(g["samples"]).to(dev), torch.from_numpy(g["weights"]).to(dev)
    n, T1, T2t = est.normalize(s)
    for got, key in ((est.estimate_non_minimal_model(n, T1, T2t), "F"), (est.estimate_non_minimal_model(n, T1, T2t, w), "F_w"),
                     (est.estimate_non_minimal_model(n, None, None), "F_plain")):
        ref = torch.from_numpy(g[key]).to(dev)
        sgn = torch.sign((got * ref).flatten(1).sum(1))[:, None, None]
        scale = ref.flatten(1).norm(dim=1)[:, None, None]
        assert float(((got * sgn - ref) / scale).abs().max()) < 1e-6, key
    u = __import__("differentiable_ransac_amd.samplers", fromlist=["UniformSampler"]).UniformSampler(4, 8, device=dev)
    idx = u.unique_generate(list(range(50)))
    assert idx.shape == (8,) and int(idx.min()) >= 0 and int(idx.max()) <= 49
