import sys, numpy as np, torch, importlib.util
sys.path.insert(0, ".")
spec = importlib.util.spec_from_file_location("t", "tests/test_gpu_roots.py"); m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
from differentiable_ransac_amd import ops
rng = np.random.default_rng(5)
coef, truth = m._make(rng, 4096, [0, 2, 2, 4, 4, 6, 8, 10])
for method in (1,):
    roots, counts = ops.debug_real_roots10(torch.from_numpy(coef).cuda(), method)
    roots, counts = roots.cpu().numpy(), counts.cpu().numpy()
    errs = []; miss = extra = 0
    for i, real in enumerate(truth):
        got = m._found(roots, counts, i)
        if len(got) != len(real):
            miss += max(0, len(real) - len(got)); extra += max(0, len(got) - len(real)); continue
        errs += list(np.abs(got - real) / (1 + np.abs(real)))
    e = np.array(errs)
    print("method", method, "miss", miss, "extra", extra, "roots", len(e), "median", np.median(e), "p99", np.percentile(e, 99), "p999", np.percentile(e, 99.9), "max", e.max(), "frac>1e-9", (e > 1e-9).mean())
    be = []
    for i in range(len(truth)):
        got = m._found(roots, counts, i)
        for r in got:
            be.append(abs(np.polyval(coef[i][::-1], r)) / np.sum(np.abs(coef[i]) * np.abs(r) ** np.arange(11)))
    be = np.array(be)
    print("   backward error |p(r)| / sum|c_i||r|^i: median", np.median(be), "p99", np.percentile(be, 99), "max", be.max(), "frac>1e-13", (be > 1e-13).mean())
