"""Round 6, review item 1(b): would the five-point root search pay as its OWN launch at two (or more) waves per SIMD?  The standalone
root-search kernel of the test hook (dr_debug_real_roots10: polynomial in, 88 B per sample; roots out) compiled for 1 / 2 waves per
SIMD, on the degree-10 polynomials of 131 072 real RANSAC samples (2048 distinct ones from the oracle, tiled).
  build: python scratch/roots_occupancy.py --build      run (GPU box): python scratch/roots_occupancy.py"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
variants = {"w1": ["-DDR_DBG_ROOTS_WAVES=1"], "w2": ["-DDR_DBG_ROOTS_WAVES=2"]}
if "--build" in sys.argv:
    for name, flags in variants.items():
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=fast", *flags,
                               "-o", f"{ROOT}/scratch/libroots_{name}.so", f"{ROOT}/differentiable_ransac_amd/csrc/solve_fivepoint.hip",
                               f"{ROOT}/differentiable_ransac_amd/csrc/dr_core.hip"])
    sys.exit(0)
import torch
from differentiable_ransac_amd import ops, synth
from oracle import cpu_ref as O
dev = "cuda"
pair = synth.two_view_pair(0, 2000)
idx = ops.gumbel_topk(pair["logits"][None].to(dev), 2048, 5, 1.0, None, 3)["idx"]
smp = ops.gather(pair["matches"][None].to(dev), idx)[0].cpu().double()
cs = O.nister_poly_system(smp)["cs"]                       # [2048, 11] ascending
coef = cs.repeat(64, 1).contiguous().to(dev)               # 131 072 polynomials
n = coef.shape[0]
for rep in range(2):
    for name in variants:
        lib = ctypes.CDLL(f"{ROOT}/scratch/libroots_{name}.so")
        roots = torch.zeros(n, 2, 10, device=dev, dtype=torch.float64)
        counts = torch.zeros(n, 2, device=dev, dtype=torch.int32)
        cp = lambda t: ctypes.c_void_p(t.data_ptr())
        f = lambda: lib.dr_debug_real_roots10(cp(coef), n, 1, cp(roots), cp(counts), None)
        assert f() == 0
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            f()
        b.record()
        torch.cuda.synchronize()
        print(f"{name}: {a.elapsed_time(b) / 20 * 1e3:7.1f} us per launch of {n} polynomials; real roots per polynomial {counts.sum().item() / n:.3f}", flush=True)
