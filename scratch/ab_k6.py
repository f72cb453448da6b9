"""Where the update launch's time goes (round 6): the kernel cut off after its arg-max phase (-DDR_K6_DBG_CUT=1) against the whole kernel,
at the headline shape (128 pairs x 10 240 slots) and at config 3's (32 pairs x 40 960 slots).
  build: python scratch/ab_k6.py --build     run (GPU box): python scratch/ab_k6.py
(the knob is NOT in the tree -- msac_score.hip is hash-pinned by the PMC capture: put
    #if DR_K6_DBG_CUT == 1
      if (tid == 0) best_inliers[p] = R == 1 ? one_idx : s_sub_idx[0];
      return;
    #endif
 behind the `if (R > 1) __syncthreads();` of ransac_update_kernel to repeat the measurement.)
Measured (one box, event pair around one launch incl. ~12 us of marker overhead): 128 pairs x 10 240 slots 23.0-24.6 us whole, 20.8-21.4
cut off; 32 x 40 960: 36.7-37.8 / 26.4-28.0; one pair x 20 480: 23.3-24.2 / 21.6-21.7 -- three quarters of the launch are the arg-max
phase (the gate's read, then the scores K4 wrote on other XCDs: dependent trips to memory), not the mask / stop-rule tail."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
variants = {"full": [], "argmax": ["-DDR_K6_DBG_CUT=1"]}
if "--build" in sys.argv:
    for name, flags in variants.items():
        subprocess.check_call([sys.executable, os.path.join(ROOT, "scratch", "build_variant.py"), "k6_" + name, "msac_score.hip", *flags])
    sys.exit(0)
if "--child" not in sys.argv:
    for n in variants:
        subprocess.check_call([sys.executable, __file__, "--child", n], env=dict(os.environ, DRANSAC_LIB=os.path.join(ROOT, "scratch", f"libdransac_k6_{n}.so")))
    sys.exit(0)
name = sys.argv[2]
sys.path.insert(0, ROOT)
import torch
from differentiable_ransac_amd import ops, synth
from differentiable_ransac_amd.ransac import BatchedRANSAC
dev = torch.device("cuda:0")
for (P, B, solver) in ((128, 1024, "nister"), (32, 4096, "stewenius"), (1, 2048, "nister")):
    N = 2000
    d = synth.batch_two_view(P, N)
    m, lg, K1, K2 = (d[k].to(dev) for k in ("matches", "logits", "K1", "K2"))
    rn = BatchedRANSAC(solver, ransac_batch_size=B, train=False, threshold=0.75, max_iterations=B, seed=1, keep_masks=False, refit=False)
    with torch.no_grad():
        models, valid, _ = rn.hypotheses(m, lg)
    flat = models.reshape(P, -1, 9).contiguous()
    st, thr = ops.ransac_init(P, N, 5000, 0.75, K1, K2, dev, torch.float32)
    scores, _ = ops.msac_score(m, flat.reshape(P, -1, 3, 3), thr, valid=valid.reshape(P, -1), want_masks=False)
    def f():
        st.iters.zero_(); st.max_iters.fill_(5000.0)
        ops.ransac_update(st, m, flat, valid.reshape(P, -1), scores, thr, B, 5, 0.999, 1e-5)
    for _ in range(5): f()
    ts = []
    for rep in range(3):
        evs = []
        for _ in range(30):
            st.iters.zero_(); st.max_iters.fill_(5000.0)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            ops.ransac_update(st, m, flat, valid.reshape(P, -1), scores, thr, B, 5, 0.999, 1e-5)
            b.record()
            evs.append((a, b))
        torch.cuda.synchronize()
        ts.append(sorted(x.elapsed_time(y) for x, y in evs)[15] * 1e3)
    print(f"{name:8s} P={P} M={flat.shape[1]}: " + " / ".join(f"{t:.1f}" for t in ts) + " us (event pair around one launch, incl. ~12 us of marker overhead)")
