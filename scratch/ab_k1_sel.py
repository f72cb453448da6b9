"""Where the register sampler's time goes (race form, 128 x 1024 x 2000, k = 5): pass A alone, pass A + threshold, the whole kernel.
  build: python scratch/ab_k1_sel.py --build     run (GPU box): python scratch/ab_k1_sel.py [variant ...]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
variants = {"full": [], "passA": ["-DDR_K1_DBG_SELECT=1"], "passA_thr": ["-DDR_K1_DBG_SELECT=2"]}
extra = [a for a in sys.argv[1:] if a.startswith("-D")]
if "--build" in sys.argv:
    for name, flags in variants.items():
        subprocess.check_call([sys.executable, os.path.join(ROOT, "scratch", "build_variant.py"), "k1sel_" + name, "gumbel_topk.hip", *flags, *extra])
    sys.exit(0)
names = [a for a in sys.argv[1:] if not a.startswith("-")] or list(variants)
if "--child" not in sys.argv:
    for n in names:
        subprocess.check_call([sys.executable, __file__, "--child", n], env=dict(os.environ, DRANSAC_LIB=os.path.join(ROOT, "scratch", f"libdransac_k1sel_{n}.so")))
    sys.exit(0)
sys.path.insert(0, ROOT)
import torch
from differentiable_ransac_amd import ops, synth
dev = torch.device("cuda:0")
P, B, N, k = 128, 1024, 2000, 5
d = synth.batch_two_view(P, N)
m, lg = d["matches"].to(dev), d["logits"].to(dev)
for race in (True, False):
    f = lambda: ops.gumbel_topk_gather(m, lg, B, k, 1.0, 7, race=race)
    for _ in range(10): f()
    ts = []
    for rep in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(50): f()
        b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / 50 * 1e3)
    print(f"{names[0]:12s} race={race}: " + " / ".join(f"{t:.1f}" for t in ts) + " us per call")
