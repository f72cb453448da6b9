"""K1 at the headline shape (128 pairs x 1024 rows x 2000 points, k = 5, fused gather): screened vs unscreened register kernel"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from differentiable_ransac_amd import ops, synth
dev = 'cuda'
for P in (128, 32, 1):
    d = synth.batch_two_view(P, 2000)
    m, lg = d['matches'].to(dev), d['logits'].to(dev)
    def t(fn, reps=30, rounds=7):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        out = []
        for _ in range(rounds):
            e0.record()
            for _ in range(reps): fn()
            e1.record(); torch.cuda.synchronize()
            out.append(e0.elapsed_time(e1) / reps * 1e3)
        return sorted(out)[len(out) // 2]
    res = {}
    for rnd in range(2):
        for sc in (False, True):
            res.setdefault(sc, []).append(t(lambda: ops.gumbel_topk_gather(m, lg, 1024, 5, 1.0, 7, screen=sc)))
    print(f"pairs {P:4d}: unscreened {min(res[False]):7.1f} us   screened (incl. the threshold launch) {min(res[True]):7.1f} us   ratio {min(res[True]) / min(res[False]):.3f}")
