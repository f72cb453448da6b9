"""Round 5: the five-point kernels, lane pairs (path 1) against the two-phase kernels (path 2), interleaved on one box.
    [DRANSAC_LIB=scratch/libdransac_<v>.so] python scratch/r5_k3_time.py [sizes...]
Prints per (solver, size): microseconds per launch of both paths (median of 7 x 20 launches, HIP events), valid counts, slots whose
validity differs, largest model difference where both are valid."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from differentiable_ransac_amd import ops, synth

dev = 'cuda'
sizes = [int(a) for a in sys.argv[1:]] or [131072, 65536, 32768]
pair = synth.two_view_pair(3, 2000)
def samples(n):
    per = 4096
    r = ops.gumbel_topk(pair['logits'][None].to(dev), per, 5, 1.0, None, 17)
    s = ops.gather(pair['matches'][None].to(dev), r['idx'])[0]
    return s.repeat((n + per - 1) // per, 1, 1)[:n].contiguous()
def t(fn, reps=20, rounds=7):
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
print('library:', os.environ.get('DRANSAC_LIB') or 'tree')
for n in sizes:
    s = samples(n)
    for name, fn in (('nister', ops.solve_nister5), ('stewenius', ops.solve_stewenius5)):
        res = {}
        for rnd in range(2):
            for path in ((1,) if os.environ.get('K3_ONLY_PAIRS') else (1, 2)):
                res.setdefault(path, []).append(t(lambda: fn(s, path=path)))
        if os.environ.get('K3_ONLY_PAIRS'):
            print(f'{name:10s} n={n:7d}  pairs {min(res[1]):8.1f} us'); continue
        Ea, va = fn(s, path=1); Eb, vb = fn(s, path=2)
        both = va & vb
        d = (Ea[both] - Eb[both]).abs().amax((-1, -2))
        print(f'{name:10s} n={n:7d}  pairs {min(res[1]):8.1f} us   two-phase {min(res[2]):8.1f} us   ratio {min(res[2]) / min(res[1]):.3f}'
              f'   valid {int(va.sum())} / {int(vb.sum())}  flips {int((va != vb).sum())}  max|dE| {float(d.max()):.2e}  >1e-6: {int((d > 1e-6).sum())}')
if len(sizes) and os.environ.get('K3_HP'):
    s = samples(sizes[0])
    for path in (1, 2):
        print('nister_hp path', path, t(lambda: ops.solve_nister5_hp(s, path=path)), 'us')
