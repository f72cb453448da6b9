import torch, sys
sys.path.insert(0, '.')
from oracle import cpu_ref as O
from differentiable_ransac_amd import ops, synth
from tests.conftest import load_golden
dev = 'cuda'
pair = synth.two_view_pair(77, 640, inlier_ratio=1.0, noise=0.0, dtype=torch.float64)
smp = pair["matches"].reshape(128, 5, 4)
one = torch.ones(1, dtype=torch.bool)
Eo, ok, real = O.nister_5pt(smp)
for name, fn in (("nister", ops.solve_nister5), ("stew", ops.solve_stewenius5)):
    E, valid = fn(smp.to(dev)); E = E.cpu(); valid = valid.cpu()
    d = torch.stack([O.match_solution_sets(pair["gt_E"][None], one, E[b], valid[b])[0] for b in range(128)])
    print(name, 'gt dist top', d.topk(4))
    fw = [O.match_solution_sets(E[b], valid[b], Eo[b], real[b]) for b in range(128)]
    bw = [O.match_solution_sets(Eo[b], real[b], E[b], valid[b]) for b in range(128)]
    fwm = torch.tensor([f.max() if f.numel() else 0 for f in fw]); bwm = torch.tensor([f.max() if f.numel() else 0 for f in bw])
    print(name, 'fw top', fwm.topk(4), 'bw top', bwm.topk(4))
    b = int(d.argmax())
    print('sample', b, 'n valid hip', int(valid[b].sum()), 'oracle real', int(real[b].sum()))
    s = O.nister_poly_system(smp[b:b+1])
    cs = s['cs'][0]
    import numpy as np
    r = np.roots(cs.numpy()[::-1])
    print('roots', np.sort_complex(r))
g = load_golden("f8")
F, valid = ops.solve_f8(g["samples"].to(dev))
print('f8 valid', valid[:4], F[0], g["F_f64"][0])
