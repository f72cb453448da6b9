"""masks / scores of two library builds on the same inputs (bit-equality of masks, max score difference)"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from differentiable_ransac_amd import ops, synth
dev = 'cuda'; P, N, B = 8, 2000, 1024
d = synth.batch_two_view(P, N)
r = ops.gumbel_topk(d['logits'].to(dev), B, 5, 1.0, None, seed=1)
models, valid = ops.solve_nister5(ops.gather(d['matches'].to(dev), r['idx']))
flat = models.reshape(P, -1, 9).contiguous(); v = valid.reshape(P, -1).contiguous().view(torch.uint8)
flat[0, 3] = float('nan'); flat[0, 4] = 0.0
mt = d['matches'].to(dev).contiguous(); thr = torch.full((P,), 7.5e-4, device=dev)
M = flat.shape[1]
out = []
for name in sys.argv[1:]:
    lib = ctypes.CDLL(f'{ROOT}/differentiable_ransac_amd/libdransac.so' if name == 'cur' else f'{ROOT}/scratch/libdransac_{name}.so')
    sc = torch.empty(P, M, device=dev); mk = torch.empty(P, M, N, device=dev, dtype=torch.uint8)
    cp = lambda t: ctypes.c_void_p(t.data_ptr())
    assert lib.dr_msac_score_f32(cp(mt), cp(flat), cp(v), cp(thr), P, M, N, cp(sc), cp(mk), None, None, None) == 0
    torch.cuda.synchronize(); out.append((sc, mk))
(s0, m0), (s1, m1) = out
print('mask bytes differing:', int((m0 != m1).sum()), 'of', m0.numel(), ' max |dscore|:', float((s0.nan_to_num() - s1.nan_to_num()).abs().max()), ' nan pattern equal:', bool(torch.equal(s0.isnan(), s1.isnan())))
