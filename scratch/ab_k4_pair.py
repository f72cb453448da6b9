"""Same-box A/B of two builds of msac_score.hip (scratch/libk4_old.so vs scratch/libk4_new.so), interleaved timing of
dr_msac_score_f32 at the benchmark shape with masks and validity flags."""
import ctypes, os, sys
sys.path.insert(0, '.')
import torch
from differentiable_ransac_amd import ops, synth
dev = 'cuda'
P, N, B = 32, 2000, 1024
data = synth.batch_two_view(P, N)
r = ops.gumbel_topk(data['logits'].to(dev), B, 5, 1.0, None, seed=1)
smp = ops.gather(data['matches'].to(dev), r['idx'], r['y_sel'])
models, valid = ops.solve_nister5(smp)
flat = models.reshape(P, -1, 9).contiguous(); vflat = valid.reshape(P, -1).contiguous().view(torch.uint8)
M = flat.shape[1]
mt = data['matches'].to(dev).contiguous()
thr = torch.full((P,), 7.5e-4, device=dev)
scores = torch.empty(P, M, device=dev); masks = torch.empty(P, M, N, device=dev, dtype=torch.uint8)
names = sys.argv[1:] or ['old', 'new']
libs = {v: ctypes.CDLL(os.path.abspath(f'scratch/libk4_{v}.so')) for v in names}
def run(lib):
    lib.dr_msac_score_f32(ctypes.c_void_p(mt.data_ptr()), ctypes.c_void_p(flat.data_ptr()), ctypes.c_void_p(vflat.data_ptr()),
                          ctypes.c_void_p(thr.data_ptr()), P, M, N, ctypes.c_void_p(scores.data_ptr()),
                          ctypes.c_void_p(masks.data_ptr()), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
res = {v: [] for v in names}
for rep in range(20):
    for v in names:
        run(libs[v]); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10): run(libs[v])
        b.record(); torch.cuda.synchronize()
        res[v].append(a.elapsed_time(b) / 10)
out = {}
for v in names:
    run(libs[v]); torch.cuda.synchronize()
    out[v] = (scores.clone(), masks.clone())
    t = sorted(res[v])
    print(f'{v}: median {t[len(t)//2]*1e3:.1f} us  min {t[0]*1e3:.1f} us  sum {float(scores.nan_to_num().sum()):.3f} inl {int(masks.sum())}')
a, b = out[names[0]], out[names[-1]]
print('max |dscore| rel', float(((a[0] - b[0]).abs() / a[0].abs().clamp(min=1)).nan_to_num().max()), 'mask bits differing', int((a[1] != b[1]).sum()))
