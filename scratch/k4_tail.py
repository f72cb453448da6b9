"""Does the second partial round of blocks cost K4 time?  Same data, model count chosen so that the grid is exactly one
resident round (4096 blocks of 2 waves) vs the benchmark's 5120 blocks."""
import torch, sys
sys.path.insert(0, '.')
from differentiable_ransac_amd import ops, synth
dev = 'cuda'; P, N, B = 32, 2000, 1024
d = synth.batch_two_view(P, N); m = d['matches'].to(dev)
r = ops.gumbel_topk(d['logits'].to(dev), B, 5, 1.0, None, seed=1)
models, valid = ops.solve_nister5(ops.gather(m, r['idx'], r['y_sel']))
flat = models.reshape(P, -1, 3, 3); v = valid.reshape(P, -1)
thr = torch.full((P,), 7.5e-4, device=dev)
def t(M, reps=20):
    fm, fv = flat[:, :M].contiguous(), v[:, :M].contiguous()
    f = lambda: ops.msac_score(m, fm, thr, True, fv)
    f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3, float(fv.float().mean())
for M in (4096, 8192, 10240, 12288, 16384):
    if M > flat.shape[1]: continue
    us, vf = t(M)
    print(f'M={M:6d} blocks={P * M // 64:5d} ({P * M // 64 / 4096:.2f} rounds)  {us:7.1f} us   {us / M * 1e3:6.2f} ns per model slot   valid {vf:.3f}')
