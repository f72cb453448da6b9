"""The headline workload WITH the final refit (K7; `with_final_refit` of the bench line): 128 pairs x 2000 points x 1024 hypotheses,
one batch, refit=True -- run under `rocprofv3 --kernel-trace --stats` for the per-launch breakdown (profiles/r5_kernel_stats_with_refit.md)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from differentiable_ransac_amd import synth
from differentiable_ransac_amd.ransac import BatchedRANSAC
dev = 'cuda'
P, N, B = 128, 2000, 1024
d = synth.batch_two_view(P, N)
m, lg, K1, K2 = (d[k].to(dev) for k in ('matches', 'logits', 'K1', 'K2'))
for refit in (False, True):
    rn = BatchedRANSAC('nister', ransac_batch_size=B, train=False, threshold=0.75, max_iterations=B, seed=4321, keep_masks=True, refit=refit)
    for _ in range(10):
        rn(m, lg, K1, K2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 100
    for _ in range(n):
        rn(m, lg, K1, K2)
    torch.cuda.synchronize()
    print(f"refit={refit}: {(time.perf_counter() - t0) / n * 1e3:.4f} ms per step (eager issue)")
