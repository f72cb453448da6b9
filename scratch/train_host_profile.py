"""Host-side profile of the train step (c2, 32 pairs): where does the Python / autograd time go?"""
import os, sys, time, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device('cuda')
w = dict(bench.WORKLOADS['c2']); w['pairs'] = 32
step, info = bench.make_step(w, dev, mode='train')
for _ in range(30): step()
torch.cuda.synchronize()
for seg in range(4):
    t0 = time.perf_counter()
    for _ in range(100): step()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f'segment {seg}: host issue {10*(t1-t0):.3f} ms/step, wall {10*(t2-t0):.3f} ms/step', flush=True)
pr = cProfile.Profile(); pr.enable()
for _ in range(200): step()
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(28); print(s.getvalue()[:6000])
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(20): step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=40, max_name_column_width=60)[:9000])
