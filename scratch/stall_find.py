import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device('cuda')
for mode, pairs, n in (('train', 32, 1500), ('test', 128, 1500), ('test', 32, 3000)):
    w = dict(bench.WORKLOADS['c2']); w['pairs'] = pairs
    step, info = bench.make_step(w, dev, mode=mode)
    ts = []
    t_prev = time.perf_counter()
    for i in range(n):
        step()
        t = time.perf_counter(); ts.append(t - t_prev); t_prev = t
    torch.cuda.synchronize()
    slow = [(i, round(x * 1e3, 1)) for i, x in enumerate(ts) if x > 3e-3]
    print(mode, pairs, 'steps with host time > 3 ms:', slow[:30], ' median ms', round(sorted(ts)[n // 2] * 1e3, 3), flush=True)
