import os, sys, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device('cuda')
w = dict(bench.WORKLOADS['c2']); w['pairs'] = 32
step, info = bench.make_step(w, dev, mode='train')
for _ in range(30): step()
torch.cuda.synchronize()
for label in ('gc on', 'gc off', 'gc on'):
    if label == 'gc off': gc.collect(); gc.freeze(); gc.disable()
    else: gc.enable()
    segs = []
    for seg in range(12):
        t0 = time.perf_counter()
        for _ in range(100): step()
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        segs.append((round(10*(t1-t0), 3), round(10*(t2-t0), 3)))
    print(label, 'host/wall ms per step per 100-step segment:', segs, flush=True)
    print('   gc counts', gc.get_count(), 'mem allocated MB', torch.cuda.memory_allocated() >> 20, 'reserved MB', torch.cuda.memory_reserved() >> 20)
