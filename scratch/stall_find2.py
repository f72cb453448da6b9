import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device('cuda')
for mode, pairs, nseg, seg in (('test', 128, 24, 50), ('train', 32, 24, 100), ('test', 128, 12, 50)):
    w = dict(bench.WORKLOADS['c2']); w['pairs'] = pairs
    step, info = bench.make_step(w, dev, mode=mode)
    step(); torch.cuda.synchronize()
    out = []
    T0 = time.perf_counter()
    for s in range(nseg):
        t0 = time.perf_counter()
        for i in range(seg): step()
        torch.cuda.synchronize()
        out.append(round((time.perf_counter() - t0) / seg * 1e3, 3))
    print(mode, pairs, f'ms/step per {seg}-step segment (synced):', out, ' total s', round(time.perf_counter() - T0, 2), flush=True)
