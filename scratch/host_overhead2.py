"""Host issue time vs device time of one step (c2, 32 pairs; train step): is the small-batch step host-bound on this box?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device('cuda')
for mode in ('test', 'train'):
    w = dict(bench.WORKLOADS['c2']); w['pairs'] = 32
    step, info = bench.make_step(w, dev, mode=mode)
    for where in ('null stream', 'side stream'):
        st = torch.cuda.Stream(device=dev) if where == 'side stream' else torch.cuda.current_stream()
        with torch.cuda.stream(st):
            for _ in range(20): step()
            torch.cuda.synchronize()
            n = 200
            t0 = time.perf_counter()
            for _ in range(n): step()
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            # device time of the same loop with a sync per step (host never ahead)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            dts = []
            for _ in range(30):
                torch.cuda.synchronize(); e0.record(); step(); e1.record(); torch.cuda.synchronize(); dts.append(e0.elapsed_time(e1))
        print(f'{mode:5s} {where:12s}: host issue {1e3*(t1-t0)/n:.3f} ms/step, wall {1e3*(t2-t0)/n:.3f} ms/step, device (event, synced) median {sorted(dts)[15]:.3f} ms', flush=True)
