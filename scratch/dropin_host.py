"""host-side cost of the pieces of one drop-in call (graph path): microseconds per call, device idle (synchronised between pieces)"""
import os, sys, types, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from differentiable_ransac_amd import layers, synth
dev = torch.device('cuda:0')
N, B = 2000, 1024
d = synth.batch_two_view(4, N)
m, lg, K1, K2 = (d[k].to(dev) for k in ("matches", "logits", "K1", "K2"))
im = torch.tensor([1000.0, 1000.0], device=dev)
opt = types.SimpleNamespace(fmat=False, sampler=2, ransac_batch_size=B, tr=False, weighted=0, threshold=0.75, precision=1, device=str(dev))
layer = layers.RANSACLayer(opt)
rn = layer.estimator
for _ in range(5): layer(m[0], lg[0], K1[0], K2[0], im, im, None)
g = next(iter(rn._graphs.values()))
def t(fn, n=300):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize()
    return (t1 - t0) / n * 1e6, (time.perf_counter() - t0) / n * 1e6
print("issue us / with final sync us")
print("points.clone()            ", t(lambda: m[0].clone()))
print("_fused_solver()           ", t(lambda: rn._fused_solver()))
print("staging _foreach_copy_    ", t(lambda: torch._foreach_copy_([g.matches[0], g.logits[0], g.K1[0], g.K2[0]], [m[0], lg[0], K1[0], K2[0]])))
print("graph replay              ", t(lambda: g.step()))
print("graph replay raw          ", t(lambda: g.step.graph.replay()))
pk = g.step()
print("packed.clone()            ", t(lambda: pk.clone()))
def views():
    p = pk
    return p[:36].view(torch.float32).view(3, 3), p[36:40].view(torch.float32)[0], p[40:44].view(torch.int32)[0], p[44:].view(torch.bool)
print("four views                ", t(views))
print("_GraphedCall              ", t(lambda: g(m[0], lg[0], K1[0], K2[0])))
print("RANSAC.__call__           ", t(lambda: rn(m[0], lg[0], K1[0], K2[0], None)))
print("RANSACLayer.forward       ", t(lambda: layer(m[0], lg[0], K1[0], K2[0], im, im, None)))
print("graph nodes (kernels per replay): see rocprof; rounds", 5)
