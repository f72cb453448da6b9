"""target of the rocprofv3 pass over the drop-in per-pair loop: 32 pairs x N passes through layers.RANSACLayer.forward (test mode)"""
import os, sys, types, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from differentiable_ransac_amd import layers, synth
dev = torch.device('cuda:0')
pairs, N, B = 32, 2000, int(os.environ.get("DROPIN_RBS", 1024))
d = synth.batch_two_view(pairs, N)
m, lg, K1, K2 = (d[k].to(dev) for k in ("matches", "logits", "K1", "K2"))
im = torch.tensor([1000.0, 1000.0], device=dev)
opt = types.SimpleNamespace(fmat=False, sampler=2, ransac_batch_size=B, tr=False, weighted=0, threshold=0.75, precision=1, device=str(dev))
layer = layers.RANSACLayer(opt)
layer.estimator.graph = os.environ.get("DROPIN_GRAPH", "1") == "1"
if os.environ.get("DROPIN_HYPS"):
    layer.estimator.graph_hypotheses = tuple(int(x) for x in os.environ["DROPIN_HYPS"].split(","))
def one_pass():
    return [layer(m[p], lg[p], K1[p], K2[p], im, im, None)[0] for p in range(pairs)]
for _ in range(3): one_pass()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(int(os.environ.get("DROPIN_PASSES", 6))): one_pass()
torch.cuda.synchronize()
print("ms per pair", (time.perf_counter() - t0) / int(os.environ.get("DROPIN_PASSES", 6)) / pairs * 1e3)
# host-only cost: the same loop with the device idle between pairs is not measurable directly; time the issue alone
t0 = time.perf_counter()
one_pass()
t1 = time.perf_counter()
torch.cuda.synchronize()
print("issue time of one pass per pair (host, before the final sync)", (t1 - t0) / pairs * 1e3, "ms; with sync", (time.perf_counter() - t0) / pairs * 1e3)
