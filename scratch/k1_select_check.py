"""Round 6: the register sampler's selection on wave masks (DR_K1_SALU_SELECT) against the list selection of rounds 2-5 (a variant
library built with -DDR_K1_SALU_SELECT=0), every mode: index sets, samples, soft-max weights and log-sum-exps must be IDENTICAL; timing
of both at the headline shape.
   python scratch/build_variant.py k1_oldsel gumbel_topk.hip -DDR_K1_SALU_SELECT=0;  python scratch/k1_select_check.py"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if "--child" not in sys.argv:
    out = os.path.join(ROOT, "gpurun_out", "k1sel")
    os.makedirs(out, exist_ok=True)
    for name, lib in (("new", None), ("old", os.path.join(ROOT, "scratch", "libdransac_k1_oldsel.so"))):
        env = dict(os.environ)
        if lib: env["DRANSAC_LIB"] = lib
        subprocess.check_call([sys.executable, __file__, "--child", name, out], env=env)
    import torch
    a, b = torch.load(os.path.join(out, "new.pt")), torch.load(os.path.join(out, "old.pt"))
    bad = 0
    for key in a:
        same = all(torch.equal(x, y) for x, y in zip(a[key], b[key]))
        bad += not same
        print(f"{key:60s} {'identical' if same else 'DIFFERENT'}")
    print("ALL IDENTICAL" if not bad else f"{bad} CASES DIFFER")
    sys.exit(1 if bad else 0)
name, out = sys.argv[2], sys.argv[3]
sys.path.insert(0, ROOT)
import torch
from differentiable_ransac_amd import ops, synth
dev = torch.device("cuda:0")
res = {}
cases = [(128, 1024, 2000, 5), (32, 1024, 2000, 5), (3, 4096, 2048, 5), (7, 512, 512, 3), (5, 256, 1000, 8), (4, 300, 64, 5), (2, 2048, 2000, 1),
         (6, 640, 1996, 7), (9, 128, 260, 2)]
for (P, B, N, k) in cases:
    d = synth.batch_two_view(P, N)
    m, lg = d["matches"].to(dev), d["logits"].to(dev)
    for seed in (7, 99):
        i1, s1 = ops.gumbel_topk_gather(m, lg, B, k, 1.0, seed, race=True)
        res[f"P{P} B{B} N{N} k{k} seed{seed}"] = (i1.cpu(), s1.cpu())
        i0, _ = ops.gumbel_topk_gather(m, lg, B, k, 1.0, seed, race=False)
        diff = int((i1 != i0).any(-1).sum())
        assert (i1[..., 1:] > i1[..., :-1]).all() and i1.min() >= 0 and i1.max() < N
        print(f"[{name}] P{P} B{B} N{N} k{k} seed{seed}: rows differing from the two-logarithm form: {diff} of {P * B}")
    # the other modes of the register kernel: two-logarithm index sets; soft-max statistics (train mode), alone and with the gather
    i4, s4 = ops.gumbel_topk_gather(m, lg, B, k, 1.0, 13, race=False)
    res[f"P{P} B{B} N{N} k{k} two-log"] = (i4.cpu(), s4.cpu())
    r5 = ops.gumbel_topk(lg, B, k, 1.0, None, 13)
    res[f"P{P} B{B} N{N} k{k} soft"] = (r5["idx"].cpu(), r5["y_sel"].cpu(), r5["lse"].cpu())
    s6, y6, i6 = ops.SampleGather.apply(m, lg, B, k, 1.0, None, 13)
    res[f"P{P} B{B} N{N} k{k} soft+gather"] = (i6.cpu(), y6.cpu(), s6.cpu())
    r7 = ops.gumbel_topk(torch.round(lg), B, k, 1.0, None, 17)
    res[f"P{P} B{B} N{N} k{k} soft quantised"] = (r7["idx"].cpu(), r7["y_sel"].cpu(), r7["lse"].cpu())
    # sub-batched rows (super-rounds) and peaked / flat / tied logits
    i2, s2 = ops.gumbel_topk_gather(m, lg, B, k, 1.0, 5, race=True, sub=max(1, B // 4))
    res[f"P{P} B{B} N{N} k{k} sub"] = (i2.cpu(), s2.cpu())
    for tag, l2 in (("flat", torch.zeros_like(lg)), ("peaked", lg * 8.0), ("ties", torch.round(lg)), ("span79", lg / lg.abs().max() * 39.5)):
        i3, s3 = ops.gumbel_topk_gather(m, l2, B, k, 1.0, 11, race=True)
        res[f"P{P} B{B} N{N} k{k} {tag}"] = (i3.cpu(), s3.cpu())
torch.save(res, os.path.join(out, name + ".pt"))
P, B, N, k = 128, 1024, 2000, 5
d = synth.batch_two_view(P, N)
m, lg = d["matches"].to(dev), d["logits"].to(dev)
f = lambda: ops.gumbel_topk_gather(m, lg, B, k, 1.0, 7, race=True)
for _ in range(30): f()
ts = []
for rep in range(4):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50): f()
    b.record(); torch.cuda.synchronize()
    ts.append(a.elapsed_time(b) / 50 * 1e3)
print(f"[{name}] 128 x 1024 x 2000, k = 5, sampler + gather + weights prologue: " + " / ".join(f"{t:.1f}" for t in ts) + " us")
