"""Randomised cross-checks of the round-6 sampler paths (not part of the test suite: run on the GPU box, `python scratch/fuzz_r6.py
[trials]`): the register kernel's selection on wave masks in all three of its forms -- two-logarithm index sets, one-logarithm index
sets, train mode with either form's soft-max statistics -- against torch.topk / torch.softmax on the noise the general kernel dumps for
the same seed (same Philox counters), at random shapes (1-40 pairs, 1-700 rows, 8-2048 points, k = 1-8) and logit styles."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from differentiable_ransac_amd import ops, synth

dev = "cuda"
T = int(sys.argv[1]) if len(sys.argv) > 1 else 40
g = torch.Generator().manual_seed(2606)
ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
fails = 0
ops._RACE_MIN = (1, 1)


def check(name, ok, info=""):
    global fails
    if not ok:
        fails += 1
        print("FAIL", name, info)


EDGE = [(3, 70, 3, 4), (2, 33, 8, 8), (4, 65, 5, 8), (1, 1, 1, 4), (5, 129, 4, 4), (2, 64, 8, 12), (40, 7, 2, 2048)]   # (P, B, k, N): rows as short as k
for t in range(T + len(EDGE)):
    if t >= T:
        P, B, k, N = EDGE[t - T]
    else:
        P, B, k = ri(1, 40), ri(1, 700), ri(1, 8)
        N = 4 * ri(max(2, (k + 3) // 4 + 1), 512)
    d = synth.batch_two_view(P, N, seed0=100 + t)
    m, lg = d["matches"].to(dev), d["logits"].to(dev)
    style = ri(0, 5)
    if style == 1: lg = torch.zeros_like(lg)
    if style == 2: lg = torch.round(lg * 2) / 2
    if style == 3: lg = lg * 6.0
    if style == 4:
        lg = lg.clone(); lg[:, ri(0, N - 1)] += 28.0; lg[:, ri(0, N - 1)] += 27.0
    if style == 5: lg = lg / lg.abs().max() * 39.9
    seed = ri(0, 2 ** 40)
    r = ops.gumbel_topk(lg, B, k, 1.0, None, seed, want_noise=True)          # the general kernel: its noise, its own selection
    s = lg[:, None, :] + r["gumbel"]
    srt = torch.sort(s, dim=-1, descending=True)
    top = srt.indices[..., :k].sort(-1).values.int()
    gap_ok = (srt.values[..., k - 1] - srt.values[..., min(k, N - 1)]) > 1e-5 * srt.values[..., k - 1].abs().clamp_min(1.0)   # rows without a near-tie at the cut
    tag = f"t{t} P{P} B{B} N{N} k{k} style{style}"
    check(tag + " general", torch.equal(r["idx"][gap_ok], top[gap_ok]))
    for race in (False, True):
        i1, s1 = ops.gumbel_topk_gather(m, lg, B, k, 1.0, seed, race=race)
        check(tag + f" index race={race}", torch.equal(i1[gap_ok], top[gap_ok]), int((i1 != top).any(-1)[gap_ok].sum()))
        check(tag + f" sorted race={race}", bool((i1[..., 1:] > i1[..., :-1]).all()) if k > 1 else True)
        check(tag + f" samples race={race}", torch.equal(s1, torch.gather(m, 1, i1.reshape(P, B * k, 1).expand(-1, -1, 4).long()).reshape(P, B, k, 4)))
    soft = torch.softmax(s.double(), -1)
    lse = torch.logsumexp(s.double(), -1)
    for on in (False, True):
        ops.K1_RACE_SOFT = on
        smp, y, i2 = ops.SampleGather.apply(m, lg.contiguous(), B, k, 1.0, None, seed)
        check(tag + f" soft index race={on}", torch.equal(i2[gap_ok], top[gap_ok]), int((i2 != top).any(-1)[gap_ok].sum()))
        want = torch.gather(soft, 2, i2.long())
        ok = (i2 == top).all(-1)
        rel = ((y.double() - want).abs() / want.clamp_min(1e-300))[ok]
        check(tag + f" soft weights race={on}", rel.numel() == 0 or float(rel.max()) < 1e-4, float(rel.max()) if rel.numel() else 0)
    r2 = ops.gumbel_topk(lg, B, k, 1.0, None, seed)                           # soft, two-logarithm, no gather: lse
    check(tag + " lse", float((r2["lse"].double() - lse).abs().max()) < 2e-5 * max(1.0, float(lse.abs().max())))
print("fuzz_r6:", T, "random trials +", len(EDGE), "edge shapes,", fails, "failures")
