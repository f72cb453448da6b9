#!/usr/bin/env python
"""Headline benchmark: hypotheses/s of the differentiable-RANSAC hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One *step* = one pass of the hot path (test mode, ransac.py:55-144, one RANSAC batch) over one batch of synthetic image
pairs resident in HBM: K1 Gumbel top-k sampling (in-kernel Philox) -> K2 gather -> K3 Nister 5-point -> K4 MSAC scoring
of all 10*B models against all N points (masks materialised, as MSACScore.score's contract requires) -> K6 per-pair
arg-max / best mask / inlier count.
Headline workload = BASELINE.json configs[1] ("c2"): Nister 5-pt, N = 2000 points, B = 1024 hypotheses per pair, Gumbel
sampler, MSAC; `--pairs` pairs per GPU per step (weak scaling: pairs shard across ranks, no data-path collective --
SURVEY 8(e)).  Prints ONE JSON line on rank 0.

The other BASELINE configs are measured after the headline region (N = 1 only) and reported under "configs":
  c1  8-point F, 128 points, 64 hypotheses, uniform sampler (256 pairs per step: one pair is launch-bound)
  c3  Stewenius 5-point, 2000 points, 4096 hypotheses, 32 pairs
  c4  rigid SVD, 50 000 points, 2048 hypotheses, one pair
`--workload c1|c3|c4` makes one of them the timed region instead (for profiling).
`--mode train`: sampler -> solver -> best-of-10 vs GT -> MatchLoss -> backward to the logits; with N > 1 every step carries
the training step's one collective (train.py:150-175): a flat RCCL all-reduce of the scores network's gradient bucket
(622 616 f32, the reference's CLNet; the per-pair logits gradient stays on its rank), issued asynchronously and waited for
after the next step's launches have been enqueued (`collective_ms` / `collective_exposed_ms`).  `--split hypotheses`: fewer pairs than GPUs -- every
rank draws B / N hypotheses for the SAME pairs and the per-pair winners are merged (strong scaling).
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
FP32_PEAK_TFLOPS = 157.3   # packed-f32 VALU = f32 MFMA dense peak
CLNET_PARAMS = 622616      # parameters of the reference's scores network
# the reference's own code imported in the survey container (8 CPU cores, torch 2.10 CPU, f32,
# no_grad, the same synthetic recipe) -- hypotheses/s per BASELINE config.  Quoted next to cpu_baseline so that nobody
# compares the GPU figure with the (faster, vectorised) oracle alone.
REFERENCE_IMPORT = {"c1": 1.6e3, "c2": 1.05e3, "c3": 3.1e3, "c4": 0.68e3}

WORKLOADS = {
    "c1": dict(solver="f8", pairs=256, points=128, hyps=64, sampler="uniform", baseline_config=0,
               text="8-point F, 128 pts x 64 hyps per pair, uniform sampler, MSAC with masks"),
    # 128 pairs per step since round 2: the step's kernels are grid-size limited at 32 pairs (K3 has one wave per SIMD, K4
    # 2.5 rounds of blocks); measured 80.1 / 90.9 / 98.0 / 91.4 M hypotheses/s at 32 / 64 / 128 / 256 pairs per step
    # (2.6 GB of masks at 128).  The 32-pair figure of round 1 stays in the line as configs["c2_p32"].
    "c2": dict(solver="nister", pairs=128, points=2000, hyps=1024, sampler="gumbel", baseline_config=1,
               text="nister 5-pt E, 2000 pts x 1024 hyps per pair, Gumbel top-k sampler (in-kernel Philox), MSAC scoring with masks"),
    "c3": dict(solver="stewenius", pairs=32, points=2000, hyps=4096, sampler="gumbel", baseline_config=2,
               text="stewenius 5-pt E, 2000 pts x 4096 hyps per pair, Gumbel top-k sampler, MSAC scoring with masks"),
    "c4": dict(solver="rigid", pairs=1, points=50000, hyps=2048, sampler="gumbel", baseline_config=3,
               text="rigid SVD (3-D registration), 50000 pts x 2048 hyps, Gumbel top-k sampler, squared residuals with masks"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS), help="which BASELINE config is the timed region")
    ap.add_argument("--pairs", type=int, default=None, help="image pairs per GPU per step (default: the workload's)")
    ap.add_argument("--points", type=int, default=None)
    ap.add_argument("--hyps", type=int, default=None)
    ap.add_argument("--solver", default=None, choices=["nister", "stewenius", "f8"], help="(legacy) overrides the workload's solver")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the sub-records of the other BASELINE configs")
    ap.add_argument("--cpu-seconds", type=float, default=6.0, help="budget of the bounded CPU-baseline sample (the other configs share twice this)")
    ap.add_argument("--profile-kernels", action="store_true", help="per-kernel HIP-event breakdown (extra syncs)")
    ap.add_argument("--mode", default="test", choices=["test", "train"])
    ap.add_argument("--split", default="pairs", choices=["pairs", "hypotheses"],
                    help="pairs: every rank owns its own pairs (weak scaling); hypotheses: every rank draws hyps/N hypotheses "
                         "for the SAME pairs and the winners are merged with two tiny all_gathers (strong scaling, P < G)")
    ap.add_argument("--no-extras", dest="extras", action="store_false",
                    help="skip the informational regions after the timed one (two batches in flight on two streams; the step "
                         "followed by the final refit; the step with the top-down sampler)")
    ap.add_argument("--extras", dest="extras", action="store_true", help="(default) kept for older command lines")
    ap.set_defaults(extras=True)
    ap.add_argument("--sampler", default=None, choices=["gumbel", "topdown", "uniform"])
    ap.add_argument("--logits-fixture", action="store_true",
                    help="timed region on tests/golden/clnet_logits.npz: reader-produced pairs scored by the REFERENCE's network "
                         "with its shipped weights (generated in the build container by tests/golden/gen_clnet_logits.py), "
                         "tiled to --pairs; the default run reports the same thing as the sub-record `clnet_logits`")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend.  nccl = RCCL over xGMI (the real thing).  gloo exists to run the N > 1 code paths "
                         "on a box with ONE GPU (RCCL refuses two ranks on one device): --gpus-shared lets the ranks share it")
    ap.add_argument("--gpus-shared", action="store_true", help="functional check only: every rank uses cuda:0")
    ap.add_argument("--graph", choices=("auto", "on", "off"), default="auto",
                    help="HIP-graph replay of the step (sampler seeds advance on the device, so every replay draws fresh "
                         "hypotheses).  auto: the train step and the sub-records of the other configs are replayed; the "
                         "headline test-mode region stays eager, because its roofline needs HIP events around the scoring "
                         "launch of every step (it is device-bound either way)")
    ap.add_argument("--segments", type=int, default=5,
                    help="the timed region is repeated as this many segments of EXACTLY --steps steps, each bracketed by a "
                         "barrier + synchronize on both sides; the line reports the median segment (value, ms_per_step, the "
                         "scoring launch's HIP-event mean) and lists all of them")
    ap.add_argument("--prewarm-s", type=float, default=0.6,
                    help="time-based pre-conditioning before the counted warm-up: the same step is issued until this many "
                         "seconds of wall time have passed (clocks, allocator pools, caches), reported as prewarm_s.  A "
                         "20-step run is 24 ms of device time: without this it measures the power-state ramp, not the step")
    ap.add_argument("--rendezvous-only", action="store_true",
                    help="launcher check (no GPU needed): start the ranks, form the process group, count them, print the line")
    ap.add_argument("--streams", type=int, default=1,
                    help="HIP streams the K timed steps are issued on round-robin.  Default 1: strictly one kernel at a "
                         "time, so that the HIP-event duration of the scoring kernel in the timed region is its own "
                         "(roofline attribution).  Two batches in flight are always measured after the timed region and "
                         "reported as `two_batches_in_flight`")
    return ap.parse_args()


def resolve(args):
    w = dict(WORKLOADS[args.workload])
    if args.mode == "train" and args.workload == "c2":
        w["pairs"] = 32                      # BASELINE configs[4]: 256 pairs over 8 GPUs = 32 per GPU and step
    if args.solver:
        w["solver"] = args.solver
    for k in ("pairs", "points", "hyps", "sampler"):
        v = getattr(args, k)
        if v is not None:
            w[k] = v
    return w


# ----------------------------------------------------------------------------------------------------------------------
def load_logits_fixture(P):
    """tests/golden/clnet_logits.npz tiled to P pairs: matches, the network's log-probabilities (`-p 2`, the reference's
    default input of the sampler), intrinsics, the geometric inlier mask."""
    import numpy as np
    z = np.load(os.path.join(ROOT, "tests", "golden", "clnet_logits.npz"))
    rep = (P + z["matches"].shape[0] - 1) // z["matches"].shape[0]
    t = lambda k: torch.from_numpy(z[k]).repeat(rep, *([1] * (z[k].ndim - 1)))[:P].contiguous()
    return {"matches": t("matches"), "logits": t("log_probs"), "K1": t("K1"), "K2": t("K2"), "gt_E": t("gt_E"),
            "inliers": t("geometric_inliers"), "gt_F": t("gt_E")}


def make_step(w, dev, rank=0, mode="test", seed=1234, keep_masks=True, fixture=False, device_seeds=False):
    """Builds the resident inputs of a workload and returns (step callable, info dict).  device_seeds: the drivers advance
    their sampler seed on the device, which makes the step capturable in a HIP graph (differentiable_ransac_amd.graphs)."""
    from differentiable_ransac_amd import synth
    from differentiable_ransac_amd.ransac import BatchedRANSAC, BatchedRANSAC3D
    P, N, B = w["pairs"], w["points"], w["hyps"]
    if w["solver"] == "rigid":
        items = [synth.rigid_pair(rank * P + p, N) for p in range(P)]
        matches = torch.stack([it["matches"] for it in items]).to(dev)
        logits = torch.stack([it["logits"] for it in items]).to(dev)
        rn = BatchedRANSAC3D(ransac_batch_size=B, train=False, threshold=0.03, max_iterations=B, seed=seed + rank, flag=False,
                             keep_masks=keep_masks)
        if device_seeds:
            rn.device_seeds(dev)

        def step():
            return rn(matches, logits)
        return step, dict(rn=rn, matches=matches, logits=logits, S=1, data=None, K=(None, None))
    data = load_logits_fixture(P) if fixture else synth.batch_two_view(P, N, seed0=rank * P, pixel=(w["solver"] == "f8"))
    matches, logits = data["matches"].to(dev), data["logits"].to(dev)
    K1, K2 = data["K1"].to(dev), data["K2"].to(dev)
    S = 1 if w["solver"] == "f8" else 10
    if mode == "train":
        from differentiable_ransac_amd.loss import MatchLoss
        gt = data["gt_E"].to(dev) if w["solver"] != "f8" else data["gt_F"].to(dev)
        tr = BatchedRANSAC(w["solver"], ransac_batch_size=B, train=True, max_iterations=B, seed=99 + rank)
        if device_seeds:
            tr.device_seeds(dev)
        lg = logits.clone().requires_grad_(True)
        match_loss = MatchLoss()                     # the reference's default training loss (-w2 1, train.py:70-79)
        gt_mask = data["inliers"].to(dev)
        no_inliers = torch.zeros(P, device=dev)      # the train step reports no inlier counts (allocated once, not per step)
        one = torch.ones((), device=dev)             # the root gradient of loss.backward(), allocated once (not a fill per step)

        def step():
            lg.grad = None
            chosen, keep = tr(matches, lg, gt_model=gt)
            if w["solver"] == "f8":                  # pixel coordinates: plain distance to the ground-truth F
                d = torch.minimum(((chosen - gt[:, None]) ** 2).sum((-1, -2)), ((chosen + gt[:, None]) ** 2).sum((-1, -2)))
                loss = (d * keep).sum()
            else:
                loss = match_loss(chosen, matches, gt_mask, keep)
            loss.backward(one)
            return {"inliers": no_inliers, "grad": lg.grad}
        return step, dict(rn=tr, matches=matches, logits=lg, S=S, data=data, K=(K1, K2))
    rn = BatchedRANSAC(w["solver"], ransac_batch_size=B, train=False, threshold=0.75, max_iterations=B, seed=seed + rank,
                       keep_masks=keep_masks, refit=False, sampling=w["sampler"])
    if device_seeds:
        rn.device_seeds(dev)

    def step():
        return rn(matches, logits, K1, K2)
    return step, dict(rn=rn, matches=matches, logits=logits, S=S, data=data, K=(K1, K2))


def run_bounded(fn, n, window=32, stride=4):
    """fn(i) for i < n with at most `window` steps queued ahead of the device: step i is issued when step i - window (rounded to the
    stride) has finished.  Round 6: the event is recorded after every `stride`-th step only -- a record is a marker packet with a
    barrier, 5.6 us of idle device (profiles/r6_headline_timeline.md): one per step was 16 % of config 1's 35 us step.  fn may switch
    streams itself: it then returns the stream to record on, and every step carries a record (each stream needs its own)."""
    ev = [torch.cuda.Event() for _ in range(window)]
    for i in range(n):
        if i >= window and i % stride == 0:
            ev[((i - window) // stride) % window].synchronize()        # recorded after step i - window + stride - 1
        st = fn(i)
        own = isinstance(st, torch.cuda.Stream)
        if own:
            stride = 1
        if (i + 1) % stride == 0:
            ev[(i // stride) % window].record(st if own else torch.cuda.current_stream())


K4_EVENT_EVERY = 4   # the scoring launch's HIP-event pair: every 4th step of the timed region (a pair costs the step ~11 us of idle device)


class CallTimer:
    """HIP events around selected libdransac launches (the ctypes call), recorded on the stream the launch goes to."""

    def __init__(self, prefixes, slots):
        from differentiable_ransac_amd import _lib as L
        self.L, self.prefixes = L, tuple(prefixes)
        self.ev = [[torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)] for _ in range(slots)]
        self.i = -1
        self.used = set()                 # slots whose pair was recorded (the timed region samples: every K4_EVENT_EVERY-th step)
        self.orig = L.call
        L.call = self._call

    def _call(self, name, *a):
        i = self.i
        if 0 <= i < len(self.ev) and name.startswith(self.prefixes):
            self.ev[i][0].record()
            self.orig(name, *a)
            self.ev[i][1].record()
            self.used.add(i)
        else:
            self.orig(name, *a)

    def mean_ms(self, n=None, lo=0):
        idx = [i for i in range(lo, len(self.ev) if n is None else lo + n) if i in self.used]
        return sum(self.ev[i][0].elapsed_time(self.ev[i][1]) for i in idx) / max(1, len(idx))

    def close(self):
        self.L.call = self.orig


def per_call_breakdown(step, reps=10):
    """Device time of every libdransac entry point of one step (events around each ctypes call; one stream): per entry the sum
    over its calls within a step, MEDIAN over `reps` steps (a host hiccup between an event and its launch lands in the mean)."""
    from differentiable_ransac_amd import _lib as L
    orig = L.call
    rec = []

    def call(name, *a):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        orig(name, *a)
        e1.record()
        rec[-1].append((name, e0, e1))

    step()
    torch.cuda.synchronize()
    L.call = call
    try:
        for _ in range(reps):
            rec.append([])
            step()
        torch.cuda.synchronize()
    finally:
        L.call = orig
    per_rep = []
    for calls in rec:
        d = {}
        for name, a, b in calls:
            d[name] = d.get(name, 0.0) + a.elapsed_time(b)
        per_rep.append(d)
    return {name: sorted(d.get(name, 0.0) for d in per_rep)[len(per_rep) // 2] for name in per_rep[0]}


def k4_bytes(P, N, M):
    return P * (16 * N + 36 * M + 4 * M + M * N)          # SURVEY 8(d), masks included (all M rows are written)


def k4r_bytes(P, N, M):
    return P * (24 * N + 48 * M + 4 * M + 4 + M * N)     # rigid residuals with masks


def config_record(key, dev, steps, warmup, pairs=None, graph=True):
    """Sub-record of one BASELINE config: ms/step, hypotheses/s, the dominant libdransac launch and its roofline share."""
    w = dict(WORKLOADS[key])
    if pairs is not None:
        w["pairs"] = pairs
    step, info = make_step(w, dev)
    for _ in range(max(warmup, 20)):
        step()
    torch.cuda.synchronize()
    # equal segments, the median one is reported (a sub-record shares the process with everything measured before
    # it -- allocator state, clocks -- and one slow segment should not stand for the config)
    seg = max(1, steps // 3)

    def segments(fn):
        # five segments, median reported (round 6: three were not enough for config 3 -- a step allocates 2.6 GB of masks and one
        # allocator stall of ~40 ms inside a 20-step segment moved two of the three)
        out = []
        for _ in range(5):
            t0 = time.perf_counter()
            run_bounded(lambda i: fn() and None, seg)
            torch.cuda.synchronize()
            out.append((time.perf_counter() - t0) / seg * 1e3)
        return out
    eager_ms = segments(step)
    seg_ms, issue, graph_ms = eager_ms, "eager: one Python call per launch", None
    if graph:
        # the same step (same launches, fresh hypotheses per replay: the seed advances on the device) replayed as ONE HIP graph
        from differentiable_ransac_amd.graphs import GraphedStep
        gstep = GraphedStep(make_step(w, dev, device_seeds=True)[0])
        for _ in range(5):
            gstep()
        torch.cuda.synchronize()
        graph_ms = segments(gstep)
        del gstep
        # launch-bound steps (config 1: 0.05 ms of device time in six launches) gain from the replay, device-bound ones lose a
        # few per cent to the graph's inter-node barriers.  The sub-record always carries BOTH figures, labelled; which one
        # `ms_per_step` quotes is fixed per config (not the minimum of the two): the replay for the launch-bound c1 and for
        # the one-pair call of c2, the eager issue for the device-bound others
        if key == "c1" or (key == "c2" and w["pairs"] == 1):
            seg_ms, issue = graph_ms, "HIP graph replay of the whole step (differentiable_ransac_amd.graphs.GraphedStep)"
    el, steps = sorted(seg_ms)[len(seg_ms) // 2] * 1e-3 * seg, seg
    calls = per_call_breakdown(step)
    dom = max(calls, key=calls.get)
    P, N, B = w["pairs"], w["points"], w["hyps"]
    M = B * info["S"]
    rec = {"baseline_config_index": w["baseline_config"], "workload": f"{w['text']}, {P} pair(s) per step",
           "steps": steps, "issue": issue, "segments_ms_per_step": [round(x, 5) for x in seg_ms],
           "ms_per_step": el / steps * 1e3, "hypotheses_per_s": P * B * steps / el,
           "eager_ms_per_step": sorted(eager_ms)[len(eager_ms) // 2], "graph_replay_ms_per_step": sorted(graph_ms)[len(graph_ms) // 2] if graph else None,
           "pairs_per_s": P * steps / el, "launch_ms": {k: round(v, 5) for k, v in sorted(calls.items(), key=lambda kv: -kv[1])},
           "dominant_launch": dom, "dominant_ms": calls[dom],
           "reference_import_hypotheses_per_s": REFERENCE_IMPORT.get(key)}
    score_call = "dr_msac_score_f32"
    if w["solver"] == "rigid":
        score_call = "dr_rigid_residual_f32"
    if score_call in calls:
        nbytes = k4r_bytes(P, N, M) if w["solver"] == "rigid" else k4_bytes(P, N, M)
        ach = nbytes / (calls[score_call] * 1e-3) / 1e9
        rec["scoring_roofline"] = {"bound": "hbm", "launch": score_call, "avg_launch_ms": calls[score_call],
                                   "algorithmic_bytes_per_launch": nbytes, "achieved": ach, "peak": HBM_PEAK_GBS,
                                   "unit": "GB/s", "frac": ach / HBM_PEAK_GBS}
        if w["solver"] == "rigid":
            # Round 6: an event pair around ONE launch of a 43 us kernel also times the two marker packets and the dispatch gap
            # between them (the round-5 review: 49.7 us here against 43.2 us by rocprofv3 for the same kernel).  The same launch 50
            # times between ONE event pair amortises that; `event_pair_overhead_ms` is what an event pair around a one-thread kernel
            # reads.  `frac` above stays the single-launch figure; `back_to_back` is the one comparable with rocprofv3's duration.
            from differentiable_ransac_amd import ops
            rn3, m3 = info["rn"], info["matches"]
            idx3 = ops.gumbel_topk(info["logits"], B, 3, 1.0, None, 1, soft=False)["idx"]
            res3 = torch.zeros((P, B), device=m3.device, dtype=torch.float32)
            model3, _ = ops.solve_rigid_gather(m3, idx3, rn3.flag, zero_sums=res3)
            seedw = ops.DeviceSeed(0, m3.device)

            def timed(fn, n):
                for _ in range(5):
                    fn()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(n):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                return e0.elapsed_time(e1) / n
            b2b = timed(lambda: ops.rigid_residual(m3, model3, rn3.threshold, True, res=res3), 50)
            pairs_ = []
            for _ in range(20):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                seedw.next()
                e1.record()
                pairs_.append((e0, e1))
            torch.cuda.synchronize()
            ov = sorted(a.elapsed_time(b) for a, b in pairs_)[10]
            ach2 = nbytes / (b2b * 1e-3) / 1e9
            rec["scoring_roofline"]["back_to_back"] = {"avg_launch_ms": b2b, "launches": 50, "achieved": ach2, "frac": ach2 / HBM_PEAK_GBS,
                                                       "event_pair_overhead_ms": ov,
                                                       "note": "50 launches between one HIP-event pair; the single-launch figure "
                                                               "above includes the event pair's own overhead"}
    return rec


def _median3_ms(fn, seg):
    out = []
    for _ in range(3):
        t0 = time.perf_counter()
        run_bounded(lambda i: fn() and None, seg)
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / seg * 1e3)
    return sorted(out)[1], out


def dropin_layer_loop_record(dev, pairs=32, reps=6):
    """The reference's own call pattern (model_cl.py:488-511, test.py:38): the pairs of a batch pushed ONE BY ONE through
    `layers.RANSACLayer.forward` -- the five-line import swap of INTEGRATION.md section 1, nothing batched by the caller.  Test mode
    (`-tr 0`: max_iters = 5000 = up to five batches of `-rbs 1024`, adaptive stop, final refit K7) and train mode (`-tr 1`:
    one batch, best-of-ten vs the ground truth, autograd graph kept).  Wall time per pair over `reps` passes over the batch,
    device-synchronised once per pass (nothing inside the loop reads a result back)."""
    import types
    from differentiable_ransac_amd import layers, synth
    from differentiable_ransac_amd.ransac import BatchedRANSAC
    N, B = 2000, 1024
    d = synth.batch_two_view(pairs, N)
    m, lg, K1, K2, gt = (d[k_].to(dev) for k_ in ("matches", "logits", "K1", "K2", "gt_E"))
    im = torch.tensor([1000.0, 1000.0], device=dev)
    out = {"workload": f"{pairs} pairs x {N} points pushed one by one through layers.RANSACLayer.forward, -rbs {B}, sampler 2 "
                       "(Gumbel), Nister; model_cl.py:488-511"}

    def run(layer, train):
        def one_pass():
            res = []
            for p in range(pairs):
                w = lg[p].clone().requires_grad_(True) if train else lg[p]
                Es, _ = layer(m[p], w, K1[p], K2[p], im, im, gt[p] if train else None)
                res.append(Es)
            return res
        for _ in range(3):
            res = one_pass()
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            res = one_pass()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / pairs * 1e3)
        return sorted(ts)[len(ts) // 2], res
    opt = types.SimpleNamespace(fmat=False, sampler=2, ransac_batch_size=B, tr=False, weighted=0, threshold=0.75, precision=1,
                                device=str(dev))
    layer = layers.RANSACLayer(opt)
    ms, res = run(layer, False)
    out["test_mode"] = {"ms_per_pair": ms, "hypotheses_per_s": B / (ms * 1e-3), "issue": "one replayed HIP graph per pair: device "
                        "rounds of %s hypotheses walked batch by batch with the loop's stop rule on the device (RANSAC.graph, "
                        "graph_hypotheses, ransac._GraphedCall), final refit included" % (layer.estimator.graph_hypotheses,)}
    # the layer's second return value is a wall time (test.py:100 averages it): timing = "sync" waits for the pair's result
    layer.timing = "sync"
    ms_s, _ = run(layer, False)
    layer.timing = "enqueue"
    out["test_mode_sync_timing"] = {"ms_per_pair": ms_s, "issue": "the same with RANSACLayer.timing = 'sync': every call waits for "
                                    "its result, so that the returned seconds are the call's wall time as upstream"}
    layer.estimator.graph = False
    ms_e, _ = run(layer, False)
    out["test_mode_eager"] = {"ms_per_pair": ms_e, "issue": "eager launches, termination read back after every batch (rounds 1-4)"}
    # the reference's DEFAULT batch size (utils.py:33 `-rbs 64`; ransac.py:8-39 ransac_batch_size=64): up to 79 batches per call
    opt64 = types.SimpleNamespace(**{**vars(opt), "ransac_batch_size": 64})
    layer64 = layers.RANSACLayer(opt64)
    ms64, _ = run(layer64, False)
    out["dropin_layer_loop_rbs64"] = {"ms_per_pair": ms64, "ransac_batch_size": 64,
                                      "device_rounds": layer64.estimator._graph_rounds,
                                      "issue": "one replayed HIP graph per pair: 79 batches of 64 as two device rounds (32 + 47 "
                                               "sub-batches), walked in order on the device"}
    # the same pairs through ONE batched call (what batched_forward does): the device time the loop competes with
    drv = BatchedRANSAC("nister", ransac_batch_size=B, threshold=0.75, max_iterations=5000, refit=True)
    for _ in range(3):
        drv(m, lg, K1, K2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        drv(m, lg, K1, K2)
    torch.cuda.synchronize()
    out["batched_forward_ms_per_pair"] = (time.perf_counter() - t0) / reps / pairs * 1e3
    opt_t = types.SimpleNamespace(**{**vars(opt), "tr": True})
    ms_t, res_t = run(layers.RANSACLayer(opt_t), True)
    out["train_mode"] = {"ms_per_pair": ms_t, "issue": "eager (the autograd tape of a call cannot be replayed); forward only, "
                         "models per pair: %d" % int(res_t[0].shape[0])}
    return out


def train_record(dev, steps=150, pairs=32):
    """BASELINE configs[4]'s per-GPU share as a sub-record of the default line: the train step (sampler -> solver -> best-of-10
    vs GT -> MatchLoss -> backward to the logits) on 32 pairs x 2000 points x 1024 hypotheses, eager and replayed as one HIP
    graph, with the device time of every libdransac launch of the eager step (forward and backward entries)."""
    from differentiable_ransac_amd.graphs import GraphedStep
    w = dict(WORKLOADS["c2"], pairs=pairs)
    step, _ = make_step(w, dev, mode="train")
    for _ in range(20):
        out = step()
    torch.cuda.synchronize()
    eager_ms, eager_all = _median3_ms(step, steps)
    calls = per_call_breakdown(step)
    gstep = GraphedStep(make_step(w, dev, mode="train", device_seeds=True)[0])
    for _ in range(5):
        gout = gstep()
    torch.cuda.synchronize()
    graph_ms, graph_all = _median3_ms(gstep, steps)
    B = w["hyps"]
    return {"baseline_config_index": 4,
            "workload": f"nister train step (sample, solve, best-of-10 vs GT, MatchLoss, backward to the logits), {w['points']} pts x "
                        f"{B} hyps per pair, {pairs} pairs = one GPU's share of 256 pairs over 8 GPUs (no collective at N = 1)",
            "issue": "HIP graph replay of forward + loss + backward (one launch per step)",
            "steps": steps, "ms_per_step": graph_ms, "hypotheses_per_s": pairs * B / (graph_ms * 1e-3),
            "pairs_per_s": pairs / (graph_ms * 1e-3), "graph_replay_segments_ms": [round(x, 5) for x in graph_all],
            "eager_ms_per_step": eager_ms, "eager_segments_ms": [round(x, 5) for x in eager_all],
            "launch_ms": {k: round(v, 5) for k, v in sorted(calls.items(), key=lambda kv: -kv[1])},
            "device_ms_sum_of_launches": sum(calls.values()),
            "grad_finite": bool(torch.isfinite(out["grad"]).all() and torch.isfinite(gout["grad"]).all())}


def fused_driver_record(dev, steps=100):
    """The driver as a user runs it: BatchedRANSAC's default keep_masks=False -- scores + ONE best mask per pair, no [M,N] mask
    tensor.  Its own byte accounting (16N + 40M + N per pair), never mixed with the masks-on accounting of the
    headline; the scoring launch is priced against BOTH roofs."""
    w = dict(WORKLOADS["c2"])
    step, info = make_step(w, dev, keep_masks=False)
    for _ in range(20):
        step()
    torch.cuda.synchronize()
    ms, all_ms = _median3_ms(step, steps)
    calls = per_call_breakdown(step)
    P, N, B = w["pairs"], w["points"], w["hyps"]
    M = B * info["S"]
    with torch.no_grad():
        _, v_, _ = info["rn"].hypotheses(info["matches"], info["logits"])
    vf = float(v_.float().mean())
    k4 = calls.get("dr_msac_score_f32")
    nbytes = P * (16 * N + 40 * M + N)
    rec = {"workload": f"{w['text'].replace(' with masks', '')}, keep_masks=False (scores + one best mask per pair), {P} pairs per step",
           "steps": steps, "ms_per_step": ms, "segments_ms_per_step": [round(x, 5) for x in all_ms],
           "hypotheses_per_s": P * B / (ms * 1e-3), "pairs_per_s": P / (ms * 1e-3),
           "launch_ms": {k: round(v, 5) for k, v in sorted(calls.items(), key=lambda kv: -kv[1])}}
    if k4:
        fl = 39.0 * P * M * N * vf
        rec["scoring_roofline"] = {"bound": "valu_f32", "launch": "dr_msac_score_f32 (masks == NULL)", "avg_launch_ms": k4,
                                   "algorithmic_bytes_per_launch": nbytes, "bytes_formula": "P (16 N + 40 M + N)",
                                   "hbm_achieved_GBs": nbytes / (k4 * 1e-3) / 1e9, "hbm_frac": nbytes / (k4 * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                   "valid_slot_fraction": vf, "valu_tflops": fl / (k4 * 1e-3) / 1e12,
                                   "valu_frac_of_157.3": fl / (k4 * 1e-3) / 1e12 / FP32_PEAK_TFLOPS,
                                   "note": "without the mask stream the launch moves 0.5 MB per pair: the HBM roof is irrelevant, "
                                           "the f32 vector ALU is the only roof (39 flop per evaluated (model, point))"}
    return rec


def k4_all_valid_record(dev, launches=30):
    """The scoring kernel's intrinsic rate: the headline shape with EVERY slot evaluated (valid == NULL: the identity fillers of
    the non-real roots are scored like any model), next to the same launch with the solver's validity flags.  The headline's
    roofline fraction counts the mask bytes of skipped slots too (55 % of the rows are zero rows at this workload)."""
    from differentiable_ransac_amd import ops
    w = dict(WORKLOADS["c2"])
    _, info = make_step(w, dev)
    P, N, B = w["pairs"], w["points"], w["hyps"]
    with torch.no_grad():
        models, valid, _ = info["rn"].hypotheses(info["matches"], info["logits"])
    flat = models.reshape(P, -1, 3, 3).contiguous()
    vflat = valid.reshape(P, -1).contiguous()
    thr = torch.full((P,), 7.5e-4, device=dev)
    M = flat.shape[1]
    nbytes = k4_bytes(P, N, M)

    def t(v):
        for _ in range(5):
            ops.msac_score(info["matches"], flat, thr, True, v)
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(launches)]
        for a, b in ev:
            a.record()
            ops.msac_score(info["matches"], flat, thr, True, v)
            b.record()
        torch.cuda.synchronize()
        return sorted(a.elapsed_time(b) for a, b in ev)[launches // 2]
    ms_all, ms_flag = t(None), t(vflat)
    vf = float(vflat.float().mean())
    out = {}
    for key, ms, frac_eval in (("all_slots_valid", ms_all, 1.0), ("solver_validity_flags", ms_flag, vf)):
        fl = 39.0 * P * M * N * frac_eval
        out[key] = {"avg_launch_ms": ms, "evaluated_slot_fraction": frac_eval, "algorithmic_bytes_per_launch": nbytes,
                    "hbm_achieved_GBs": nbytes / (ms * 1e-3) / 1e9, "hbm_frac": nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "valu_tflops": fl / (ms * 1e-3) / 1e12, "valu_frac_of_157.3": fl / (ms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS}
    out["workload"] = f"scoring launch alone, {P} pairs x {M} slots x {N} points, masks on, median of {launches} launches (HIP events)"
    return out


# ----------------------------------------------------------------------------------------------------------------------
CPU_THREAD_COUNTS = (1, 8, 32)


def _cpu_rate(unit_fn, hyps_per_unit, budget_s, min_reps=3, max_reps=64):
    """The host's best for one workload unit (round 6): unit_fn(i) -> seconds, timed at 1, 8 and 32 threads (capped at the core count;
    intra-op threads pinned by torch.set_num_threads, inter-op threads at 1), at least `min_reps` repetitions per thread count inside
    an equal share of the budget; the rate of a thread count is hypotheses per MEDIAN unit time, `value` the best of them -- with its
    thread count, the spread (max - min) / median of its repetitions, and every thread count's figure beside it.  A fixed 32 threads
    (round 5) was the wrong choice for the small configs and a single pass at one count did not repeat from box to box."""
    cores = os.cpu_count() or 1
    counts = sorted({max(1, min(t, cores)) for t in CPU_THREAD_COUNTS})
    per, total = {}, 0.0
    for n in counts:
        torch.set_num_threads(n)
        if n > 1:
            unit_fn(0)                    # the thread pool's first use at this size
        times, used, share = [], 0.0, budget_s / len(counts)
        while len(times) < max_reps and (len(times) < min_reps or used < share):
            dt = unit_fn(len(times))
            times.append(dt)
            used += dt
            if used >= 2.0 * share:       # a slow host: fewer than min_reps repetitions rather than a bench run of minutes
                break
        total += used
        med = sorted(times)[len(times) // 2]
        per[n] = {"rate": hyps_per_unit / max(med, 1e-9), "reps": len(times), "spread": (max(times) - min(times)) / max(med, 1e-9),
                  "seconds": used}
    best = max(per, key=lambda n: per[n]["rate"])
    return {"value": per[best]["rate"], "unit": "hypotheses/s", "cores": best, "threads": best, "kind": "port", "host_cores": cores,
            "spread": round(per[best]["spread"], 3), "reps": per[best]["reps"],
            "by_threads": {str(n): round(per[n]["rate"], 1) for n in counts},
            "single_thread_value": per[counts[0]]["rate"], "seconds": round(total, 1)}


def _pin_cpu_threads():
    try:
        torch.set_num_interop_threads(1)  # (once per process, before any inter-op work: later calls raise)
    except RuntimeError:
        pass


def cpu_baseline(args, w, pairs_data):
    """The CPU oracle (oracle/cpu_ref.py, a vectorised torch restatement of the reference path) timed on the host
    cores, on a bounded sample of the same workload: whole pairs (N points x B hypotheses), one after the other like
    the reference's per-pair loop (model_cl.py:488).  Round 6: the best MEDIAN over 1 / 8 / 32 threads (_cpu_rate) -- the fixed 32
    threads of round 5 lost to one thread on the small configs and a single pass did not repeat (4 052 / 6 435 / 6 001 / 7 747
    hypotheses/s for the same port over the boxes of rounds 3-5)."""
    from oracle import cpu_ref as O
    from differentiable_ransac_amd import synth
    cores = os.cpu_count() or 1
    solver, N, B = w["solver"], w["points"], w["hyps"]
    k = 8 if solver == "f8" else 5
    noise_cache = {}

    def one_pair(i):
        i = 1 + i % 64
        m = pairs_data["matches"][i % pairs_data["matches"].shape[0]]
        lg = pairs_data["logits"][i % pairs_data["logits"].shape[0]]
        if i not in noise_cache:
            noise_cache[i] = synth.gumbel_noise((B, N), seed=1000 + i)
        noise = noise_cache[i]
        t0 = time.perf_counter()
        with torch.no_grad():
            idx, ret, _ = O.gumbel_topk(lg, noise, 1.0, k)
            smp = O.gather_samples(m, ret)
            if solver == "f8":
                models = O.fundamental_8pt(smp)
            elif solver == "stewenius":
                models = O.stewenius_5pt(smp)[0].reshape(-1, 3, 3)
            else:
                E, ok, _ = O.nister_5pt(smp)
                models = O.compact_models(E, ok)
            scores, masks = O.msac_score(m, models, 7.5e-4, chunk=2048)
            b = int(torch.argmax(torch.nan_to_num(scores, nan=-1.0)))
            _ = int(masks[b].sum())
        return time.perf_counter() - t0

    _pin_cpu_threads()
    torch.set_num_threads(1)
    one_pair(0)
    rec = _cpu_rate(one_pair, B, args.cpu_seconds)
    torch.set_num_threads(min(32, cores))
    return {**rec,
            "sample": f"one pair per unit: {N} pts x {B} hyps, torch-CPU f32 oracle (sample+gather+solve+score+argmax); best of "
                      f"{sorted(rec['by_threads'])} threads = {rec['threads']}, median of {rec['reps']} units, {rec['seconds']} s in all",
            "reference_import": {"value": REFERENCE_IMPORT.get(args.workload), "unit": "hypotheses/s", "cores": 8,
                                 "kind": "reference",
                                 "note": "the reference's own Python imported in the build container (BASELINE.md section 2: 8 "
                                         "cores, torch CPU f32, no_grad, same synthetic recipe); it cannot travel to the GPU box, "
                                         "so this figure is quoted, not re-measured here.  The oracle above is a vectorised "
                                         "restatement and therefore faster than the reference's per-sample Python loops"}}


def config_cpu_baselines(budget_s=24.0):
    """The CPU oracle on a bounded sample of every OTHER BASELINE config, on this box's host cores (round-4 review: the figures
    quoted from the build container are not measurements of this box): c1 = configs[0] (uniform 8-point samples -> LSQ F -> MSAC:
    literally the reference's `-d cpu` case), c3 = configs[2] (Stewenius, 4096 hypotheses per pair), c4 = configs[3] (rigid SVD +
    residuals, 50 000 points x 2048 hypotheses), c5_train = configs[4]'s step per pair (sampler -> Nister -> best-of-ten vs the
    ground truth -> MatchLoss, forward + autograd backward to the logits).  Best median over 1 / 8 / 32 threads (_cpu_rate)."""
    from oracle import cpu_ref as O
    from differentiable_ransac_amd import synth
    cores = os.cpu_count() or 1
    threads = min(32, cores)
    _pin_cpu_threads()
    out = {}

    def measure(name, unit_fn, hyps_per_unit, what):
        torch.set_num_threads(1)
        unit_fn(0)
        rec = _cpu_rate(unit_fn, hyps_per_unit, budget_s / 4)
        rec["sample"] = (f"unit = {what}; best of {sorted(rec['by_threads'])} threads = {rec['threads']}, median of {rec['reps']} units, "
                         f"{rec['seconds']} s in all")
        rec["best_value"] = rec["value"]      # (round 5's name for max(32 threads, 1 thread): kept for older readers)
        out[name] = rec

    # c1: 128 correspondences, 64 hypotheses, uniform sampler, 8-point F, MSAC
    p1 = synth.two_view_pair(0, 128, pixel=True)
    gen = torch.Generator().manual_seed(1)

    def c1(i):
        t0 = time.perf_counter()
        with torch.no_grad():
            idx = O.uniform_sample(64, 8, 128, generator=gen)
            F = O.fundamental_8pt(p1["matches"][idx])
            sc, mk = O.msac_score(p1["matches"], F, 0.75)
            _ = int(mk[int(torch.argmax(torch.nan_to_num(sc, nan=-1.0)))].sum())
        return time.perf_counter() - t0
    measure("c1", c1, 64, "128 pts x 64 hyps: uniform sample + 8-point LSQ F + MSAC + argmax")

    # c3: Stewenius, 2000 points x 4096 hypotheses per pair
    p3 = synth.two_view_pair(1, 2000)
    n3 = synth.gumbel_noise((4096, 2000), seed=77)

    def c3(i):
        t0 = time.perf_counter()
        with torch.no_grad():
            idx, ret, _ = O.gumbel_topk(p3["logits"], n3, 1.0, 5)
            models = O.stewenius_5pt(O.gather_samples(p3["matches"], ret))[0].reshape(-1, 3, 3)
            sc, mk = O.msac_score(p3["matches"], models, 7.5e-4, chunk=2048)
            _ = int(mk[int(torch.argmax(torch.nan_to_num(sc, nan=-1.0)))].sum())
        return time.perf_counter() - t0
    measure("c3", c3, 4096, "one pair: 2000 pts x 4096 hyps, Gumbel sampler + Stewenius + MSAC + argmax")

    # c4: rigid SVD, 50 000 points x 2048 hypotheses (noise and masks of a full batch are 0.4 GB each: a quarter of the hypotheses
    # per unit, same per-hypothesis work)
    p4 = synth.rigid_pair(0, 50000)
    n4 = synth.gumbel_noise((512, 50000), seed=78)

    def c4(i):
        t0 = time.perf_counter()
        with torch.no_grad():
            O.ransac3d_train_batch(p4["matches"].float(), p4["logits"], n4, flag=False)
        return time.perf_counter() - t0
    measure("c4", c4, 512, "50 000 pts x 512 of the 2048 hyps: Gumbel sampler + rigid SVD + squared residuals and masks")

    # c5 train step, one pair: forward + backward to the logits
    p5 = synth.two_view_pair(2, 2000)
    n5 = synth.gumbel_noise((1024, 2000), seed=79)

    def c5(i):
        lg = p5["logits"].clone().requires_grad_(True)
        t0 = time.perf_counter()
        chosen, _ = O.ransac_train_batch(p5["matches"], lg, n5, p5["gt_E"], "nister")
        loss = O.match_loss(chosen, p5["matches"], p5["inliers"])
        loss.backward()
        return time.perf_counter() - t0
    measure("c5_train_p32", c5, 1024, "one pair of the train step: 2000 pts x 1024 hyps, sampler + Nister + best-of-ten + MatchLoss, "
                                      "forward + autograd backward to the logits")
    torch.set_num_threads(threads)
    return out


PMC_CAPTURE = "r6_pmc_fetch_write.json"   # the round's rocprofv3 --pmc capture of the headline workload (tools/rocprof_pmc_summary.py)


def pmc_traffic(kernel_key, shape):
    """HBM bytes per launch of the scoring kernel from the committed rocprofv3 PMC passes, valid only for the kernel
    source they were collected on: the JSON carries the sha256 of the source files, compared with the tree's."""
    path = os.path.join(ROOT, "profiles", PMC_CAPTURE)
    if not os.path.exists(path):
        return None, "no PMC capture committed for this round"
    rec = json.load(open(path))
    srcs = rec.get("kernel_sources", {})
    for rel, sha in srcs.items():
        p = os.path.join(ROOT, rel)
        if not os.path.exists(p) or hashlib.sha256(open(p, "rb").read()).hexdigest() != sha:
            return None, f"{rel} changed since profiles/{PMC_CAPTURE} was collected: traffic not attributable"
    if rec.get("workload") != shape:
        return None, f"profiles/{PMC_CAPTURE} was collected on {rec.get('workload')}, this run is {shape}"
    pmc = rec.get("kernels", {}).get(kernel_key, {})
    if "FETCH_SIZE" not in pmc or "WRITE_SIZE" not in pmc:
        return None, f"no counters for {kernel_key} in profiles/{PMC_CAPTURE}"
    # gfx950: FETCH_SIZE shows half the bytes of 16-B/lane streams (MI355X_MICROARCH.md, HBM) -> doubled (upper bound:
    # most of this kernel's reads are scalar-cache model loads); WRITE_SIZE taken as is (matches the mask bytes to 0.2 %)
    return ((2.0 * pmc["FETCH_SIZE"]["avg"] + pmc["WRITE_SIZE"]["avg"]) * 1024.0,
            f"NOT measured in this run: read from the builder-run capture profiles/{PMC_CAPTURE} ((2*FETCH_SIZE + "
            "WRITE_SIZE) KiB per dispatch), accepted only because the sha256 of the kernel sources it was collected on equals the tree's")


# ----------------------------------------------------------------------------------------------------------------------
def self_launch(args):
    """`python bench.py --gpus N` started plain (no RANK in the environment): start the N ranks here, one process per GPU,
    under torch.distributed.run on the loopback address -- the command the driver would have used -- and hand its exit code
    back.  Rank 0 of the child job prints the JSON line on this process's stdout."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
               OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", "8"))
    return subprocess.call(cmd, env=env)


def rendezvous_only(args, world, rank):
    """Launcher check: the ranks form the process group (gloo when there is no GPU) and count themselves."""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    backend = args.backend if torch.cuda.is_available() and not args.gpus_shared else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    dist.init_process_group(backend, rank=rank, world_size=world)
    one = torch.ones(1, device="cuda" if backend == "nccl" else "cpu")
    dist.all_reduce(one)
    dist.barrier()
    if rank == 0:
        print(json.dumps({"metric": "rendezvous only", "n_gpus": world, "n_ranks_seen": int(one.item()), "backend": backend,
                          "requested_gpus": args.gpus}))
    dist.destroy_process_group()


def main():
    args = parse()
    if args.gpus > 1 and "RANK" not in os.environ:
        raise SystemExit(self_launch(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.rendezvous_only:
        return rendezvous_only(args, world, rank)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    if args.gpus_shared:
        local = 0
    elif world > torch.cuda.device_count():
        raise SystemExit(f"bench.py --gpus {world}: this node shows {torch.cuda.device_count()} GPU(s); one rank per GPU "
                         "(--backend gloo --gpus-shared runs the N > 1 code paths on one device, functional check only)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1 or "RANK" in os.environ:   # under torch.distributed.run even a single rank initialises RCCL
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    from differentiable_ransac_amd import ops, sharding
    from differentiable_ransac_amd.ransac import BatchedRANSAC

    w = resolve(args)
    P, N, B = w["pairs"], w["points"], w["hyps"]
    split_h = args.split == "hypotheses" and world > 1
    if split_h:
        if args.mode != "test" or w["solver"] == "rigid":
            raise SystemExit("--split hypotheses is a test-mode option of the two-view workloads")
        if B % world:
            raise SystemExit("--split hypotheses needs hyps divisible by the number of GPUs")
        w = dict(w, hyps=B // world)
    # hypothesis split: every rank holds the SAME pairs (rank 0's data) and its own sampler stream
    if args.logits_fixture and (w["solver"] in ("f8", "rigid") or N != 2000):
        raise SystemExit("--logits-fixture holds 2000-point essential-matrix pairs: use it with the c2 / c3 workloads")
    use_graph = args.graph == "on" or (args.graph == "auto" and args.mode == "train")
    step, info = make_step(w, dev, rank=0 if split_h else rank, mode=args.mode,
                           seed=sharding.hypothesis_seed(1234, rank) if split_h else 1234, fixture=args.logits_fixture,
                           device_seeds=use_graph)
    eager_step = step
    if use_graph:
        # one graph launch per step instead of 6 (test) / ~25 (train, forward + backward) Python-issued launches; the
        # collective of an N > 1 train step and the winner merge of --split hypotheses stay outside the graph
        from differentiable_ransac_amd.graphs import GraphedStep
        step = GraphedStep(step)
    rn, matches, logits, S = info["rn"], info["matches"], info["logits"], info["S"]
    K1, K2 = info["K"]
    M = w["hyps"] * S
    score_prefix = "dr_rigid_residual_" if w["solver"] == "rigid" else "dr_msac_score_f"
    n_seg = max(1, args.segments)
    timer = CallTimer((score_prefix,), args.steps * n_seg)

    # the training step's collective: one flat bucket = the scores network's gradient (random stand-in of the reference
    # CLNet's size: the network itself is out of scope).  Ranks own different pairs: the per-pair logits gradient continues into
    # each rank's own backward through the scores network; what the ranks average is that network's parameter gradient
    # (train.py:150-175), nothing else.  It is issued ASYNCHRONOUSLY (sharding.AsyncGradientBucket): step i's bucket is handed to
    # RCCL right after step i's backward has been enqueued, and the compute stream waits for it only after step i + 1's forward
    # + backward have been enqueued (where the optimizer would read it) -- the collective runs under the next step's kernels.
    bucket = None
    base_step = step
    if args.mode == "train" and world > 1:
        bucket = sharding.AsyncGradientBucket(CLNET_PARAMS, dev, dist, fill=torch.randn(CLNET_PARAMS, device=dev))
        step = sharding.OverlappedStep(base_step, bucket)
    elif split_h:
        def step():
            out = base_step()
            sc, md, win, inl = sharding.merge_best(out["score"], out["model"], (out["inliers"],), dist)
            return dict(out, score=sc, model=md, inliers=inl, winner=win)

    streams = [torch.cuda.Stream(device=dev) for _ in range(max(1, args.streams))]
    outs = [None] * len(streams)

    def issue(i):
        if len(streams) == 1:
            # one batch in flight: the caller's current stream (a stream context costs the train step's autograd engine
            # 0.2 ms of host time per step for cross-thread stream bookkeeping: scratch/host_overhead2.py)
            outs[0] = step()
            return
        st = streams[i % len(streams)]
        with torch.cuda.stream(st):
            outs[i % len(streams)] = step()

    # bounded run-ahead: the host issues a step in ~0.1 ms and would otherwise queue hundreds of steps; the first time the
    # runtime's queue fills it stalls the device for ~40 ms, once (scratch/stall_find.py, stall_find2.py).  Step i is issued
    # when step i - kWindow has finished -- the device always has kWindow steps queued, so it never waits for the host.
    # Round 6: an event record is a marker packet with a barrier -- 5.6 us of idle device per record in the step's timeline
    # (profiles/r6_headline_timeline.md: one after every step for this window, two around the scoring launch = 17 us of a 0.905 ms
    # step were the measurement).  One stream: the window's event after every kStride-th step, the scoring launch's pair in every
    # K4_EVENT_EVERY-th step; several streams keep one record per step (each stream needs its own).
    kWindow = 32
    kStride = 4 if len(streams) == 1 else 1
    done_ev = [torch.cuda.Event() for _ in range(kWindow // kStride)]

    def issue_bounded(i, n_issued):
        if n_issued >= kWindow and n_issued % kStride == 0:
            done_ev[((n_issued - kWindow) // kStride) % len(done_ev)].synchronize()   # recorded after step n_issued - kWindow + kStride - 1
        issue(i)
        if (n_issued + 1) % kStride == 0:
            done_ev[(n_issued // kStride) % len(done_ev)].record(streams[i % len(streams)] if len(streams) > 1 else torch.cuda.current_stream())

    # time-based pre-conditioning (not counted in `warmup`): the same step until --prewarm-s of wall time has passed.  The
    # driver's `--steps 20 --warmup 5` is 6 ms + 24 ms of device time: without this the region is the clock ramp of a cold box
    # (round 2: 0.668 ms per scoring launch on the driver's run against 0.612 ms in the 1000-step runs).
    n_issued, t_pre = 0, time.perf_counter()
    while True:
        for _ in range(8):
            issue_bounded(n_issued, n_issued)
            n_issued += 1
        done_ev[((n_issued - 1) // kStride) % len(done_ev)].synchronize()   # (8 steps per chunk: its last step carries a record)
        more = time.perf_counter() - t_pre < args.prewarm_s
        if dist is not None:
            # every rank must issue the SAME number of steps (a train step carries a collective): the ranks agree per chunk
            flag = torch.tensor([1.0 if more else 0.0], device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            more = bool(flag.item() > 0)
        if not more:
            break
    torch.cuda.synchronize()
    prewarm_s, prewarm_steps = time.perf_counter() - t_pre, n_issued
    for i in range(args.warmup):
        issue_bounded(n_issued, n_issued)
        n_issued += 1
    torch.cuda.synchronize()
    # the timed region: n_seg segments of EXACTLY --steps steps, each bracketed by barrier + synchronize on both sides; whole-job
    # rate of a segment = sum of hypotheses over ranks / max elapsed over ranks; the line reports the MEDIAN segment
    seg_elapsed, seg_rate, seg_k4, seg_coll = [], [], [], []
    for sgi in range(n_seg):
        if bucket is not None:
            bucket.drain()
            bucket.exposed_events.clear()
            bucket.trace.clear()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            timer.i = sgi * args.steps + i if i % K4_EVENT_EVERY == 0 else -1
            issue_bounded(i, n_issued)
            n_issued += 1
        if bucket is not None:
            bucket.drain()               # the last step's collective belongs to the timed region
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        timer.i = -1
        rate, el = sharding.job_throughput(P * w["hyps"] * args.steps, el, dist, dev)
        seg_elapsed.append(el)
        seg_rate.append(rate)
        if bucket is not None and bucket.exposed_events:
            seg_coll.append(sum(a.elapsed_time(b) for a, b in bucket.exposed_events) / len(bucket.exposed_events))
    out = outs[(args.steps - 1) % len(streams)]
    med = sorted(range(n_seg), key=lambda i: seg_elapsed[i])[n_seg // 2]
    elapsed, job_hyps_per_s = seg_elapsed[med], seg_rate[med]
    n_ranks_seen = 1
    if dist is not None:
        one = torch.ones(1, device=dev)
        dist.all_reduce(one)
        n_ranks_seen = int(one.item())
    # exposed = what the compute stream waited at the optimizer's read (median segment); collective_ms = the same all-reduce
    # issued BLOCKING on an idle device after the timed region (its own duration, nothing to hide behind)
    collective_exposed_ms = sorted(seg_coll)[len(seg_coll) // 2] if seg_coll else (0.0 if world > 1 else None)
    collective_ms = 0.0 if world > 1 else None
    if bucket is not None:
        torch.cuda.synchronize()
        evs = []
        for _ in range(20):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            sharding.allreduce_mean_([bucket.buf[0]], dist)
            e1.record()
            evs.append((e0, e1))
        torch.cuda.synchronize()
        collective_ms = sorted(a.elapsed_time(b) for a, b in evs)[len(evs) // 2]

    common = {"unit": "hypotheses/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
              "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
              "scaling": "strong" if split_h else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
              "n_ranks_seen": n_ranks_seen, "collective_ms": collective_ms, "collective_exposed_ms": collective_exposed_ms,
              "prewarm_s": round(prewarm_s, 3), "prewarm_steps": prewarm_steps,
              "segments": {"n": n_seg, "steps_each": args.steps, "reported": "median segment",
                           "ms_per_step": [round(e / args.steps * 1e3, 5) for e in seg_elapsed],
                           "hypotheses_per_s": [round(r, 1) for r in seg_rate]}}

    if args.mode == "train":
        # multi-rank readiness (round 6): which pairs every rank owned -- rank r builds the synthetic pairs rank * P .. rank * P + P - 1,
        # the block sharding.pair_range(P * world, r, world) -- gathered and checked to cover the job's pairs exactly once; and the
        # program order of the overlapped collective on this rank's call trace (step i + 1 enqueued before the wait on bucket i)
        partition, trace_ok = None, None
        if world > 1:
            mine = torch.tensor([rank * P, rank * P + P], dtype=torch.int64, device=dev if args.backend == "nccl" else "cpu")
            got = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(got, mine)
            ranges = [[int(g[0]), int(g[1])] for g in got]
            seen = sorted(q for lo, hi in ranges for q in range(lo, hi))
            partition = {"total_pairs": P * world, "ranges": ranges,
                         "equals_pair_range": ranges == [list(sharding.pair_range(P * world, r, world)) for r in range(world)],
                         "covers_once": seen == list(range(P * world))}
            tr = bucket.trace                      # (the last timed segment's: step / bucket numbers run on from the earlier ones)
            pos = {ev: j for j, ev in enumerate(tr)}
            checked = [i for tag, i in tr if tag == "wait" and ("enqueued", i + 1) in pos and ("launch", i + 1) in pos]
            trace_ok = bool(checked) and all(pos[("enqueued", i + 1)] < pos[("wait", i)] < pos[("launch", i + 1)] for i in checked)
        if rank == 0:
            print(json.dumps({"metric": "hypotheses/sec, train step (forward + backward to the logits"
                                        + (", gradient all-reduce)" if world > 1 else ")"),
                              "value": job_hyps_per_s, **common,
                              "config": {"workload": f"{w['solver']} train step (sample, solve, best-of-10 vs GT, MatchLoss, "
                                                     f"backward), {N} pts x {B} hyps per pair, {P} pairs/GPU",
                                         "mode": "train",
                                         "issue": ("HIP graph replay of forward + loss + backward (one launch per step)"
                                                   if use_graph else "eager: one Python call per launch, autograd backward"),
                                         "parallelism": f"pairs sharded over {world} GPU(s); one flat RCCL all-reduce of "
                                                        f"{CLNET_PARAMS} f32 (the scores network's gradient) per step" if world > 1 else "single GPU"},
                              "collective_issue": ("asynchronous: bucket i is all-reduced under the kernels of step i + 1, waited for "
                                                   "where the optimizer reads it (sharding.AsyncGradientBucket)" if world > 1 else None),
                              "collective_exposed_share_of_step": (collective_exposed_ms / (elapsed / args.steps * 1e3))
                              if collective_exposed_ms else None,
                              "collective_bytes": 4 * CLNET_PARAMS if world > 1 else 0,
                              "pair_partition": partition, "overlap_trace_ok": trace_ok,
                              "grad_finite": bool(torch.isfinite(out["grad"]).all())}))
        timer.close()
        if dist is not None:
            dist.destroy_process_group()
        return

    # informational second region: the same K steps with two batches in flight on two streams
    overlap = None
    if world == 1 and args.extras:
        n2 = 1 if len(streams) > 1 else 2
        s2 = [torch.cuda.Stream(device=dev) for _ in range(n2)]
        keep = [None] * n2
        for i in range(2 * n2):                  # warm the per-stream allocator pools
            with torch.cuda.stream(s2[i % n2]):
                keep[i % n2] = step()
        torch.cuda.synchronize()
        def issue2(i):
            with torch.cuda.stream(s2[i % n2]):
                keep[i % n2] = step()
            return s2[i % n2]
        n_ov = max(args.steps, 100)
        t1 = time.perf_counter()
        run_bounded(issue2, n_ov)
        torch.cuda.synchronize()
        e2 = time.perf_counter() - t1
        overlap = {"streams": n2, "steps": n_ov, "value": P * B * n_ov / e2, "ms_per_step": e2 / n_ov * 1e3}
        del keep

    with_refit, topdown = None, None
    if args.extras and world == 1 and w["solver"] != "rigid":
        n_x = 200
        rn_refit = BatchedRANSAC(w["solver"], ransac_batch_size=B, train=False, threshold=0.75, max_iterations=B,
                                 seed=4321, keep_masks=True, refit=True)
        for _ in range(3):
            rn_refit(matches, logits, K1, K2)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        run_bounded(lambda i: rn_refit(matches, logits, K1, K2) and None, n_x)
        torch.cuda.synchronize()
        e3 = time.perf_counter() - t2
        with_refit = {"value": P * B * n_x / e3, "ms_per_step": e3 / n_x * 1e3}
        if w["sampler"] == "gumbel":
            rn_td = BatchedRANSAC(w["solver"], ransac_batch_size=B, train=False, threshold=0.75, max_iterations=B,
                                  seed=77, keep_masks=True, refit=False, sampling="topdown")
            for _ in range(3):
                rn_td(matches, logits, K1, K2)
            torch.cuda.synchronize()
            t3 = time.perf_counter()
            run_bounded(lambda i: rn_td(matches, logits, K1, K2) and None, n_x)
            torch.cuda.synchronize()
            e4 = time.perf_counter() - t3
            topdown = {"value": P * B * n_x / e4, "ms_per_step": e4 / n_x * 1e3, "streams": 1,
                       "note": "informational: the index sets drawn top-down (k draws without replacement from softmax(logits) = "
                               "the law of the Gumbel top-k, O(k log N) per hypothesis instead of noise for every point); test "
                               "mode only -- the headline keeps the reference's sampler"}

    if use_graph:             # a replayed graph has no Python-level launches to put events around: time the scoring launch
        n_ev = min(20, args.steps)   # in a few eager steps of the same driver instead
        for i in range(n_ev):
            timer.i = i
            eager_step()
        torch.cuda.synchronize()
        timer.i = -1
        k4_ms = timer.mean_ms(n_ev)
        seg_k4 = [k4_ms]
    else:
        # HIP-event mean of the scoring launch per segment; the median of the segment means is the roofline's duration
        seg_k4 = [timer.mean_ms(args.steps, sgi * args.steps) for sgi in range(n_seg)]
        k4_ms = sorted(seg_k4)[n_seg // 2]
    iso_ms = k4_ms
    if len(streams) > 1:      # with two batches in flight the kernel shares the CUs: also measure it alone
        n_iso = min(10, args.steps)
        for i in range(n_iso):
            timer.i = i
            step()
        torch.cuda.synchronize()
        timer.i = -1
        iso_ms = timer.mean_ms(n_iso)
    timer.close()
    # what an event pair reads with NOTHING between its two records (the second marker packet's own processing): the part of
    # `avg_launch_ms` that is not the kernel.  `frac` stays the raw figure; `frac_net_of_event_pair` is the one to hold against
    # rocprofv3's duration of the same kernel (profiles/r6_kernel_stats_c2.md)
    torch.cuda.synchronize()
    empty = []
    for _ in range(40):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        e1.record()
        empty.append((e0, e1))
    torch.cuda.synchronize()
    empty_pair_ms = sorted(a.elapsed_time(b) for a, b in empty)[len(empty) // 2]
    rigid = w["solver"] == "rigid"
    bytes_per_launch = k4r_bytes(P, N, M) if rigid else k4_bytes(P, N, M)
    valid_frac = None
    flops_per_launch = 28.0 * P * M * N
    if not rigid:
        with torch.no_grad():
            _, v_, _ = rn.hypotheses(matches, logits)
        valid_frac = float(v_.float().mean())
        flops_per_launch = 39.0 * P * M * N * valid_frac   # only the slots the solver marked valid are evaluated
    achieved = bytes_per_launch / (k4_ms * 1e-3) / 1e9
    kernel_name = "rigid_residual_kernel" if rigid else ("msac_score_kernel_f32_fast16" if N % 16 == 0 else "msac_score_kernel_f32_fast")
    traffic, traffic_note = (None, "PMC capture exists for the c2 workload only")
    if args.workload == "c2" and not split_h:
        traffic, traffic_note = pmc_traffic("dr::" + kernel_name, {"pairs": P, "points": N, "hyps": B})

    inl_frac = None if rigid else float(out["inliers"].float().mean()) / N

    result = {
        "metric": "hypotheses/sec (and image-pairs/sec) at 2000 pts x 1024 hyps, 1/2/4/8 GPU",
        "value": job_hyps_per_s,
        **common,
        "config": {"workload": f"{w['text']}, test mode, {P} pairs/GPU/step", "baseline_config_index": w["baseline_config"],
                   "sampler": w["sampler"], "pairs_per_gpu": P, "points": N, "hypotheses_per_pair": B,
                   "hypotheses_per_pair_per_gpu": w["hyps"], "models_per_pair": B * S, "solver": w["solver"],
                   "parallelism": (f"hypotheses of the same {P} pair(s) split over {world} GPUs, per-pair winners merged by two "
                                   "all_gathers per step" if split_h else f"pairs sharded over {world} GPU(s), no collective"),
                   "streams": len(streams),
                   "issue": f"{len(streams)} batch(es) in flight, round-robin over {len(streams)} HIP stream(s), "
                            + ("HIP graph replay" if use_graph else "eager launches (HIP events around the scoring launch)")},
        "pairs_per_s": (1 if split_h else world) * P * args.steps / elapsed,
        "roofline": {"bound": "hbm", "binding_unit": None if rigid else "valu_f32",
                     "bound_note": ("`bound` names the roof `frac` is priced against (BASELINE's target is stated against HBM: the mask "
                                    "stream is the launch's only large transfer); the unit that saturates first is the f32 vector ALU "
                                    "(`binding_unit`, `what_bounds_it`, `valu_frac_of_157.3`): 38 flop/B against a machine balance of "
                                    "19.7.  `k4_all_valid` is the same launch with every slot evaluated"),
                     "kernel": kernel_name, "valid_slot_fraction": valid_frac, "achieved": achieved,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "traffic_source": traffic_note,
                     "avg_launch_ms": k4_ms, "segments_avg_launch_ms": [round(x, 5) for x in seg_k4],
                     "empty_event_pair_ms": empty_pair_ms,
                     "frac_net_of_event_pair": bytes_per_launch / (max(k4_ms - empty_pair_ms, 1e-6) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "event_sampling": (f"HIP-event pair around the scoring launch of every {K4_EVENT_EVERY}th step of the timed region "
                                        f"({len(timer.used)} launches timed), on the stream the kernel is launched on: a pair is two "
                                        "marker packets with barriers, ~11 us of idle device in the step that carries it"),
                     "algorithmic_bytes_per_launch": bytes_per_launch,
                     "isolated": {"avg_launch_ms": iso_ms, "achieved": bytes_per_launch / (iso_ms * 1e-3) / 1e9,
                                  "frac": bytes_per_launch / (iso_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                  "note": "same launch, one stream, nothing else resident"},
                     "frac_of_measured_copy_peak_6290": achieved / 6290.0,
                     "what_bounds_it": ("the f32 vector ALU under the chip's power budget: SQ_ACTIVE_INST_VALU = 0.84-0.88 of the "
                                        "kernel's cycles at 1.87-2.15 GHz effective (2.4 nominal), 211 vector instructions per (model, 16 "
                                        "points per lane) of which 152 are the residual's packed FMAs / multiplies; the write path is not "
                                        "backed up.  A variant with the points streamed from LDS at 6-8 waves per SIMD issues 0.89 of the "
                                        "cycles, needs 9 % fewer of them and is 12 % SLOWER in the step: the chip clocks 7 % lower under it "
                                        "(profiles/r4_k4_lds_variants.md; counters + ISA budget: profiles/r3_k4_counters_p128.md)"
                                        if (not rigid and P >= 64) else None),
                     "valu_tflops": flops_per_launch / (iso_ms * 1e-3) / 1e12,
                     "valu_frac_of_157.3": flops_per_launch / (iso_ms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS},
        "check": {"mean_inlier_fraction_of_best_model": inl_frac},
        ("two_batches_in_flight" if len(streams) == 1 else "one_stream"): overlap,
        "with_final_refit": with_refit,
        "sampler_topdown": topdown,
    }
    if args.profile_kernels and rank == 0 and not rigid:
        result["kernel_breakdown_ms"] = kernel_breakdown(w, rn, matches, logits, ops)
    if rank == 0 and world == 1 and not args.no_configs:
        n_cfg = {"c1": 300, "c2": 200, "c3": 60, "c4": 150}
        g_ = args.graph != "off"
        result["configs"] = {k: config_record(k, dev, n_cfg[k], 5, graph=g_) for k in sorted(WORKLOADS) if k != args.workload}
        if args.workload == "c2" and P != 32:
            result["configs"]["c2_p32"] = config_record("c2", dev, 300, 5, pairs=32, graph=g_)   # round 1's batch size
        if args.workload == "c2":
            # SURVEY's literal C2: ONE pair per call (test.py:38, model_cl.py:488-490), eager and replayed as one graph
            result["configs"]["c2_p1"] = config_record("c2", dev, 600, 5, pairs=1, graph=True)
            result["configs"]["c5_train_p32"] = train_record(dev)
            result["configs"]["dropin_layer_loop"] = dropin_layer_loop_record(dev)
            result["fused_driver"] = fused_driver_record(dev)
            result["k4_all_valid"] = k4_all_valid_record(dev)
        if not args.logits_fixture and os.path.exists(os.path.join(ROOT, "tests", "golden", "clnet_logits.npz")):
            # the c2 workload on reader-produced pairs with the reference network's scores instead of synthetic logits
            fstep, finfo = make_step(dict(WORKLOADS["c2"], pairs=32), dev, fixture=True)
            for _ in range(5):
                fout = fstep()
            torch.cuda.synchronize()
            tf = time.perf_counter()
            run_bounded(lambda i: fstep() and None, 199)
            fout = fstep()
            torch.cuda.synchronize()
            ef = time.perf_counter() - tf
            gi = finfo["data"]["inliers"].to(dev)
            result["clnet_logits"] = {
                "workload": "c2 on tests/golden/clnet_logits.npz (4 reader-produced pairs tiled to 32; sampler input = the "
                            "reference CLNet's log-probabilities, shipped weights saved_model_5PC_l_epi)",
                "steps": 200, "ms_per_step": ef / 200 * 1e3, "hypotheses_per_s": 32 * 1024 * 200 / ef,
                "mean_inlier_fraction_of_best_model": float(fout["inliers"].float().mean()) / 2000,
                "geometric_inlier_fraction_of_the_data": float(gi.float().mean()),
                "best_mask_agreement_with_geometric_inliers": float((fout["mask"] == gi).float().mean())}
    if rank == 0 and world == 1 and not args.no_cpu_baseline and info["data"] is not None:
        result["cpu_baseline"] = cpu_baseline(args, w, info["data"])
        if "configs" in result:
            # a measured CPU figure next to every config record (the reference-import numbers stay beside them, quoted)
            for key, rec in config_cpu_baselines(2.0 * args.cpu_seconds).items():
                if key in result["configs"]:
                    result["configs"][key]["cpu_baseline"] = rec
    if rank == 0:
        print(json.dumps(result))
    if dist is not None:
        dist.destroy_process_group()


def kernel_breakdown(w, rn, matches, logits, ops):
    """Per-stage device time by HIP events (each stage synchronised): sampler / gather / solver / scoring / select."""
    P, N, B = w["pairs"], w["points"], w["hyps"]
    k = rn.k
    out = {}

    def t(fn, reps=10):
        fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            r = fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / reps, r

    out["K1_gumbel_topk"], r = t(lambda: ops.gumbel_topk(logits, B, k, 1.0, None, seed=1))
    out["K2_gather"], smp = t(lambda: ops.gather(matches, r["idx"], r["y_sel"]))
    if w["solver"] == "f8":
        out["K3_solver"], (F, v) = t(lambda: ops.solve_f8(smp))
        models, valid = F.unsqueeze(2), v.unsqueeze(2)
    elif w["solver"] == "stewenius":
        out["K3_solver"], (models, valid) = t(lambda: ops.solve_stewenius5(smp))
    else:
        out["K3_solver"], (models, valid) = t(lambda: ops.solve_nister5(smp))
    flat = models.reshape(P, -1, 3, 3)
    thr = torch.full((P,), 7.5e-4, device=matches.device)
    vflat = valid.reshape(P, -1)
    out["K4_msac_masks"], (sc, mk) = t(lambda: ops.msac_score(matches, flat, thr, True, vflat))
    out["K4_msac_nomask"], _ = t(lambda: ops.msac_score(matches, flat, thr, False, vflat))
    out["K4_msac_masks_all_slots"], _ = t(lambda: ops.msac_score(matches, flat, thr, True))
    out["valid_fraction"] = float(valid.float().mean())
    out["K6_select_best"], _ = t(lambda: ops.select_best(matches, flat, sc, thr, valid.reshape(P, -1)))
    return out


if __name__ == "__main__":
    main()
