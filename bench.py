#!/usr/bin/env python
"""Headline benchmark: hypotheses/s of the differentiable-RANSAC hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One *step* = one pass of the hot path (test mode, ransac.py:55-144, one RANSAC batch) over one batch of
synthetic image pairs resident in HBM: K1 Gumbel top-k sampling (in-kernel Philox) -> K2 gather ->
K3 Nister 5-point -> K4 MSAC scoring of all 10*B models against all N points (masks materialised, as
MSACScore.score's contract requires) -> K6 per-pair arg-max / best mask / inlier count.
Workload = BASELINE.json configs[1]: Nister 5-pt, N = 2000 points, B = 1024 hypotheses per pair, Gumbel
sampler, MSAC; `--pairs` pairs per GPU per step (weak scaling: pairs shard across ranks, no data-path
collective -- SURVEY 8(e)).  Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
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


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--pairs", type=int, default=32, help="image pairs per GPU per step")
    ap.add_argument("--points", type=int, default=2000)
    ap.add_argument("--hyps", type=int, default=1024)
    ap.add_argument("--solver", default="nister", choices=["nister", "stewenius", "f8"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the bounded CPU-baseline sample")
    ap.add_argument("--profile-kernels", action="store_true", help="per-kernel HIP-event breakdown (extra syncs)")
    ap.add_argument("--mode", default="test", choices=["test", "train"],
                    help="test: the headline inference path; train: sampler -> solver -> best-of-10 vs GT -> loss, forward "
                         "+ backward to the logits (ransac.py:78-108 + train.py:150), reported with the same JSON shape")
    ap.add_argument("--extras", action="store_true",
                    help="after the official timed region also measure (a) two-stream overlap of consecutive batches and "
                         "(b) the step followed by the final refit; off by default so that a profiler sees only the official loop")
    ap.add_argument("--sampler", default="gumbel", choices=["gumbel", "topdown"],
                    help="gumbel: the reference's sampler (noise for every point of every hypothesis + top-k, in-kernel "
                         "Philox); topdown: the same index-set distribution drawn as k soft-max draws without replacement "
                         "(test mode only) -- reported as a variant, never the default")
    ap.add_argument("--streams", type=int, default=1,
                    help="HIP streams the K timed steps are issued on round-robin.  Default 1: strictly one kernel at a "
                         "time, so that the HIP-event duration of the scoring kernel in the timed region is its own "
                         "(roofline attribution).  2 = two batches in flight (the latency-bound sampler/solver of batch "
                         "i+1 runs under the VALU-bound scoring of batch i): +14 % throughput, always measured after the "
                         "timed region and reported as `two_batches_in_flight`")
    return ap.parse_args()


def cpu_baseline(args, pairs_data):
    """The CPU oracle (oracle/cpu_ref.py, a vectorised torch restatement of the reference path) timed on the host
    cores, on a bounded sample of the same workload: whole pairs (N points x B hypotheses), one after the other like
    the reference's per-pair loop (model_cl.py:488), until the time budget is spent.  The thread count is calibrated
    first (torch's small batched LAPACK calls collapse when oversubscribed: 256 threads are 250x slower than 16 here)."""
    from oracle import cpu_ref as O
    from differentiable_ransac_amd import synth
    cores = os.cpu_count() or 1
    k = 8 if args.solver == "f8" else 5
    noise_cache = {}

    def one_pair(i):
        m = pairs_data["matches"][i % pairs_data["matches"].shape[0]]
        lg = pairs_data["logits"][i % pairs_data["logits"].shape[0]]
        if i not in noise_cache:
            noise_cache[i] = synth.gumbel_noise((args.hyps, args.points), seed=1000 + i)
        noise = noise_cache[i]
        t0 = time.perf_counter()
        with torch.no_grad():
            idx, ret, _ = O.gumbel_topk(lg, noise, 1.0, k)
            smp = O.gather_samples(m, ret)
            if args.solver == "f8":
                models = O.fundamental_8pt(smp)
            elif args.solver == "stewenius":
                models = O.stewenius_5pt(smp)[0].reshape(-1, 3, 3)
            else:
                E, ok, _ = O.nister_5pt(smp)
                models = O.compact_models(E, ok)
            scores, masks = O.msac_score(m, models, 7.5e-4, chunk=2048)
            b = int(torch.argmax(torch.nan_to_num(scores, nan=-1.0)))
            _ = int(masks[b].sum())
        return time.perf_counter() - t0

    best_t, best_n = None, 1
    for n in sorted({1, min(8, cores), min(16, cores), min(32, cores)}):
        torch.set_num_threads(n)
        one_pair(0)                       # warm-up at this thread count
        t = min(one_pair(0), one_pair(0))
        if best_t is None or t < best_t:
            best_t, best_n = t, n
    torch.set_num_threads(best_n)
    done, t_used = 0, 0.0
    while t_used < args.cpu_seconds and done < 4096:
        t_used += one_pair(1 + done % 64)
        done += 1
    return {"value": done * args.hyps / max(t_used, 1e-9), "unit": "hypotheses/s", "cores": best_n, "kind": "port",
            "host_cores": cores,
            "sample": f"{done} pair(s) x {args.points} pts x {args.hyps} hyps, torch-CPU f32 oracle "
                      f"(sample+gather+solve+score+argmax), {t_used:.1f} s, threads calibrated over 1/8/16/32"}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1 or "RANK" in os.environ:   # under torch.distributed.run even a single rank initialises RCCL
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from differentiable_ransac_amd import ops, synth
    from differentiable_ransac_amd.ransac import BatchedRANSAC

    P, N, B = args.pairs, args.points, args.hyps
    S = 1 if args.solver == "f8" else 10
    M = B * S
    data = synth.batch_two_view(P, N, seed0=rank * P, pixel=(args.solver == "f8"))
    matches = data["matches"].to(dev)
    logits = data["logits"].to(dev)
    K1, K2 = data["K1"].to(dev), data["K2"].to(dev)
    thr_px = 0.75
    rn = BatchedRANSAC(args.solver, ransac_batch_size=B, train=False, threshold=thr_px, max_iterations=B,
                       seed=1234 + rank, keep_masks=True, refit=False, sampling=args.sampler)

    # HIP events bracket exactly the dr_msac_score launch (the ctypes call), on the stream it is launched on
    from differentiable_ransac_amd import _lib as L
    ev = [[torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)] for _ in range(args.steps)]
    state = {"i": -1}
    orig_call = L.call

    def timed_call(name, *a):
        i = state["i"]
        if 0 <= i < args.steps and name.startswith("dr_msac_score_f"):
            ev[i][0].record()
            orig_call(name, *a)
            ev[i][1].record()
        else:
            orig_call(name, *a)

    L.call = timed_call

    if args.mode == "train":
        gt = data["gt_E"].to(dev) if args.solver != "f8" else data["gt_F"].to(dev)
        tr = BatchedRANSAC(args.solver, ransac_batch_size=B, train=True, max_iterations=B, seed=99 + rank)
        lg = logits.clone().requires_grad_(True)

        from differentiable_ransac_amd.loss import MatchLoss
        match_loss = MatchLoss()                     # the reference's default training loss (-w2 1, train.py:70-79)
        gt_mask = data["inliers"].to(dev)

        def step():
            lg.grad = None
            chosen, keep = tr(matches, lg, gt_model=gt)
            if args.solver == "f8":                  # pixel coordinates: plain distance to the ground-truth F
                d = torch.minimum(((chosen - gt[:, None]) ** 2).sum((-1, -2)), ((chosen + gt[:, None]) ** 2).sum((-1, -2)))
                loss = (d * keep).sum()
            else:
                loss = match_loss(chosen, matches, gt_mask, keep)
            loss.backward()
            return {"inliers": torch.zeros(P, device=dev), "grad": lg.grad}
    else:
        def step():
            return rn(matches, logits, K1, K2)

    streams = [torch.cuda.Stream(device=dev) for _ in range(max(1, args.streams))]
    outs = [None] * len(streams)

    def issue(i):
        if args.streams <= 0:      # torch's default stream
            outs[0] = step()
            return
        st = streams[i % len(streams)]
        with torch.cuda.stream(st):
            outs[i % len(streams)] = step()

    for w in range(args.warmup):
        issue(w)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        state["i"] = i
        issue(i)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    out = outs[(args.steps - 1) % len(streams)]
    elapsed = time.perf_counter() - t0
    state["i"] = -1
    # whole-job rate = sum of hypotheses over ranks / max elapsed over ranks (no data-path collective: SURVEY 8(e))
    from differentiable_ransac_amd import sharding
    job_hyps_per_s, elapsed = sharding.job_throughput(P * B * args.steps, elapsed, dist, dev)

    # informational second region: the same K steps with the other issue policy -- two batches in flight on two streams
    # when the official region ran strictly serial, and vice versa
    overlap = None
    if world == 1 and args.mode == "test":
        n2 = 1 if len(streams) > 1 else 2
        s2 = [torch.cuda.Stream(device=dev) for _ in range(n2)]
        keep = [None] * n2
        for i in range(2 * n2):                  # warm the per-stream allocator pools
            with torch.cuda.stream(s2[i % n2]):
                keep[i % n2] = step()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(args.steps):
            with torch.cuda.stream(s2[i % n2]):
                keep[i % n2] = step()
        torch.cuda.synchronize()
        e2 = time.perf_counter() - t1
        overlap = {"streams": n2, "value": P * B * args.steps / e2, "ms_per_step": e2 / args.steps * 1e3}
        del keep

    if args.mode == "train":
        if rank == 0:
            print(json.dumps({"metric": "hypotheses/sec, train step (forward + backward to the logits)",
                              "value": job_hyps_per_s, "unit": "hypotheses/s", "n_gpus": world, "steps": args.steps,
                              "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
                              "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                              "config": {"workload": f"{args.solver} train step (sample, solve, best-of-10 vs GT, MatchLoss, backward), "
                                                     f"{N} pts x {B} hyps per pair, {P} pairs/GPU",
                                         "mode": "train"},
                              "grad_finite": bool(torch.isfinite(out["grad"]).all())}))
        if dist is not None:
            dist.destroy_process_group()
        return
    # informational: the same step followed by the final refit of ransac.py:148-195 (K7: Nister on all points in f64 on a
    # side stream, re-score, keep if better) -- a per-pair epilogue, not part of the hypothesis loop the metric counts
    with_refit = None
    if args.extras and world == 1:
        rn_refit = BatchedRANSAC(args.solver, ransac_batch_size=B, train=False, threshold=thr_px, max_iterations=B,
                                 seed=4321, keep_masks=True, refit=True)
        for _ in range(3):
            rn_refit(matches, logits, K1, K2)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        for _ in range(args.steps):
            rn_refit(matches, logits, K1, K2)
        torch.cuda.synchronize()
        e3 = time.perf_counter() - t2
        with_refit = {"value": P * B * args.steps / e3, "ms_per_step": e3 / args.steps * 1e3}

    # informational (--extras): the same K steps with the top-down draw of the index sets instead of the dense Gumbel
    # sampler (identical set distribution, tests/test_gpu_sampler.py; not the reference's sampler kernel, so not the headline)
    topdown = None
    if args.extras and world == 1 and args.sampler == "gumbel":
        rn_td = BatchedRANSAC(args.solver, ransac_batch_size=B, train=False, threshold=thr_px, max_iterations=B,
                              seed=77, keep_masks=True, refit=False, sampling="topdown")
        keep = [None] * len(streams)
        for i in range(2 * len(streams)):
            with torch.cuda.stream(streams[i % len(streams)]):
                keep[i % len(streams)] = rn_td(matches, logits, K1, K2)
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        for i in range(args.steps):
            with torch.cuda.stream(streams[i % len(streams)]):
                keep[i % len(streams)] = rn_td(matches, logits, K1, K2)
        torch.cuda.synchronize()
        e4 = time.perf_counter() - t3
        topdown = {"value": P * B * args.steps / e4, "ms_per_step": e4 / args.steps * 1e3, "streams": len(streams)}
        del keep

    k4_ms = sum(a.elapsed_time(b) for a, b in ev) / args.steps
    # the same launch with nothing else on the GPU (one stream, a few extra untimed steps): with two batches in flight
    # the scoring kernel shares the CUs with the next batch's sampler/solver, so its wall duration above is longer than
    # its own cost; both are reported
    iso_ms = k4_ms
    if len(streams) > 1:
        n_iso = min(10, args.steps)
        for i in range(n_iso):
            state["i"] = i
            step()
        torch.cuda.synchronize()
        state["i"] = -1
        iso_ms = sum(ev[i][0].elapsed_time(ev[i][1]) for i in range(n_iso)) / n_iso
    bytes_per_launch = P * (16 * N + 36 * M + 4 * M + M * N)          # SURVEY 8(d), masks included (all M rows are written)
    # executed flops: only the slots the solver marked valid are evaluated
    with torch.no_grad():
        _, v_, _ = rn.hypotheses(matches, logits)
    valid_frac = float(v_.float().mean())
    flops_per_launch = 39.0 * P * M * N * valid_frac
    achieved = bytes_per_launch / (k4_ms * 1e-3) / 1e9

    # HBM traffic of the same kernel from the committed rocprofv3 PMC passes (profiles/, collected with
    # `rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python bench.py --steps 5`; KiB per dispatch)
    traffic, traffic_note = None, None
    pmc_path = os.path.join(ROOT, "profiles", "r1_pmc_fetch_write.json")
    if os.path.exists(pmc_path) and (P, N, B, args.solver) == (32, 2000, 1024, "nister"):
        pmc_all = json.load(open(pmc_path))
        pmc = pmc_all.get("dr::msac_score_kernel_f32_fast16", pmc_all.get("dr::msac_score_kernel_f32_fast", {}))
        if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
            # gfx950: FETCH_SIZE shows half the bytes of 16-B/lane streams (MI355X_MICROARCH.md, HBM) -> doubled (upper bound:
            # most of this kernel's reads are scalar-cache model loads); WRITE_SIZE taken as is (matches the mask bytes to 0.2 %)
            traffic = (2.0 * pmc["FETCH_SIZE"]["avg"] + pmc["WRITE_SIZE"]["avg"]) * 1024.0
            traffic_note = "profiles/r1_pmc_fetch_write.json: (2*FETCH_SIZE + WRITE_SIZE) KiB per dispatch"

    # sanity of the result (cheap, outside the timed region): the synthetic pairs have 50 % inliers
    inl_frac = float(out["inliers"].float().mean()) / N

    result = {
        "metric": "hypotheses/sec (and image-pairs/sec) at 2000 pts x 1024 hyps, 1/2/4/8 GPU",
        "value": job_hyps_per_s,
        "unit": "hypotheses/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": f"{args.solver} 5-pt E, {N} pts x {B} hyps per pair, "
                               + ("Gumbel top-k sampler (in-kernel Philox)" if args.sampler == "gumbel" else
                                  "top-down (Plackett-Luce) draw of the Gumbel top-k index sets")
                               + f", MSAC scoring with masks, test mode, {P} pairs/GPU/step",
                   "sampler": args.sampler,
                   "pairs_per_gpu": P, "points": N, "hypotheses_per_pair": B, "models_per_pair": M,
                   "solver": args.solver, "parallelism": f"pairs sharded over {world} GPU(s), no collective",
                   "streams": len(streams),
                   "issue": f"{len(streams)} batch(es) in flight, round-robin over {len(streams)} HIP stream(s)"},
        "pairs_per_s": world * P * args.steps / elapsed,
        "roofline": {"bound": "hbm", "kernel": "msac_score_kernel_f32_fast16", "valid_slot_fraction": valid_frac, "achieved": achieved,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "traffic_source": traffic_note,
                     "avg_launch_ms": k4_ms, "algorithmic_bytes_per_launch": bytes_per_launch,
                     "isolated": {"avg_launch_ms": iso_ms, "achieved": bytes_per_launch / (iso_ms * 1e-3) / 1e9,
                                  "frac": bytes_per_launch / (iso_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                  "note": "same launch, one stream, nothing else resident"},
                     "frac_of_measured_copy_peak_6290": achieved / 6290.0,
                     "valu_tflops": flops_per_launch / (iso_ms * 1e-3) / 1e12,
                     "valu_frac_of_157.3": flops_per_launch / (iso_ms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS},
        "check": {"mean_inlier_fraction_of_best_model": inl_frac},
        ("two_batches_in_flight" if len(streams) == 1 else "one_stream"): overlap,
        "with_final_refit": with_refit,
        "sampler_topdown": topdown,
    }
    if args.profile_kernels and rank == 0:
        result["kernel_breakdown_ms"] = kernel_breakdown(args, rn, matches, logits, K1, K2, ops)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(args, data)
    if rank == 0:
        print(json.dumps(result))
    if dist is not None:
        dist.destroy_process_group()


def kernel_breakdown(args, rn, matches, logits, K1, K2, ops):
    """Per-stage device time by HIP events (each stage synchronised): sampler / gather / solver / scoring / select."""
    import torch
    P, N, B = args.pairs, args.points, args.hyps
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
    if args.solver == "f8":
        out["K3_solver"], (F, v) = t(lambda: ops.solve_f8(smp))
        models, valid = F.unsqueeze(2), v.unsqueeze(2)
    elif args.solver == "stewenius":
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
