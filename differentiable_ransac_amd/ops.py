"""Functional wrappers over the C ABI (include/dransac.h) for batched [P, ...] GPU tensors, and
the torch.autograd.Function classes built on them.  Every function enqueues on the current
torch stream and returns freshly allocated torch tensors; nothing synchronises.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import _lib as L
from ._lib import c_int, c_uint64, ptr, stream


def _thr_tensor(threshold, P: int, like: torch.Tensor) -> torch.Tensor:
    if isinstance(threshold, torch.Tensor):
        t = threshold.to(device=like.device, dtype=like.dtype).reshape(-1)
        if t.numel() == 1 and P > 1:
            t = t.expand(P)
        return t.contiguous()
    return torch.full((P,), float(threshold), device=like.device, dtype=like.dtype)


# ------------------------------------------------------------------------------------------ K4 / K6
def msac_score(matches: torch.Tensor, models: torch.Tensor, threshold, want_masks: bool = True,
               valid: Optional[torch.Tensor] = None, path: int = 0, gate=None) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """matches [P,N,4], models [P,M,3,3] (or [P,M,9]) -> scores [P,M], masks [P,M,N] bool | None.
    valid [P,M] bool (optional): invalid slots are skipped (score 0, empty mask row).
    path: 0 = 1 = the general kernels; 2 (the matrix-core candidate filter of round 2, measured slower: DESIGN 2b) left the
    library in round 4 and is refused."""
    P, N, _ = matches.shape
    M = models.shape[1]
    matches = matches.contiguous()
    models = models.contiguous()
    thr = _thr_tensor(threshold, P, matches)
    scores = torch.empty((P, M), device=matches.device, dtype=matches.dtype)
    masks = torch.empty((P, M, N), device=matches.device, dtype=torch.bool) if want_masks else None
    v = None if valid is None else valid.contiguous().view(torch.uint8)
    if path not in (0, 1):
        raise L.DransacError("msac_score: path must be 0 or 1 (the general kernels); path 2, the matrix-core candidate filter of "
                             "round 2, was measured slower and left the library (scratch/k4_filter_kernel.patch)")
    if matches.dtype == torch.float32:
        # gate (a later round of a multi-round call): the blocks of terminated pairs (gate = RansacState) return at once, their
        # scores are never looked at (dr_ransac_update skips such pairs)
        L.call("dr_msac_score_f32", ptr(matches), ptr(models), ptr(v), ptr(thr), c_int(P), c_int(M), c_int(N), ptr(scores),
               ptr(masks), ptr(None if gate is None else gate.iters), ptr(None if gate is None else gate.max_iters), stream())
        return scores, masks
    L.call(f"dr_msac_score_{L.suffix(matches.dtype)}", ptr(matches), ptr(models), ptr(v), ptr(thr), c_int(P), c_int(M), c_int(N),
           ptr(scores), ptr(masks), stream())
    return scores, masks


def select_best(matches: torch.Tensor, models: torch.Tensor, scores: torch.Tensor, threshold,
                valid: Optional[torch.Tensor] = None):
    """Per pair: (best_idx [P] int32, best_score [P], best_model [P,3,3], best_mask [P,N] bool, inliers [P] int32)."""
    P, N, _ = matches.shape
    M = models.shape[1]
    dev, dt = matches.device, matches.dtype
    thr = _thr_tensor(threshold, P, matches)
    best_idx = torch.empty((P,), device=dev, dtype=torch.int32)
    best_score = torch.empty((P,), device=dev, dtype=dt)
    best_model = torch.empty((P, 3, 3), device=dev, dtype=dt)
    best_mask = torch.empty((P, N), device=dev, dtype=torch.bool)
    inliers = torch.empty((P,), device=dev, dtype=torch.int32)
    v = None if valid is None else valid.contiguous().view(torch.uint8)
    L.call(f"dr_select_best_{L.suffix(dt)}", ptr(matches.contiguous()), ptr(models.contiguous()), ptr(v),
           ptr(scores.contiguous()), ptr(thr), c_int(P), c_int(M), c_int(N), ptr(best_idx), ptr(best_score),
           ptr(best_model), ptr(best_mask), ptr(inliers), stream())
    return best_idx, best_score, best_model, best_mask, inliers


class RansacState:
    """Per-pair test-mode state kept on the device (best score / model / mask / inlier count, iteration counters)."""

    def __init__(self, P: int, N: int, max_iterations: int, device, dtype, _init: bool = True, packed: bool = False):
        alloc = torch.zeros if _init else torch.empty
        self.packed = None
        if packed and P == 1 and dtype == torch.float32 and not _init:
            # one pair, f32: everything a call hands out in ONE buffer -- model 36 B | score 4 | iterations 4 | mask N -- so that the
            # replayed drop-in call (ransac._GraphedCall) returns it without a gather launch (round 6)
            self.packed = torch.empty(44 + N, device=device, dtype=torch.uint8)
            self.best_model = self.packed[:36].view(torch.float32).view(1, 3, 3)
            self.best_score = self.packed[36:40].view(torch.float32)
            self.iters = self.packed[40:44].view(torch.int32)
            self.best_mask = self.packed[44:].view(torch.bool).view(1, N)
        else:
            self.best_score = alloc(P, device=device, dtype=dtype)
            self.best_model = (torch.eye(3, device=device, dtype=dtype).repeat(P, 1, 1) if _init
                               else torch.empty(P, 3, 3, device=device, dtype=dtype))
            self.best_mask = alloc(P, N, device=device, dtype=torch.bool)
            self.iters = alloc(P, device=device, dtype=torch.int32)
        self.best_inliers = alloc(P, device=device, dtype=torch.int32)
        self.max_iters = (torch.full((P,), float(max_iterations), device=device, dtype=torch.float64) if _init
                          else torch.empty(P, device=device, dtype=torch.float64))
        self.max_iterations = max_iterations


def ransac_init(P: int, N: int, max_iterations: int, threshold: float, K1: Optional[torch.Tensor],
                K2: Optional[torch.Tensor], device, dtype, seeds=None, packed: bool = False,
                race_logits: Optional[torch.Tensor] = None) -> Tuple[RansacState, torch.Tensor]:
    """dr_ransac_init: the per-pair state and the threshold normalised as ransac.py:49-53 (K1/K2 [3,3] or [P,3,3];
    None = threshold used as is), in ONE launch.  Returns (state, thr [P]).
    seeds = (DeviceSeed, n): the same launch also draws the next n sampler keys (DeviceSeed.next_block(n)); they are returned as
    `state.seeds` ([n] int64).  packed: one pair, f32 -- the state lives in one buffer (RansacState.packed).
    race_logits [P,N] f32: the same launch writes the per-pair weights of the one-logarithm sampler -> `state.race_ws`, to be handed
    to gumbel_topk_gather(race_ws=...) by every round of the call."""
    st = RansacState(P, N, max_iterations, device, dtype, _init=False, packed=packed)
    st.seeds = None
    st.race_ws = None
    if race_logits is not None:
        if race_logits.dtype != torch.float32 or race_logits.shape != (P, N) or not race_logits.is_contiguous():
            raise L.DransacError("ransac_init: race_logits must be contiguous f32 [P,N]")
        st.race_ws = torch.empty((P, N + 32), device=device, dtype=torch.float32)
    seed_state = None
    n_seeds = 0
    if seeds is not None:
        seed_state, n_seeds = seeds[0].state, int(seeds[1])
        st.seeds = torch.empty(n_seeds, dtype=torch.int64, device=device)
    thr = torch.empty(P, device=device, dtype=dtype)
    k_stride = 0
    if K1 is not None:
        K1 = K1.to(device=device, dtype=dtype).contiguous()
        K2 = K2.to(device=device, dtype=dtype).contiguous()
        if K1.dim() == 3 and K1.shape[0] != 1:
            if K1.shape[0] != P or K2.shape != K1.shape:
                raise ValueError("K1/K2 must be [3,3] or [P,3,3]")
            k_stride = 9
    L.call(f"dr_ransac_init_{L.suffix(dtype)}", ptr(K1), ptr(K2), c_int(k_stride), L.c_double(float(threshold)),
           c_int(P), c_int(N), c_int(max_iterations), ptr(thr), ptr(st.best_score), ptr(st.best_model),
           ptr(st.best_mask), ptr(st.best_inliers), ptr(st.iters), ptr(st.max_iters), ptr(seed_state), ptr(st.seeds),
           c_int(n_seeds), ptr(race_logits), ptr(st.race_ws), stream())
    return st, thr


def ransac_update(state: RansacState, matches, models, valid, scores, thr, B: int, k: int, confidence: float = 0.999,
                  eps: float = 1e-5, sub_models: int = 0) -> None:
    """K6 fused (dr_ransac_update): arg-max, best-model bookkeeping and adaptive termination, in place, one launch.
    sub_models > 0: the M models are consecutive sub-batches of B hypotheses (sub_models models each), walked in order with the
    stop rule of ransac.py:55-144 -- the state afterwards is the batch-by-batch loop's."""
    P, N, _ = matches.shape
    M = models.shape[1]
    if sub_models and M > sub_models * 512:
        raise L.DransacError("ransac_update: at most 512 sub-batches per launch")
    v = None if valid is None else valid.contiguous().view(torch.uint8)
    L.call(f"dr_ransac_update_{L.suffix(matches.dtype)}", ptr(matches), ptr(models.contiguous()), ptr(v),
           ptr(scores.contiguous()), ptr(thr), c_int(P), c_int(M), c_int(N), c_int(B), c_int(k),
           L.c_double(confidence), L.c_double(eps), c_int(state.max_iterations), ptr(state.best_score),
           ptr(state.best_model), ptr(state.best_mask), ptr(state.best_inliers), ptr(state.iters), ptr(state.max_iters),
           c_int(int(sub_models)), stream())


# ------------------------------------------------------------------------------------------ K1 / K1u / K2
class DeviceSeed:
    """The per-call sampler seed of the batched drivers, kept on the device: state = (base, calls) and `next()` launches
    dr_seed_next_n -> a one-word tensor holding base * 0x9E3779B97F4A7C15 + calls (mod 2^64), calls += 1.  The samplers
    accept such a tensor wherever they accept an int seed and read it when their kernel starts, so a step captured in a
    HIP graph (torch.cuda.graph) draws fresh hypotheses at every replay -- the same ones an eager driver with the same
    base seed draws at the same call number."""

    def __init__(self, base: int, device, calls: int = 0):
        def signed(v):
            v &= 2 ** 64 - 1
            return v - 2 ** 64 if v >= 2 ** 63 else v
        self.state = torch.tensor([signed(base), signed(calls)], dtype=torch.int64, device=device)

    def next(self) -> torch.Tensor:
        out = torch.empty(1, dtype=torch.int64, device=self.state.device)
        L.call("dr_seed_next_n", ptr(self.state), ptr(out), c_int(1), stream())
        return out

    def next_block(self, n: int) -> torch.Tensor:
        """the seeds of the next n calls from ONE launch (a multi-round call draws one per batch) as one [n] tensor of CONSECUTIVE
        integers: seeds[i:i + 1] is what next() would have returned for call i"""
        out = torch.empty(n, dtype=torch.int64, device=self.state.device)
        L.call("dr_seed_next_n", ptr(self.state), ptr(out), c_int(n), stream())
        return out

    def next_n(self, n: int):
        """next_block(n) as a list of n one-word tensors"""
        out = self.next_block(n)
        return [out[i:i + 1] for i in range(n)]


def _dev_seed(seed):
    if torch.is_tensor(seed):
        if seed.dtype != torch.int64 or seed.numel() != 1 or not seed.is_cuda:
            raise L.DransacError("a device seed is a one-element int64 CUDA tensor (DeviceSeed.next())")
        return True
    return False


def gumbel_topk(logits: Optional[torch.Tensor], B: int, k: int, tau: float = 1.0,
                gumbel: Optional[torch.Tensor] = None, seed: int = 0, N: Optional[int] = None,
                dense: bool = False, want_noise: bool = False, device=None, dtype=torch.float32, soft: bool = True,
                screen: Optional[bool] = None):
    """K1 forward.  logits [P,N] (or None = all-ones, then pass N/device/dtype); gumbel [P,B,N] explicit
    noise or None (in-kernel Philox keyed by `seed`).

    Returns dict(idx [P,B,k] int32 ascending, y_sel [P,B,k], lse [P,B]) plus, when `dense`,
    y_soft / ret [P,B,N], and when `want_noise`, gumbel [P,B,N] (the noise the kernel used).
    soft=False: index sets only (y_sel = lse = None) -- what test mode consumes; the same idx, a cheaper kernel.
    screen (index-only mode, rows longer than 2048 points): None = on when it can pay (B >= 64 rows per pair), False = off
    (A/B and tests: the index sets are the same either way)."""
    if not soft and dense:
        raise ValueError("the dense outputs need the soft-max statistics (soft=True)")
    if logits is not None:
        logits = logits.contiguous()
        P, N = logits.shape
        device, dtype = logits.device, logits.dtype
    else:
        P = 1 if gumbel is None else gumbel.shape[0]
        assert N is not None and device is not None
    if gumbel is not None:
        gumbel = gumbel.contiguous()
        assert gumbel.shape == (P, B, N) and gumbel.dtype == dtype
    idx = torch.empty((P, B, k), device=device, dtype=torch.int32)
    if (not soft and logits is not None and gumbel is None and not want_noise and dtype == torch.float32 and tau == 1.0
            and N > 2048 and N % 4 == 0 and k <= 5 and (B >= 64 if screen is None else screen)):
        # long rows, index sets only: the screened one-pass kernel (dr_gumbel_topk_index_f32; same index sets, bit for bit)
        ws = torch.empty(((N + 32) * P,), device=device, dtype=torch.int32)
        ds = _dev_seed(seed)
        L.call("dr_gumbel_topk_index_f32", ptr(logits), c_uint64(0 if ds else seed & (2 ** 64 - 1)), ptr(seed if ds else None),
               L.c_float(tau), c_int(P), c_int(B), c_int(N), c_int(k), ptr(idx), ptr(ws), stream())
        return dict(idx=idx, y_sel=None, lse=None)
    y_sel = torch.empty((P, B, k), device=device, dtype=dtype) if soft else None
    lse = torch.empty((P, B), device=device, dtype=dtype) if soft else None
    y_soft = torch.empty((P, B, N), device=device, dtype=dtype) if dense else None
    ret = torch.empty((P, B, N), device=device, dtype=dtype) if dense else None
    noise = torch.empty((P, B, N), device=device, dtype=dtype) if want_noise else None
    ds = _dev_seed(seed)
    if ds and (gumbel is not None or dense or want_noise or logits is None):
        raise L.DransacError("a device seed serves the in-kernel noise of given logits only (no explicit noise / dense outputs)")
    L.call(f"dr_gumbel_topk_fwd_{L.suffix(dtype)}", ptr(logits), ptr(gumbel), c_uint64(0 if ds else seed & (2 ** 64 - 1)),
           ptr(seed if ds else None), L.scalar(dtype, tau), c_int(P), c_int(B), c_int(N), c_int(k), ptr(idx), ptr(y_sel), ptr(lse),
           ptr(y_soft), ptr(ret), ptr(noise), stream())
    out = dict(idx=idx, y_sel=y_sel, lse=lse)
    if dense:
        out.update(y_soft=y_soft, ret=ret)
    if want_noise:
        out["gumbel"] = noise
    return out


import os as _os
# Round 5: the screened register kernel for rows of <= 2048 points (dr_gumbel_topk_gather_f32 with a screen_ws workspace) is built,
# bit-identical (tests/test_gpu_round5.py) and SLOWER than the unscreened one at the shapes measured -- 199.5 vs 164.1 us at 128
# pairs x 1024 rows x 2000 points, 57.4 vs 45.2 at 32 pairs, 25.6 vs 18.2 at one pair (scratch/ab_k1_screen.py), in both of its forms
# (words parked in LDS + one evaluation per lane and round: 200.8; slot-wise wave masks, no parking, no dependent load: 199.5):
# ~16 of 2000 points pass, but Philox (40 % of the row's instructions) cannot be screened and the unscreened transform is 7 vector
# instructions per element -- a branch per element slot costs what it saves.  Off unless asked for (screen=True, DRANSAC_SCREEN_SHORT=1).
SCREEN_SHORT_ROWS = _os.environ.get("DRANSAC_SCREEN_SHORT", "0") == "1"


# Round 6: the exponential-race form of the index-only sampler (one logarithm per element; dr_gumbel_topk_gather_f32's race_ws).
# Same top-k up to the rounding of near-ties; off = the two-logarithm form of rounds 1-5 (A/B runs, tests: DRANSAC_K1_RACE=0).
K1_RACE = _os.environ.get("DRANSAC_K1_RACE", "1") != "0"
K1_RACE_SOFT = _os.environ.get("DRANSAC_K1_RACE_SOFT", "1") != "0"   # ... in train mode (SampleGather's fused launch)
_RACE_MIN = tuple(int(v) for v in _os.environ.get("DRANSAC_K1_RACE_MIN", "32768,32").split(","))   # (rows, pairs) from which it is automatic


def race_form_pays(P: int, B: int, N: int, tau: float) -> bool:
    """the automatic choice of the one-logarithm sampler: rows the register kernel serves, from 32 pairs / 32 768 rows on (the
    prologue costs a launch, the form saves ~0.12 us per 1024 rows of 2000 points: scratch/runs/r6_gpu_o.sh)"""
    return K1_RACE and P * B >= _RACE_MIN[0] and P >= _RACE_MIN[1] and N <= 2048 and N % 4 == 0 and tau == 1.0


def gumbel_topk_gather(matches: torch.Tensor, logits: torch.Tensor, B: int, k: int, tau: float = 1.0, seed=0, gate=None,
                       screen: Optional[bool] = None, sub: int = 0, race: Optional[bool] = None, race_ws: Optional[torch.Tensor] = None):
    """K1 (index sets only, in-kernel noise) + K2 in one call: matches [P,N,4] f32, logits [P,N] f32 ->
    (idx [P,B,k] int32 ascending, samples [P,B,k,4] = matches[p, idx]).  What test mode asks of sampler + gather
    (ransac.py:58-65); `seed`: int or a DeviceSeed.next() tensor.
    sub > 0 (super-rounds): the B rows are consecutive sub-batches of `sub` rows; row b draws what row b % sub of the call with
    seed + b // sub draws (the drivers' per-call seeds are consecutive integers).
    race (None = automatic, race_form_pays): the one-logarithm exponential-race form of the same top-k (rows of <= 2048 points);
    race_ws: its per-pair weights, already computed for these logits by ransac_init(race_logits=...)."""
    if matches.dtype != torch.float32 or logits.dtype != torch.float32 or matches.shape[-1] != 4:
        raise L.DransacError("gumbel_topk_gather: f32 two-view correspondences [P,N,4]")
    matches, logits = matches.contiguous(), logits.contiguous()
    P, N = logits.shape
    idx = torch.empty((P, B, k), device=logits.device, dtype=torch.int32)
    samples = torch.empty((P, B, k, 4), device=logits.device, dtype=torch.float32)
    dev_seed = _dev_seed(seed)
    # round 5: rows of <= 2048 points through the SCREENED register kernel (a workspace of thresholds per point: same index sets)
    want_screen = SCREEN_SHORT_ROWS if screen is None else screen
    ws = (torch.empty((P, N + 32), device=logits.device, dtype=torch.int32)
          if want_screen and N <= 2048 and N % 4 == 0 and tau == 1.0 and k <= 5 and B >= 64 else None)
    # race_ws: a workspace dr_ransac_init has already filled for these logits (ransac_init(race_logits=...)): no prologue launch
    ready = race_ws is not None and ws is None and N <= 2048 and N % 4 == 0 and tau == 1.0
    want_race = not ready and (race_form_pays(P, B, N, tau) if race is None else race) and ws is None and N <= 2048 and N % 4 == 0 and tau == 1.0
    rws = race_ws if ready else (torch.empty((P, N + 32), device=logits.device, dtype=torch.float32) if want_race else None)
    # (gate: a later round of a multi-round call, terminated pairs are skipped)
    L.call("dr_gumbel_topk_gather_f32", ptr(logits), ptr(matches), c_uint64(0 if dev_seed else seed & (2 ** 64 - 1)),
           ptr(seed if dev_seed else None), L.c_float(tau), c_int(P), c_int(B), c_int(N), c_int(k), ptr(idx), ptr(samples),
           ptr(ws), ptr(None if gate is None else gate.iters), ptr(None if gate is None else gate.max_iters),
           c_int(0 if sub >= B else int(sub)), ptr(rws), c_int(1 if ready else 0), stream())
    return idx, samples


def gumbel_topk_bwd(logits, gumbel, seed, tau, idx, lse, a_sel):
    """grad_logits [P,N] from a_sel [P,B,k] (f32; f64 with a by-value seed or explicit noise)."""
    P, B, k = idx.shape
    N = logits.shape[1]
    grad = torch.empty_like(logits)
    if logits.dtype == torch.float64:
        if _dev_seed(seed):
            raise L.DransacError("gumbel_topk_bwd: device seeds serve f32 only")
        L.call("dr_gumbel_topk_bwd_f64", ptr(logits.contiguous()), ptr(gumbel), c_uint64(seed & (2 ** 64 - 1)),
               L.c_double(tau), c_int(P), c_int(B), c_int(N), c_int(k), ptr(idx), ptr(lse.contiguous()),
               ptr(a_sel.to(torch.float64).contiguous()), ptr(grad), stream())
        return grad
    ds = _dev_seed(seed)
    L.call("dr_gumbel_topk_bwd_f32", ptr(logits.contiguous()), ptr(None if ds else gumbel), c_uint64(0 if ds else seed & (2 ** 64 - 1)),
           ptr(seed if ds else None), L.c_float(tau), c_int(P), c_int(B), c_int(N), c_int(k), ptr(idx), ptr(lse),
           ptr(a_sel.contiguous()), ptr(grad), stream())
    return grad


def topdown_sample(logits: Optional[torch.Tensor], B: int, k: int, seed: int = 0, N: Optional[int] = None, P: int = 1,
                   device=None) -> torch.Tensor:
    """K1, inference variant (dr_topdown_sample): the Gumbel top-k INDEX SET drawn top-down -- k sequential draws
    without replacement from softmax(logits), the exact distribution of the top-k of logits + iid Gumbel(0,1) for any
    tau > 0 -- in O(B k log N).  logits [P,N] (or None = uniform: pass N, P, device) -> idx [P,B,k] int32 ascending.
    No y_sel / lse: train mode and weighted mode need the whole noise row (use gumbel_topk)."""
    if logits is not None:
        logits = logits.contiguous()
        P, N = logits.shape
        device = logits.device
        sfx = L.suffix(logits.dtype)
    else:
        sfx = "f32"
    ws = torch.empty((P, N), device=device, dtype=torch.float64)
    idx = torch.empty((P, B, k), device=device, dtype=torch.int32)
    ds = _dev_seed(seed)
    if sfx == "f32":
        L.call("dr_topdown_sample_f32", ptr(logits), c_uint64(0 if ds else seed & (2 ** 64 - 1)), ptr(seed if ds else None), c_int(P),
               c_int(B), c_int(N), c_int(k), ptr(ws), ptr(idx), stream())
        return idx
    if ds:
        raise L.DransacError("device seeds: f32 logits")
    L.call("dr_topdown_sample_f64", ptr(logits), c_uint64(seed & (2 ** 64 - 1)), c_int(P), c_int(B), c_int(N), c_int(k),
           ptr(ws), ptr(idx), stream())
    return idx


def uniform_sample(P: int, B: int, k: int, N: int, seed: int, device) -> torch.Tensor:
    """K1u: idx [P,B,k] int32 ~ U{0..N-2} (uniform_sampler.py:15-19 semantics)."""
    idx = torch.empty((P, B, k), device=device, dtype=torch.int32)
    ds = _dev_seed(seed)
    L.call("dr_uniform_sample", c_uint64(0 if ds else seed & (2 ** 64 - 1)), ptr(seed if ds else None), c_int(P), c_int(B), c_int(k),
           c_int(N), ptr(idx), stream())
    return idx


def gather(matches: torch.Tensor, idx: torch.Tensor, y_sel: Optional[torch.Tensor] = None) -> torch.Tensor:
    """K2 forward: matches [P,N,c], idx [P,B,k] -> samples [P,B,k,c] (times the straight-through value)."""
    P, N, c = matches.shape
    _, B, k = idx.shape
    out = torch.empty((P, B, k, c), device=matches.device, dtype=matches.dtype)
    L.call(f"dr_gather_fwd_{L.suffix(matches.dtype)}", ptr(matches.contiguous()), ptr(idx), ptr(y_sel), c_int(P),
           c_int(N), c_int(B), c_int(k), c_int(c), ptr(out), stream())
    return out


def gather_bwd(matches, idx, y_sel, grad_samples, grad_w=None, want_grad_matches=False):
    P, N, c = matches.shape
    _, B, k = idx.shape
    a_sel = torch.empty((P, B, k), device=matches.device, dtype=matches.dtype)
    gm = torch.zeros_like(matches) if want_grad_matches else None
    dt = matches.dtype
    L.call(f"dr_gather_bwd_{L.suffix(dt)}", ptr(matches.contiguous()), ptr(idx), ptr(y_sel), ptr(grad_samples.to(dt).contiguous()),
           ptr(None if grad_w is None else grad_w.to(dt).contiguous()), c_int(P), c_int(N), c_int(B), c_int(k), c_int(c),
           ptr(a_sel), ptr(gm), stream())
    return a_sel, gm


FUSED_SAMPLE_GATHER = _os.environ.get("DRANSAC_FUSED_SAMPLE_GATHER", "1") != "0"   # A/B and tests: off = the two-launch forward / backward of rounds 1-4


class SampleGather(torch.autograd.Function):
    """K1+K2 fused at the autograd level: (matches [P,N,c], logits [P,N]) -> samples [P,B,k,c], weights [P,B,k].

    Forward: dr_gumbel_topk_fwd + dr_gather_fwd (no [B,N] tensor is materialised).
    Backward: dr_gather_bwd -> a_sel, dr_gumbel_topk_bwd -> grad_logits (f32 and f64)."""

    @staticmethod
    def forward(ctx, matches, logits, B, k, tau, gumbel, seed):
        ctx.set_materialize_grads(False)   # unused / non-differentiable outputs arrive as None, not as zero-filled tensors
        if logits.dtype == torch.float64 and logits.requires_grad and gumbel is None and _dev_seed(seed):
            # refused HERE, not at backward time (round-4 advice): the f64 sampler backward takes a by-value seed or explicit noise
            raise L.DransacError("SampleGather: f64 logits with a device seed have no backward (device seeds serve f32 only)")
        fused = (FUSED_SAMPLE_GATHER and gumbel is None and logits.dtype == torch.float32 and matches.dtype == torch.float32
                 and matches.shape[-1] == 4)
        if fused:
            # round 5: sampler (soft-max statistics) + gather in ONE launch, and one launch back (dr_gumbel_topk_gather_soft_f32 /
            # dr_gumbel_topk_gather_bwd_f32): same index sets, weights and samples as the two-launch form, bit for bit
            matches, logits = matches.contiguous(), logits.contiguous()
            P, N = logits.shape
            r = dict(idx=torch.empty((P, B, k), device=logits.device, dtype=torch.int32),
                     y_sel=torch.empty((P, B, k), device=logits.device, dtype=torch.float32),
                     lse=torch.empty((P, B), device=logits.device, dtype=torch.float32))
            samples = torch.empty((P, B, k, 4), device=logits.device, dtype=torch.float32)
            ds = _dev_seed(seed)
            # round 6: the one-logarithm form in train mode too (keys, winners and soft-max statistics from one logarithm and one
            # reciprocal per element), where its weights prologue pays
            rws = (torch.empty((P, N + 32), device=logits.device, dtype=torch.float32)
                   if K1_RACE_SOFT and race_form_pays(P, B, N, tau) else None)
            L.call("dr_gumbel_topk_gather_soft_f32", ptr(logits), ptr(matches), c_uint64(0 if ds else seed & (2 ** 64 - 1)),
                   ptr(seed if ds else None), L.c_float(tau), c_int(P), c_int(B), c_int(N), c_int(k), ptr(r["idx"]), ptr(r["y_sel"]),
                   ptr(r["lse"]), ptr(samples), ptr(rws), stream())
        else:
            r = gumbel_topk(logits, B, k, tau, gumbel, seed)
            samples = gather(matches, r["idx"], r["y_sel"])
        ctx.save_for_backward(matches, logits, r["idx"], r["y_sel"], r["lse"], gumbel if gumbel is not None else
                              torch.empty(0, device=logits.device))
        ctx.cfg = (tau, seed, gumbel is not None)
        ctx.fused = fused
        ctx.mark_non_differentiable(r["idx"])
        return samples, r["y_sel"], r["idx"]

    @staticmethod
    def backward(ctx, g_samples, g_w, _g_idx):
        if g_samples is None and g_w is None:
            return (None,) * 7
        matches, logits, idx, y_sel, lse, gumbel = ctx.saved_tensors
        if g_samples is None:
            g_samples = torch.zeros(idx.shape + (matches.shape[-1],), device=matches.device, dtype=matches.dtype)
        tau, seed, has_noise = ctx.cfg
        if ctx.fused and not ctx.needs_input_grad[0]:
            P, B, k = idx.shape
            gl = torch.empty_like(logits)
            ds = _dev_seed(seed)
            L.call("dr_gumbel_topk_gather_bwd_f32", ptr(logits), ptr(matches), c_uint64(0 if ds else seed & (2 ** 64 - 1)),
                   ptr(seed if ds else None), L.c_float(tau), c_int(P), c_int(B), c_int(logits.shape[1]), c_int(k), ptr(idx), ptr(lse),
                   ptr(g_samples.contiguous()), ptr(None if g_w is None else g_w.contiguous()), ptr(gl), stream())
            return None, gl, None, None, None, None, None
        a_sel, gm = gather_bwd(matches, idx, y_sel, g_samples, g_w, want_grad_matches=ctx.needs_input_grad[0])
        gl = gumbel_topk_bwd(logits, gumbel if has_noise else None, seed, tau, idx, lse, a_sel)
        return gm, gl, None, None, None, None, None


# ------------------------------------------------------------------------------------------ K3 solvers
def _flat_samples(samples: torch.Tensor, c: int):
    s = samples.reshape(-1, samples.shape[-2], c).contiguous()
    return s, s.shape[0], s.shape[1]


def solve_nister5(samples: torch.Tensor, weights: Optional[torch.Tensor] = None, path: int = 0):
    """samples [..., n>=5, 4] -> models [..., 10, 3, 3], valid [..., 10] bool (real solutions, ascending root).
    path (f32 minimal samples only): 0 automatic, 1 lane-pair kernel, 2 two-phase kernel (include/dransac.h)."""
    s, Bt, n = _flat_samples(samples, 4)
    lead = samples.shape[:-2]
    models = torch.empty((Bt, 10, 3, 3), device=s.device, dtype=s.dtype)
    valid = torch.empty((Bt, 10), device=s.device, dtype=torch.bool)
    w = None if weights is None else weights.reshape(Bt, n).to(s.dtype).contiguous()
    if path != 0 and (n != 5 or s.dtype != torch.float32):
        raise L.DransacError("an explicit five-point kernel path exists for f32 minimal samples only")
    if s.dtype == torch.float32:
        L.call("dr_solve_nister5_f32", ptr(s), ptr(w), c_int(Bt), c_int(n), ptr(models), ptr(None), ptr(valid), c_int(path), c_int(0),
               ptr(None), ptr(None), stream())
    else:
        L.call(f"dr_solve_nister5_{L.suffix(s.dtype)}", ptr(s), ptr(w), c_int(Bt), c_int(n), ptr(models), ptr(valid), stream())
    return models.reshape(*lead, 10, 3, 3), valid.reshape(*lead, 10)


def solve_nister5_hp(samples: torch.Tensor, weights: Optional[torch.Tensor] = None, path: int = 0):
    """Minimal f32 samples [..., 5, 4] -> (models f32, models f64, valid): the train-mode entry, one launch."""
    s, Bt, n = _flat_samples(samples, 4)
    if n != 5 or s.dtype != torch.float32:
        raise L.DransacError("solve_nister5_hp takes f32 minimal samples (5 correspondences)")
    lead = samples.shape[:-2]
    models = torch.empty((Bt, 10, 3, 3), device=s.device, dtype=torch.float32)
    m64 = torch.empty((Bt, 10, 3, 3), device=s.device, dtype=torch.float64)
    valid = torch.empty((Bt, 10), device=s.device, dtype=torch.bool)
    w = None if weights is None else weights.reshape(Bt, n).to(s.dtype).contiguous()
    L.call("dr_solve_nister5_f32", ptr(s), ptr(w), c_int(Bt), c_int(5), ptr(models), ptr(m64), ptr(valid), c_int(path), c_int(0),
           ptr(None), ptr(None), stream())
    return models.reshape(*lead, 10, 3, 3), m64.reshape(*lead, 10, 3, 3), valid.reshape(*lead, 10)


def solve_essential_gated(samples: torch.Tensor, which: str, gate):
    """A later round of a multi-round test-mode call: samples [P,B,5,4] f32 -> (models [P,B,10,3,3], valid [P,B,10]); the
    blocks of pairs that have terminated (gate = RansacState: iters >= max_iters) return at once and leave their part of the
    outputs unwritten -- dr_ransac_update never looks at it (the gate arguments of dr_solve_nister5_f32 / dr_solve_stewenius5_f32)."""
    P, B = samples.shape[0], samples.shape[1]
    s = samples.reshape(P * B, 5, 4).contiguous()
    models = torch.empty((P * B, 10, 3, 3), device=s.device, dtype=torch.float32)
    valid = torch.empty((P * B, 10), device=s.device, dtype=torch.bool)
    if which == "nister":
        L.call("dr_solve_nister5_f32", ptr(s), ptr(None), c_int(P * B), c_int(5), ptr(models), ptr(None), ptr(valid), c_int(0), c_int(B),
               ptr(gate.iters), ptr(gate.max_iters), stream())
    else:
        L.call("dr_solve_stewenius5_f32", ptr(s), c_int(P * B), ptr(models), ptr(valid), c_int(0), c_int(B), ptr(gate.iters),
               ptr(gate.max_iters), stream())
    return models.reshape(P, B, 10, 3, 3), valid.reshape(P, B, 10)


def debug_real_roots10(coef: torch.Tensor, method: int = 1):
    """Test hook (dr_debug_real_roots10): coef [n,11] f64 ascending -> roots [n,2,10] f64, counts [n,2] int32; half 0 = |z| <= 1,
    half 1 = |z| > 1; method 0 = derivative chain, 1 = Sturm isolation (what the five-point kernels run)."""
    coef = coef.to(torch.float64).contiguous()
    n = coef.shape[0]
    roots = torch.zeros((n, 2, 10), device=coef.device, dtype=torch.float64)
    counts = torch.zeros((n, 2), device=coef.device, dtype=torch.int32)
    L.call("dr_debug_real_roots10", ptr(coef), c_int(n), c_int(method), ptr(roots), ptr(counts), stream())
    return roots, counts


def solve_stewenius5(samples: torch.Tensor, path: int = 0):
    """samples [..., 5, 4] -> models [..., 10, 3, 3], valid [..., 10].  path: as solve_nister5 (f32 only)."""
    s, Bt, n = _flat_samples(samples, 4)
    if n != 5:
        raise L.DransacError("the Stewenius solver takes exactly 5 correspondences per sample")
    lead = samples.shape[:-2]
    models = torch.empty((Bt, 10, 3, 3), device=s.device, dtype=s.dtype)
    valid = torch.empty((Bt, 10), device=s.device, dtype=torch.bool)
    if path != 0 and s.dtype != torch.float32:
        raise L.DransacError("an explicit five-point kernel path exists for f32 minimal samples only")
    if s.dtype == torch.float32:
        L.call("dr_solve_stewenius5_f32", ptr(s), c_int(Bt), ptr(models), ptr(valid), c_int(path), c_int(0), ptr(None), ptr(None), stream())
    else:
        L.call(f"dr_solve_stewenius5_{L.suffix(s.dtype)}", ptr(s), c_int(Bt), ptr(models), ptr(valid), stream())
    return models.reshape(*lead, 10, 3, 3), valid.reshape(*lead, 10)


def solve_f8(samples: torch.Tensor, weights: Optional[torch.Tensor] = None):
    """samples [..., n>=8, 4] -> F [..., 3, 3], valid [...] bool."""
    s, Bt, n = _flat_samples(samples, 4)
    lead = samples.shape[:-2]
    models = torch.empty((Bt, 3, 3), device=s.device, dtype=s.dtype)
    valid = torch.empty((Bt,), device=s.device, dtype=torch.bool)
    w = None if weights is None else weights.reshape(Bt, n).to(s.dtype).contiguous()
    L.call(f"dr_solve_f8_{L.suffix(s.dtype)}", ptr(s), ptr(w), c_int(Bt), c_int(n), ptr(models), ptr(valid), stream())
    return models.reshape(*lead, 3, 3), valid.reshape(lead)


def solve_f8_uniform(matches: torch.Tensor, B: int, seed):
    """K1u + K2 + K3f8 in one launch (f32): matches [P,N,4] -> (idx [P,B,8] int32, F [P,B,3,3], valid [P,B] bool); the index sets
    are those of uniform_sample(P, B, 8, N, seed), the models those of solve_f8(gather(matches, idx))."""
    if matches.dtype != torch.float32 or matches.shape[-1] != 4:
        raise L.DransacError("solve_f8_uniform: f32 correspondences [P,N,4]")
    P, N, _ = matches.shape
    idx = torch.empty((P, B, 8), device=matches.device, dtype=torch.int32)
    models = torch.empty((P, B, 3, 3), device=matches.device, dtype=torch.float32)
    valid = torch.empty((P, B), device=matches.device, dtype=torch.bool)
    ds = _dev_seed(seed)
    L.call("dr_solve_f8_uniform_f32", ptr(matches.contiguous()), c_uint64(0 if ds else seed & (2 ** 64 - 1)), ptr(seed if ds else None),
           c_int(P), c_int(B), c_int(N), ptr(idx), ptr(models), ptr(valid), stream())
    return idx, models, valid


def solve_f7(samples: torch.Tensor):
    s, Bt, n = _flat_samples(samples, 4)
    if n != 7:
        raise L.DransacError("the 7-point solver takes exactly 7 correspondences per sample")
    lead = samples.shape[:-2]
    models = torch.empty((Bt, 4, 3, 3), device=s.device, dtype=s.dtype)
    valid = torch.empty((Bt, 4), device=s.device, dtype=torch.bool)
    L.call(f"dr_solve_f7_{L.suffix(s.dtype)}", ptr(s), c_int(Bt), ptr(models), ptr(valid), stream())
    return models.reshape(*lead, 4, 3, 3), valid.reshape(*lead, 4)


def solve_rigid(samples: torch.Tensor, weights: Optional[torch.Tensor] = None, flag: bool = True):
    """samples [..., n>=3, 6] -> model [...,4,4], R [...,3,3], t [...,3], scale [...], valid [...] bool."""
    s, Bt, n = _flat_samples(samples, 6)
    lead = samples.shape[:-2]
    dev, dt = s.device, s.dtype
    model = torch.empty((Bt, 4, 4), device=dev, dtype=dt)
    R = torch.empty((Bt, 3, 3), device=dev, dtype=dt)
    t = torch.empty((Bt, 3), device=dev, dtype=dt)
    scale = torch.empty((Bt,), device=dev, dtype=dt)
    valid = torch.empty((Bt,), device=dev, dtype=torch.bool)
    w = None if weights is None else weights.reshape(Bt, n).to(dt).contiguous()
    L.call(f"dr_solve_rigid_{L.suffix(dt)}", ptr(s), ptr(w), c_int(Bt), c_int(n), c_int(1 if flag else 0), ptr(model),
           ptr(R), ptr(t), ptr(scale), ptr(valid), stream())
    return (model.reshape(*lead, 4, 4), R.reshape(*lead, 3, 3), t.reshape(*lead, 3), scale.reshape(lead),
            valid.reshape(lead))


def solve_rigid_gather(matches: torch.Tensor, idx: torch.Tensor, flag: bool = True, zero_sums: Optional[torch.Tensor] = None):
    """K2 + K3r in one launch (f32): matches [P,N,6], idx [P,B,k] int32 -> (model [P,B,4,4], valid [P,B] bool), the samples read
    through the index sets; zero_sums [P,B] (optional) is cleared by the same launch (the residual sums of the round:
    rigid_residual(..., res=zero_sums) then needs no memset)."""
    if matches.dtype != torch.float32 or matches.shape[-1] != 6:
        raise L.DransacError("solve_rigid_gather: f32 correspondences [P,N,6]")
    P, N, _ = matches.shape
    _, B, k = idx.shape
    model = torch.empty((P, B, 4, 4), device=matches.device, dtype=torch.float32)
    valid = torch.empty((P, B), device=matches.device, dtype=torch.bool)
    L.call("dr_solve_rigid_gather_f32", ptr(matches.contiguous()), ptr(idx.contiguous()), c_int(P), c_int(B), c_int(N), c_int(k),
           c_int(1 if flag else 0), ptr(model), ptr(None), ptr(None), ptr(None), ptr(valid), ptr(zero_sums), stream())
    return model, valid


def rigid_residual(pts: torch.Tensor, models: torch.Tensor, threshold: float = 0.03, want_masks: bool = True,
                   res: Optional[torch.Tensor] = None):
    """pts [P,N,6], models [P,M,4,4] -> res_sum [P,M], masks [P,M,N] bool | None.
    res (f32, optional): a [P,M] tensor of ZEROS to add the sums to (no memset launch): solve_rigid_gather(..., zero_sums=res)."""
    P, N, _ = pts.shape
    M = models.shape[1]
    masks = torch.empty((P, M, N), device=pts.device, dtype=torch.bool) if want_masks else None
    if res is not None:
        if pts.dtype != torch.float32 or res.shape != (P, M) or res.dtype != torch.float32:
            raise L.DransacError("rigid_residual: res must be an f32 [P,M] tensor of zeros")
        L.call("dr_rigid_residual_f32", ptr(pts.contiguous()), ptr(models.contiguous()), L.c_float(threshold), c_int(P),
               c_int(M), c_int(N), ptr(res), ptr(masks), c_int(1), stream())
        return res, masks
    res = torch.empty((P, M), device=pts.device, dtype=pts.dtype)
    if pts.dtype == torch.float32:
        L.call("dr_rigid_residual_f32", ptr(pts.contiguous()), ptr(models.contiguous()), L.c_float(threshold), c_int(P), c_int(M),
               c_int(N), ptr(res), ptr(masks), c_int(0), stream())
    else:
        L.call(f"dr_rigid_residual_{L.suffix(pts.dtype)}", ptr(pts.contiguous()), ptr(models.contiguous()),
               L.scalar(pts.dtype, threshold), c_int(P), c_int(M), c_int(N), ptr(res), ptr(masks), stream())
    return res, masks


def ransac3d_update(pts: torch.Tensor, models: torch.Tensor, valid: Optional[torch.Tensor], res: torch.Tensor,
                    threshold: float, best_res: Optional[torch.Tensor], best_model: Optional[torch.Tensor],
                    best_mask: Optional[torch.Tensor] = None):
    """K6 of the 3-D path (dr_ransac3d_update): per pair the valid model with the smallest residual sum replaces the state
    where it is strictly better.  pts [P,N,6], models [P,M,4,4], valid [P,M] | None, res [P,M]; state best_res [P],
    best_model [P,4,4] -> NEW (best_res, best_model) tensors (the small state is ping-ponged), best_mask [P,N] updated in
    place, idx [P] int32 = the round's winner or -1."""
    P, N, _ = pts.shape
    M = models.shape[1]
    # best_res = best_model = None: the first round (state = +inf / identity / empty mask, nothing to allocate or fill)
    out_res = torch.empty((P,), device=pts.device, dtype=pts.dtype)
    out_model = torch.empty((P, 4, 4), device=pts.device, dtype=pts.dtype)
    idx = torch.empty((P,), device=pts.device, dtype=torch.int32)
    v = None if valid is None else valid.contiguous().view(torch.uint8)
    mk = None if best_mask is None else best_mask.view(torch.uint8)
    L.call(f"dr_ransac3d_update_{L.suffix(pts.dtype)}", ptr(pts.contiguous()), ptr(models.contiguous()), ptr(v),
           ptr(res.contiguous()), L.scalar(pts.dtype, threshold), c_int(P), c_int(M), c_int(N),
           ptr(None if best_res is None else best_res.contiguous()), ptr(None if best_model is None else best_model.contiguous()),
           ptr(out_res), ptr(out_model), ptr(mk), ptr(idx), stream())
    return out_res, out_model, idx


def select_closest(models: torch.Tensor, valid: Optional[torch.Tensor], gt: torch.Tensor, want_keep: bool = False):
    """K5: models [P,B,S,3,3], valid [P,B,S] | None, gt [P,3,3] -> chosen [P,B,3,3], which [P,B] int32 (+ keep [P,B] bool =
    `which >= 0` from the same launch when want_keep)."""
    P, B, S = models.shape[:3]
    chosen = torch.empty((P, B, 3, 3), device=models.device, dtype=models.dtype)
    which = torch.empty((P, B), device=models.device, dtype=torch.int32)
    v = None if valid is None else valid.contiguous().view(torch.uint8)
    keep = torch.empty((P, B), device=models.device, dtype=torch.bool) if want_keep else None
    L.call(f"dr_select_closest_{L.suffix(models.dtype)}", ptr(models.contiguous()), ptr(v),
           ptr(gt.to(models.dtype).contiguous()), c_int(P), c_int(B), c_int(S), ptr(chosen), ptr(which), ptr(keep), stream())
    return (chosen, which, keep) if want_keep else (chosen, which)


def refit_essential(matches: torch.Tensor, mask: Optional[torch.Tensor] = None):
    """K7 (E): five-point solver on all (masked) points of every pair as one sample, f64 inside.
    matches [P,N,4] -> models [P,10,3,3], valid [P,10]."""
    P, N, _ = matches.shape
    models = torch.empty((P, 10, 3, 3), device=matches.device, dtype=matches.dtype)
    valid = torch.empty((P, 10), device=matches.device, dtype=torch.bool)
    mk = None if mask is None else mask.contiguous().view(torch.uint8)
    L.call(f"dr_refit_essential_{L.suffix(matches.dtype)}", ptr(matches.contiguous()), ptr(mk), c_int(P), c_int(N),
           ptr(models), ptr(valid), stream())
    return models, valid


def refit_accept(matches: torch.Tensor, cand: torch.Tensor, cand_valid: Optional[torch.Tensor], thr: torch.Tensor,
                 best_score: torch.Tensor, best_model: torch.Tensor) -> None:
    """K7 acceptance in one launch (dr_refit_accept): scores cand [P,S,3,3] and replaces best_score [P] / best_model
    [P,3,3] IN PLACE where a candidate scores strictly higher (ransac.py:173-185)."""
    P, N, _ = matches.shape
    S = cand.shape[1]
    cv = None if cand_valid is None else cand_valid.contiguous().view(torch.uint8)
    L.call(f"dr_refit_accept_{L.suffix(matches.dtype)}", ptr(matches.contiguous()), ptr(cand.contiguous()), ptr(cv), ptr(thr),
           c_int(P), c_int(S), c_int(N), ptr(best_score), ptr(best_model), stream())


def refit_fundamental(matches: torch.Tensor, mask: Optional[torch.Tensor] = None, weights: Optional[torch.Tensor] = None):
    """K7 (F): Hartley-normalised LSQ 8-point on the masked points of every pair.  -> F [P,3,3], valid [P].
    weights [P,N] (optional): per-point row weights, the `soft_weights[0, inlier_indices[0]]` of ransac.py:151-153."""
    P, N, _ = matches.shape
    models = torch.empty((P, 3, 3), device=matches.device, dtype=matches.dtype)
    valid = torch.empty((P,), device=matches.device, dtype=torch.bool)
    mk = None if mask is None else mask.contiguous().view(torch.uint8)
    if weights is not None and weights.shape != (P, N):
        raise L.DransacError("refit weights are [P,N], one per point")
    L.call(f"dr_refit_fundamental_{L.suffix(matches.dtype)}", ptr(matches.contiguous()), ptr(mk),
           ptr(None if weights is None else weights.to(matches.dtype).contiguous()), c_int(P), c_int(N), ptr(models), ptr(valid),
           stream())
    return models, valid


def soft_weights_row0(logits: torch.Tensor, k: int, tau: float, gumbel: Optional[torch.Tensor], seed: int) -> torch.Tensor:
    """y_soft of hypothesis 0 of a sampler call, [P,N]: softmax((logits + noise[:, 0]) / tau) with the SAME noise row the call
    with this seed (or this explicit noise) drew -- the in-kernel Philox counters are (element, hypothesis, pair), so row 0 does
    not depend on the batch size.  What ransac.py:151-153 indexes as `soft_weights[0, ...]`."""
    g = None if gumbel is None else gumbel[:, :1].contiguous()
    return gumbel_topk(logits, 1, k, tau, g, seed, dense=True)["y_soft"][:, 0]


# ------------------------------------------------------------------------------------------ autograd wrappers
class _SolveEssential(torch.autograd.Function):
    """Five-point solve with implicit-function backward (dr_solve_nister5_bwd): at a returned model E the five
    epipolar constraints x2^T E x1 = 0 restricted to the tangent space of the essential manifold determine dE/dpts."""

    @staticmethod
    def forward(ctx, samples, weights, which):
        ctx.set_materialize_grads(False)   # unused / non-differentiable outputs arrive as None, not as zero-filled tensors
        fn = (lambda s_, w_: solve_nister5(s_, w_)) if which == "nister" else (lambda s_, w_: solve_stewenius5(s_))
        need_grad = samples.requires_grad and samples.dtype == torch.float32
        if need_grad and which == "nister" and samples.shape[-2] == 5:
            # train mode: the kernel computes in f64 anyway; it writes the f64 models next to the f32 ones and the backward
            # keeps them -- its tangent-space system is conditioned ~1e5, which f32-rounded models cannot afford
            models, m64, valid = solve_nister5_hp(samples, weights)
        elif need_grad:
            m64, valid = fn(samples.double(), None if weights is None else weights.double())
            models = m64.float()
        else:
            models, valid = fn(samples, weights)
            m64 = torch.empty(0, device=samples.device, dtype=torch.float64)
        ctx.save_for_backward(samples, models, m64, valid)
        ctx.minimal = samples.shape[-2] == 5
        ctx.weights = None if ctx.minimal or weights is None else weights.detach()
        ctx.mark_non_differentiable(valid)
        return models, valid

    @staticmethod
    def backward(ctx, g_models, _g_valid):
        if g_models is None:
            return None, None, None
        samples, models, m64, valid = ctx.saved_tensors
        f64 = samples.dtype == torch.float64
        if f64 and ctx.minimal:
            # `-pr 2 -tr 1`, round 5: f64 samples, models and gradients straight through (dr_solve_nister5_bwd_f64)
            s, Bt, n = _flat_samples(samples, 4)
            gs = torch.empty_like(s)
            L.call("dr_solve_nister5_bwd_f64", ptr(s), ptr(models.contiguous()), ptr(valid.contiguous().view(torch.uint8)),
                   ptr(g_models.to(torch.float64).contiguous()), c_int(Bt), ptr(gs), stream())
            return gs.reshape(samples.shape), None, None
        if f64:
            # n > 5 rows per sample in f64 (`-sam 3 -fmat 0 -tr 1 -pr 2`), round 6: f64 samples, weights, models and gradients straight
            # through (dr_solve_nister5_nm_bwd_f64; rounds 3-5 rounded samples and gradients to f32 on the way in)
            s, Bt, n = _flat_samples(samples, 4)
            gs = torch.empty_like(s)
            w = ctx.weights
            gw = None if w is None or not ctx.needs_input_grad[1] else torch.empty((Bt, n), device=s.device, dtype=torch.float64)
            L.call("dr_solve_nister5_nm_bwd_f64", ptr(s), ptr(None if w is None else w.reshape(Bt, n).double().contiguous()),
                   ptr(models.contiguous()), ptr(valid.contiguous().view(torch.uint8)), ptr(g_models.to(torch.float64).contiguous()),
                   c_int(Bt), c_int(n), ptr(gs), ptr(gw), stream())
            return gs.reshape(samples.shape), (None if gw is None else gw.reshape(w.shape).to(w.dtype)), None
        s, Bt, n = _flat_samples(samples, 4)
        gs = torch.empty_like(s)
        if not ctx.minimal:
            # n > 5 rows per sample (the reference's `-sam 3`: 8-point Gumbel sampler feeding the five-point estimator,
            # ransac.py:82-83): derivative of the invariant subspace the models live in, row weights included
            w = ctx.weights
            gw = None if w is None or not ctx.needs_input_grad[1] else torch.empty((Bt, n), device=s.device, dtype=torch.float32)
            L.call("dr_solve_nister5_nm_bwd_f32", ptr(s), ptr(None if w is None else w.reshape(Bt, n).float().contiguous()),
                   ptr(models.contiguous()), ptr(m64.contiguous() if m64.numel() else None),
                   ptr(valid.contiguous().view(torch.uint8)), ptr(g_models.contiguous()), c_int(Bt), c_int(n), ptr(gs), ptr(gw),
                   stream())
            gs = gs.reshape(samples.shape)
            return (gs.double() if f64 else gs), (None if gw is None else gw.reshape(w.shape).to(w.dtype)), None
        L.call("dr_solve_nister5_bwd_f32", ptr(s), ptr(models.contiguous()), ptr(m64.contiguous() if m64.numel() else None),
               ptr(valid.contiguous().view(torch.uint8)), ptr(g_models.contiguous()), c_int(Bt), ptr(gs), stream())
        gs = gs.reshape(samples.shape)
        return (gs.double() if f64 else gs), None, None


def solve_essential(samples, weights=None, which="nister"):
    """Differentiable five-point solve: samples [...,n,4] -> (models [...,10,3,3], valid [...,10]).
    Minimal samples (n = 5): implicit derivative of the five epipolar constraints.  Non-minimal samples (n > 5, Nister only:
    the reference's `-sam 3`, eight-point Gumbel sampler feeding the five-point estimator, ransac.py:82-83): implicit
    derivative of the invariant subspace the models live in (dr_solve_nister5_nm_bwd), gradients for the row weights too."""
    return _SolveEssential.apply(samples, weights, which)


class _SolveSelectEssential(torch.autograd.Function):
    """K3 (minimal five-point, train mode) + K5 (closest of the ten models to the ground truth) as ONE autograd node:
    samples [P,B,5,4] f32, gt [P,3,3] -> chosen [P,B,3,3], which [P,B], keep [P,B], models [P,B,10,3,3], valid [P,B,10].
    The backward hands the gradient of `chosen` to the solver's implicit-function backward in the sparse form
    (dr_solve_nister5_bwd_sel): no dense [P,B,10,9] gradient is written and scanned."""

    @staticmethod
    def forward(ctx, samples, weights, gt):
        ctx.set_materialize_grads(False)
        models, m64, valid = solve_nister5_hp(samples, weights)
        chosen, which, keep = select_closest(models, valid, gt, want_keep=True)
        ctx.save_for_backward(samples, models, m64, valid, which)
        ctx.mark_non_differentiable(which, keep, valid)
        return chosen, which, keep, models.detach(), valid

    @staticmethod
    def backward(ctx, g_chosen, _gw, _gk, g_models, _gv):
        if g_models is not None:
            raise L.DransacError("solve_select_essential: gradients flow through `chosen` only (use solve_essential + "
                                 "select_closest_autograd to differentiate the full model set)")
        if g_chosen is None:
            return None, None, None
        samples, models, m64, valid, which = ctx.saved_tensors
        s, Bt, _ = _flat_samples(samples, 4)
        gs = torch.empty_like(s)
        L.call("dr_solve_nister5_bwd_sel_f32", ptr(s), ptr(models.contiguous()), ptr(m64.contiguous()),
               ptr(valid.contiguous().view(torch.uint8)), ptr(g_chosen.contiguous()), ptr(which.contiguous()), c_int(Bt),
               ptr(gs), stream())
        return gs.reshape(samples.shape), None, None


def solve_select_essential(samples, gt, weights=None):
    """Train-mode K3 + K5 in one autograd node (Nister, minimal f32 samples): -> (chosen, which, keep, models, valid);
    `models` is returned detached (inspection / logging), the gradient path is samples -> chosen."""
    if samples.shape[-2] != 5 or samples.dtype != torch.float32:
        raise L.DransacError("solve_select_essential takes minimal f32 samples [...,5,4]")
    return _SolveSelectEssential.apply(samples, weights, gt)


class _SolveF8(torch.autograd.Function):
    @staticmethod
    def forward(ctx, samples, weights):
        ctx.set_materialize_grads(False)   # unused / non-differentiable outputs arrive as None, not as zero-filled tensors
        F, valid = solve_f8(samples, weights)
        ctx.save_for_backward(samples, weights if weights is not None else torch.empty(0, device=samples.device), F)
        ctx.has_w = weights is not None
        ctx.mark_non_differentiable(valid)
        return F, valid

    @staticmethod
    def backward(ctx, gF, _gv):
        if gF is None:
            return None, None
        samples, weights, F = ctx.saved_tensors
        dt = samples.dtype      # f32, or f64 (`-pr 2 -tr 1`): the same kernel with f64 in memory (round 5: nothing rounded to f32)
        s, Bt, n = _flat_samples(samples, 4)
        gs = torch.empty_like(s)
        w = weights.reshape(Bt, n).to(dt).contiguous() if ctx.has_w else None
        gw = torch.empty_like(w) if ctx.has_w else None
        L.call(f"dr_solve_f8_bwd_{L.suffix(dt)}", ptr(s), ptr(w), ptr(F.to(dt).contiguous()), ptr(gF.to(dt).contiguous()), c_int(Bt),
               c_int(n), ptr(gs), ptr(gw), stream())
        return gs.reshape(samples.shape), (gw.reshape(weights.shape) if ctx.has_w else None)


def solve_fundamental8(samples, weights=None):
    return _SolveF8.apply(samples, weights)


class _SolveRigid(torch.autograd.Function):
    @staticmethod
    def forward(ctx, samples, weights, flag):
        ctx.set_materialize_grads(False)   # unused / non-differentiable outputs arrive as None, not as zero-filled tensors
        out = solve_rigid(samples, weights, flag)
        ctx.save_for_backward(samples, out[0])
        ctx.flag = flag
        ctx.has_w = weights is not None
        ctx.mark_non_differentiable(out[4])
        return out

    @staticmethod
    def backward(ctx, g_model, g_R, g_t, g_scale, _gv):
        if g_model is None and g_R is None and g_t is None:
            return None, None, None
        samples, model = ctx.saved_tensors
        if samples.dtype != torch.float32:
            raise L.DransacError("backward is implemented for f32 only")
        if ctx.has_w:
            raise L.DransacError("backward of the weighted rigid solver is not implemented")
        g = g_model.clone() if g_model is not None else torch.zeros_like(model)
        if g_R is not None:
            g[..., :3, :3] += g_R
        if g_t is not None:
            g[..., :3, 3] += g_t
        s, Bt, n = _flat_samples(samples, 6)
        gs = torch.empty_like(s)
        L.call("dr_solve_rigid_bwd_f32", ptr(s), ptr(model.contiguous()), ptr(g.contiguous()), c_int(Bt), c_int(n),
               c_int(1 if ctx.flag else 0), ptr(gs), stream())
        return gs.reshape(samples.shape), None, None


def solve_rigid_autograd(samples, weights=None, flag=True):
    return _SolveRigid.apply(samples, weights, flag)


class _MsacScore(torch.autograd.Function):
    @staticmethod
    def forward(ctx, matches, models, thr, want_masks):
        ctx.set_materialize_grads(False)   # unused / non-differentiable outputs arrive as None, not as zero-filled tensors
        scores, masks = msac_score(matches, models, thr, want_masks)
        ctx.save_for_backward(matches, models, thr)
        if masks is not None:
            ctx.mark_non_differentiable(masks)
        return scores, masks

    @staticmethod
    def backward(ctx, g_scores, _gm):
        if g_scores is None:
            return None, None, None, None
        matches, models, thr = ctx.saved_tensors
        if matches.dtype != torch.float32:
            raise L.DransacError("backward is implemented for f32 only")
        P, N, _ = matches.shape
        M = models.shape[1]
        gm = torch.empty_like(models)
        L.call("dr_msac_score_bwd_f32", ptr(matches.contiguous()), ptr(models.contiguous()), ptr(thr),
               ptr(g_scores.contiguous()), c_int(P), c_int(M), c_int(N), ptr(gm), stream())
        return None, gm, None, None


def msac_score_autograd(matches, models, threshold, want_masks=True):
    thr = _thr_tensor(threshold, matches.shape[0], matches)
    return _MsacScore.apply(matches, models.reshape(models.shape[0], -1, 3, 3), thr, want_masks)


class _RigidResidual(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pts, models, threshold):
        ctx.set_materialize_grads(False)   # unused / non-differentiable outputs arrive as None, not as zero-filled tensors
        res, masks = rigid_residual(pts, models, threshold, True)
        ctx.save_for_backward(pts, models)
        ctx.mark_non_differentiable(masks)
        return res, masks

    @staticmethod
    def backward(ctx, g_res, _gm):
        if g_res is None:
            return None, None, None
        pts, models = ctx.saved_tensors
        if pts.dtype != torch.float32:
            raise L.DransacError("backward is implemented for f32 only")
        P, N, _ = pts.shape
        M = models.shape[1]
        gm = torch.empty_like(models)
        L.call("dr_rigid_residual_bwd_f32", ptr(pts.contiguous()), ptr(models.contiguous()), ptr(g_res.contiguous()),
               c_int(P), c_int(M), c_int(N), ptr(gm), stream())
        return None, gm, None


def rigid_residual_autograd(pts, models, threshold=0.03):
    return _RigidResidual.apply(pts, models, threshold)


class _SelectClosest(torch.autograd.Function):
    @staticmethod
    def forward(ctx, models, valid, gt):
        ctx.set_materialize_grads(False)   # unused / non-differentiable outputs arrive as None, not as zero-filled tensors
        chosen, which, keep = select_closest(models, valid, gt, want_keep=True)
        ctx.save_for_backward(which)
        ctx.shape = models.shape
        ctx.mark_non_differentiable(which, keep)
        return chosen, which, keep

    @staticmethod
    def backward(ctx, g_chosen, _gw, _gk):
        if g_chosen is None:
            return None, None, None
        (which,) = ctx.saved_tensors
        P, B, S = ctx.shape[:3]
        g = torch.empty(ctx.shape, device=g_chosen.device, dtype=g_chosen.dtype)
        L.call(f"dr_select_closest_bwd_{L.suffix(g_chosen.dtype)}", ptr(g_chosen.contiguous()), ptr(which), c_int(P),
               c_int(B), c_int(S), ptr(g), stream())
        return g, None, None


def select_closest_autograd(models, valid, gt, want_keep: bool = False):
    """-> (chosen, which) or, with want_keep, (chosen, which, keep [P,B] bool = which >= 0, written by the same launch)."""
    chosen, which, keep = _SelectClosest.apply(models, valid, gt)
    return (chosen, which, keep) if want_keep else (chosen, which)


# ------------------------------------------------------------------------------------------ MatchLoss residual (8(f) rank 2)
class _EpisymSums(torch.autograd.Function):
    @staticmethod
    def forward(ctx, matches, mask, models, valid):
        ctx.set_materialize_grads(False)   # unused / non-differentiable outputs arrive as None, not as zero-filled tensors
        P, N, _ = matches.shape
        M = models.shape[1]
        sums = torch.empty((P, M), device=matches.device, dtype=matches.dtype)   # every slot is written (invalid: 0)
        mk = None if mask is None else mask.contiguous().view(torch.uint8)
        v = None if valid is None else valid.contiguous().view(torch.uint8)
        L.call("dr_episym_fwd_f32", ptr(matches.contiguous()), ptr(mk), ptr(models.contiguous()), ptr(v), c_int(P), c_int(M),
               c_int(N), ptr(sums), stream())
        ctx.save_for_backward(matches, models)
        ctx.aux = (mk, v)
        return sums

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return None, None, None, None
        matches, models = ctx.saved_tensors
        mk, v = ctx.aux
        P, N, _ = matches.shape
        M = models.shape[1]
        gm = torch.empty_like(models)   # every slot is written (invalid: 0)
        L.call("dr_episym_bwd_f32", ptr(matches.contiguous()), ptr(mk), ptr(models.contiguous()), ptr(v),
               ptr(g.contiguous()), c_int(P), c_int(M), c_int(N), ptr(gm), stream())
        return None, None, gm, None


def episym_sums(matches, mask, models, valid=None):
    """sum over masked points of min(symmetric epipolar error, 1) for every model: matches [P,N,4], mask [P,N] bool |
    None, models [P,M,3,3], valid [P,M] bool | None -> [P,M] (differentiable w.r.t. models, f32)."""
    if matches.dtype != torch.float32:
        raise L.DransacError("episym_sums is implemented for f32")
    return _EpisymSums.apply(matches, mask, models.reshape(models.shape[0], -1, 3, 3), valid)


class _MatchLossPair(torch.autograd.Function):
    """episym forward + the per-pair reduction of MatchLoss in two launches; backward in one tiny torch op + one launch
    (the per-pair gradient goes straight into the episym backward: no [P,M] gradient tensor)."""

    @staticmethod
    def forward(ctx, matches, mask, models, keep):
        P, N, _ = matches.shape
        M = models.shape[1]
        matches, models = matches.contiguous(), models.contiguous()
        mk = None if mask is None else mask.contiguous().view(torch.uint8)
        v = None if keep is None else keep.contiguous().view(torch.uint8)
        sums = torch.empty((P, M), device=matches.device, dtype=matches.dtype)
        L.call("dr_episym_fwd_f32", ptr(matches), ptr(mk), ptr(models), ptr(v), c_int(P), c_int(M), c_int(N), ptr(sums),
               stream())
        per_pair = torch.empty((P,), device=matches.device, dtype=matches.dtype)
        coef = torch.empty((P,), device=matches.device, dtype=matches.dtype)
        L.call("dr_match_loss_pair_f32", ptr(sums), ptr(mk), ptr(v), c_int(P), c_int(M), c_int(N), ptr(per_pair), ptr(coef),
               stream())
        ctx.save_for_backward(matches, models, coef)
        ctx.aux = (mk, v)
        return per_pair

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return None, None, None, None
        matches, models, coef = ctx.saved_tensors
        mk, v = ctx.aux
        P, N, _ = matches.shape
        M = models.shape[1]
        gp = (g * coef).contiguous()
        gm = torch.empty_like(models)   # every slot is written (invalid: 0)
        L.call("dr_episym_bwd_pair_f32", ptr(matches), ptr(mk), ptr(models), ptr(v), ptr(gp), c_int(P), c_int(M), c_int(N),
               ptr(gm), stream())
        return None, None, gm, None


class _MatchLossMean(torch.autograd.Function):
    """MatchLoss down to the scalar: episym forward + dr_match_loss_mean (per-pair means AND their mean over the pairs) in two
    launches; backward = ONE launch (dr_episym_bwd_mean reads the upstream scalar gradient from device memory)."""

    @staticmethod
    def forward(ctx, matches, mask, models, keep):
        P, N, _ = matches.shape
        M = models.shape[1]
        matches, models = matches.contiguous(), models.contiguous()
        mk = None if mask is None else mask.contiguous().view(torch.uint8)
        v = None if keep is None else keep.contiguous().view(torch.uint8)
        sums = torch.empty((P, M), device=matches.device, dtype=matches.dtype)
        L.call("dr_episym_fwd_f32", ptr(matches), ptr(mk), ptr(models), ptr(v), c_int(P), c_int(M), c_int(N), ptr(sums),
               stream())
        per_pair = torch.empty((P,), device=matches.device, dtype=matches.dtype)
        coef = torch.empty((P,), device=matches.device, dtype=matches.dtype)
        mean = torch.empty((), device=matches.device, dtype=matches.dtype)
        L.call("dr_match_loss_mean_f32", ptr(sums), ptr(mk), ptr(v), c_int(P), c_int(M), c_int(N), ptr(per_pair), ptr(coef),
               ptr(mean), stream())
        ctx.save_for_backward(matches, models, coef)
        ctx.aux = (mk, v)
        return mean

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return None, None, None, None
        matches, models, coef = ctx.saved_tensors
        mk, v = ctx.aux
        P, N, _ = matches.shape
        M = models.shape[1]
        gm = torch.empty_like(models)   # every slot is written (invalid: 0)
        L.call("dr_episym_bwd_mean_f32", ptr(matches), ptr(mk), ptr(models), ptr(v), ptr(coef),
               ptr(g.to(matches.dtype).contiguous()), c_int(P), c_int(M), c_int(N), ptr(gm), stream())
        return None, None, gm, None


class _MatchLossFused(torch.autograd.Function):
    """MatchLoss value AND gradient in one pass (round 5: dr_match_loss_fused_f32): the forward writes, next to the loss, the
    unscaled gradient of every model; the backward is one elementwise launch (x coef[p] x upstream / P) -- no second walk over
    the (model x point) grid.  Taken in training (the models require grad)."""

    @staticmethod
    def forward(ctx, matches, mask, models, keep):
        P, N, _ = matches.shape
        M = models.shape[1]
        matches, models = matches.contiguous(), models.contiguous()
        mk = None if mask is None else mask.contiguous().view(torch.uint8)
        v = None if keep is None else keep.contiguous().view(torch.uint8)
        dev, dt = matches.device, matches.dtype
        sums = torch.empty((P, M), device=dev, dtype=dt)
        gun = torch.empty((P, M, 3, 3), device=dev, dtype=dt)
        per_pair = torch.empty((P,), device=dev, dtype=dt)
        coef = torch.empty((P,), device=dev, dtype=dt)
        mean = torch.empty((), device=dev, dtype=dt)
        L.call("dr_match_loss_fused_f32", ptr(matches), ptr(mk), ptr(models), ptr(v), c_int(P), c_int(M), c_int(N), ptr(sums),
               ptr(gun), ptr(per_pair), ptr(coef), ptr(mean), stream())
        ctx.save_for_backward(gun, coef)
        return mean

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return None, None, None, None
        gun, coef = ctx.saved_tensors
        P, M = gun.shape[0], gun.shape[1]
        gm = torch.empty_like(gun)
        L.call("dr_match_loss_scale_f32", ptr(gun), ptr(coef), ptr(g.to(gun.dtype).contiguous()), c_int(P), c_int(M), ptr(gm),
               stream())
        return None, None, gm, None


def match_loss_mean(matches, mask, models, keep=None):
    """MatchLoss of a batch: mean over pairs of the per-pair means (match_loss_per_pair(...).mean()) -> scalar, with the mean
    and its backward folded into the kernels (P <= 64; larger batches take the per-pair kernel + torch.mean).  When the models
    require grad (training), value and gradient come from ONE pass over the (model x point) grid (_MatchLossFused)."""
    models = models.reshape(models.shape[0], -1, 3, 3)
    if matches.dtype == torch.float64:
        return _match_loss_mean_f64(matches, mask, models, keep)
    if matches.dtype != torch.float32:
        raise L.DransacError("match_loss_mean is implemented for f32 and f64")
    if FUSED_MATCH_LOSS and torch.is_grad_enabled() and models.requires_grad:
        return _MatchLossFused.apply(matches, mask, models, keep)
    if matches.shape[0] > 64:
        return match_loss_per_pair(matches, mask, models, keep).mean()
    return _MatchLossMean.apply(matches, mask, models, keep)


FUSED_MATCH_LOSS = True   # tests / A-B runs: False = the two-pass form of rounds 3-4


def _match_loss_mean_f64(matches, mask, models, keep, chunk: int = 256):
    """`-pr 2 -tr 1 -w2 1` (model_cl.py:164-169, Q17): MatchLoss in double precision.  The episym kernels are packed-f32 code; the
    f64 parity path evaluates loss.py:137-153 / cv_utils.py:680-695 with torch ops ON THE DEVICE (autograd provides the backward),
    `chunk` models at a time so that the [P, chunk, N] temporaries stay small.  Not a hot path: the training path is f32."""
    P, N, _ = matches.shape
    M = models.shape[1]
    one = torch.ones_like(matches[..., :1])
    x1 = torch.cat((matches[..., :2], one), -1)
    x2 = torch.cat((matches[..., 2:], one), -1)
    w_pt = torch.ones((P, N), device=matches.device, dtype=matches.dtype) if mask is None else mask.to(matches.dtype)
    w_md = torch.ones((P, M), device=matches.device, dtype=matches.dtype) if keep is None else keep.to(matches.dtype)
    total = torch.zeros((P,), device=matches.device, dtype=matches.dtype)
    for m0 in range(0, M, chunk):
        F = models[:, m0:m0 + chunk]
        if keep is not None:
            # dropped slots are SELECTED out, not multiplied by 0: an invalid f8 / LSQ hypothesis may be NaN, and NaN * 0 would
            # turn the loss and every gradient into NaN where the f32 kernels skip the slot (round-5 advice)
            F = torch.where(keep[:, m0:m0 + chunk, None, None].bool(), F, torch.zeros((), device=F.device, dtype=F.dtype))
        Fx1 = torch.einsum("pmij,pnj->pmni", F, x1)
        Ftx2 = torch.einsum("pmji,pnj->pmni", F, x2)
        r = (x2[:, None] * Fx1).sum(-1)
        ys = r ** 2 * (1.0 / (Fx1[..., 0] ** 2 + Fx1[..., 1] ** 2 + 1e-15) + 1.0 / (Ftx2[..., 0] ** 2 + Ftx2[..., 1] ** 2 + 1e-15))
        ys = ys.clamp(max=1.0)
        if mask is not None:
            ys = torch.where(mask[:, None, :].bool(), ys, torch.zeros((), device=ys.device, dtype=ys.dtype))
        total = total + (ys * w_md[:, m0:m0 + chunk, None]).sum((1, 2))
    den = (w_pt.sum(1) * w_md.sum(1)).clamp(min=1.0)
    return (total / den).mean()


def match_loss_per_pair(matches, mask, models, keep=None):
    """MatchLoss per pair: mean over the kept models and the masked points of min(symmetric epipolar error, 1).
    matches [P,N,4] f32, mask [P,N] bool | None, models [P,M,3,3], keep [P,M] bool | None -> [P] (differentiable w.r.t.
    models)."""
    if matches.dtype != torch.float32:
        raise L.DransacError("match_loss_per_pair is implemented for f32")
    return _MatchLossPair.apply(matches, mask, models.reshape(models.shape[0], -1, 3, 3), keep)


# ------------------------------------------------------------------------------------------ PoseLoss pose error (8(f) rank 3)
class _PoseError(torch.autograd.Function):
    @staticmethod
    def forward(ctx, matches, models, gt_R, gt_t, distance_threshold, want_votes):
        ctx.set_materialize_grads(False)   # unused / non-differentiable outputs arrive as None, not as zero-filled tensors
        P, N, _ = matches.shape
        M = models.shape[1]
        dev, dt = matches.device, matches.dtype
        sfx = L.suffix(dt)
        matches, models = matches.contiguous(), models.contiguous()
        gt_R, gt_t = gt_R.to(dt).contiguous(), gt_t.to(dt).contiguous()
        err_R = torch.empty((P, M), device=dev, dtype=dt)
        err_t = torch.empty((P, M), device=dev, dtype=dt)
        which = torch.empty((P, M), device=dev, dtype=torch.int32)
        votes = torch.empty((P, M, 4), device=dev, dtype=torch.int32) if want_votes else None
        L.call(f"dr_pose_error_fwd_{sfx}", ptr(matches), ptr(models), ptr(gt_R), ptr(gt_t), c_int(P), c_int(M), c_int(N),
               L.c_double(float(distance_threshold)), ptr(err_R), ptr(err_t), ptr(which), ptr(votes), stream())
        ctx.save_for_backward(models, gt_R, gt_t, which)
        ctx.mark_non_differentiable(which)
        if votes is not None:
            ctx.mark_non_differentiable(votes)
        return err_R, err_t, which, votes

    @staticmethod
    def backward(ctx, g_R, g_t, _gw, _gv):
        if g_R is None and g_t is None:
            return (None,) * 6
        models, gt_R, gt_t, which = ctx.saved_tensors
        P, M = which.shape
        gm = torch.empty_like(models)
        zero = None
        if g_R is None or g_t is None:
            zero = torch.zeros((P, M), device=models.device, dtype=models.dtype)
        g_R = zero if g_R is None else g_R.contiguous()
        g_t = zero if g_t is None else g_t.contiguous()
        L.call(f"dr_pose_error_bwd_{L.suffix(models.dtype)}", ptr(models), ptr(gt_R), ptr(gt_t), ptr(which), ptr(g_R),
               ptr(g_t), c_int(P), c_int(M), ptr(gm), stream())
        return None, gm, None, None, None, None


def pose_error(matches, models, gt_R, gt_t, distance_threshold: float = 50.0, want_votes: bool = False, svd: bool = False):
    """eval_essential_matrix (cv_utils.py:503-525) for all models of all pairs: matches [P,N,4] normalised,
    models [P,M,3,3], gt_R [P,3,3], gt_t [P,3] -> err_R, err_t [P,M] in degrees, the chosen candidate [P,M]
    (0..3 = (R1,t) (R2,t) (R1,-t) (R2,-t)) and, optionally, the four cheirality votes.
    svd=False: Horn decomposition (what train.py passes), differentiable w.r.t. the models.  svd=True: the SVD
    decomposition of decompose_E (cv_utils.py:83-116), forward only -- the reference's gradient goes through
    torch.linalg.svd of a matrix with two equal singular values, where it does not exist."""
    P = matches.shape[0]
    models = models.reshape(P, -1, 3, 3)
    if not svd:
        return _PoseError.apply(matches, models, gt_R.reshape(P, 9), gt_t.reshape(P, 3), distance_threshold, want_votes)
    if models.requires_grad and torch.is_grad_enabled():
        raise L.DransacError("pose_error(svd=True) is forward only: the gradient through the SVD of an essential matrix "
                             "(sigma_1 = sigma_2) is undefined; use svd=False (Horn) for training, as train.py does")
    N, M = matches.shape[1], models.shape[1]
    dev, dt = matches.device, matches.dtype
    err_R = torch.empty((P, M), device=dev, dtype=dt)
    err_t = torch.empty((P, M), device=dev, dtype=dt)
    which = torch.empty((P, M), device=dev, dtype=torch.int32)
    votes = torch.empty((P, M, 4), device=dev, dtype=torch.int32) if want_votes else None
    # keep every converted tensor alive until the launch is enqueued: a temporary passed straight to ptr() is freed at
    # once and the caching allocator hands its block to the next temporary
    mt_, md_ = matches.contiguous(), models.detach().contiguous()
    gR_, gt_ = gt_R.reshape(P, 9).to(dt).contiguous(), gt_t.reshape(P, 3).to(dt).contiguous()
    L.call(f"dr_pose_error_svd_fwd_{L.suffix(dt)}", ptr(mt_), ptr(md_), ptr(gR_), ptr(gt_), c_int(P), c_int(M), c_int(N),
           L.c_double(float(distance_threshold)), ptr(err_R), ptr(err_t), ptr(which), ptr(votes), stream())
    return err_R, err_t, which, votes


def recover_pose_mask(matches, models, distance_threshold: float = 50.0):
    """The inlier mask of cv2.recoverPose(E, pts1, pts2) (what loss.py:99,134 uses as ground-truth inliers): matches
    [P,N,4] normalised, models [P,M,3,3] or [P,3,3] -> (mask [P,M,N] bool, which [P,M] int32 = winning candidate)."""
    P, N, _ = matches.shape
    models = models.reshape(P, -1, 3, 3).to(matches.dtype).contiguous()
    M = models.shape[1]
    mask = torch.empty((P, M, N), device=matches.device, dtype=torch.bool)
    which = torch.empty((P, M), device=matches.device, dtype=torch.int32)
    L.call(f"dr_recover_pose_mask_{L.suffix(matches.dtype)}", ptr(matches.contiguous()), ptr(models), c_int(P), c_int(M),
           c_int(N), L.c_double(float(distance_threshold)), ptr(which), ptr(mask), stream())
    return mask, which
