// K1 -- Gumbel-softmax top-k straight-through sampler (reference: GumbelSoftmaxSampler.sample,
// samplers/gumbel_sampler.py:25-42), K1u uniform sampler (uniform_sampler.py:15-19), K2 gather
// (ransac.py:58-65) and the backward of K1+K2.
//
// The reference materialises five [B,N] tensors per call (repeat, noise, softmax, one-hot, ret).
// Here one wave owns one (pair, hypothesis) row and never writes a [B,N] tensor unless the caller
// asks for the API-faithful dense outputs:
//   pass A  g = (logit + gumbel)/tau for the lane's 4-element groups (16-byte coalesced loads, or
//           Philox4x32-7 in-kernel: one call = 4 elements), online soft-max (running max / sum)
//           and the lane maximum; g is parked in LDS (4N bytes per wave) when it fits;
//   select  T = k-th largest of the 64 lane maxima (k wave-max rounds) is a lower bound of the k-th
//           largest element, so { g >= T } is a small superset of the top-k: it is compacted with
//           ballots into <= 64 candidates, ranked by counting, and the winners are emitted in
//           ASCENDING point index (the order `points[samples != 0]` produces, ransac.py:65);
//           pathological rows (> 64 candidates: massive ties) take a k-round arg-max slow path.
// HBM traffic in fused mode: 4N bytes of logits (L2-resident across the B rows of a pair) in,
// B(8k+4) bytes out.
#include <algorithm>

#include "dr_common.hpp"

namespace dr {

#ifndef DR_K1_FAST
#define DR_K1_FAST 1   // 0: always the general kernel (A/B builds)
#endif
#ifndef DR_K1_PASSB_LDS
#define DR_K1_PASSB_LDS 0   // 1: register kernel: the (few) lanes whose maximum reaches the threshold park their 32 values in LDS and
#endif                      // the whole wave scans those, instead of a ballot per element of every group (see pass B there).
                            // Measured in the step: 1.0219 vs 1.0195 ms (slower: the compare + branch per element it
                            // replaces mostly falls through) -- off.
#ifndef DR_K1_PASSB_ATOMIC
#define DR_K1_PASSB_ATOMIC 0   // 1: register kernel collects the candidates in the lanes that own them (LDS counter) instead of by
                               // wave-wide ballots.  In the step (scratch/r3_gpu_s.sh): 1.035 / 1.036 ms against 1.026 / 1.032 ms
                               // with the ballots -- the per-element divergent branches cost more than the ballots they replace: off
#endif
constexpr int kRowsPerBlock = 4;   // one wave per row
constexpr int kMaxK = 8;
constexpr int kMaxCand = 64;

template <typename T>
struct GumbelArgs {
  const T *logits;   // [P,N] or null (ones)
  const T *gumbel;   // [P,B,N] or null (Philox)
  uint64_t seed;
  T tau;
  int P, B, N, k;
  const uint64_t *seed_ptr;   // optional: the seed lives in device memory (captured graphs: one word updated per replay)
  // sub > 0 (round 6, super-rounds of the test-mode drivers): the B rows are ceil(B / sub) consecutive SUB-BATCHES of `sub` rows, the
  // batches a batch-by-batch loop would have drawn one call after the other: row b draws what row b % sub of the call with the seed
  // `seed + b / sub` draws (the drivers' per-call seeds are consecutive integers)
  int sub;
};
// Philox key and row counter of row b (see GumbelArgs::sub); wave-uniform
__device__ __forceinline__ void sub_batch_row(uint64_t &seed, int &bq, int b, int sub) {
  bq = b;
  if (sub > 0) {
    seed += (uint64_t)(b / sub);
    bq = b % sub;
  }
}

template <typename T> __device__ __forceinline__ T gumbel_from_bits_t(uint32_t bits);
template <> __device__ __forceinline__ float gumbel_from_bits_t<float>(uint32_t bits) { return gumbel_from_bits(bits); }
template <> __device__ __forceinline__ double gumbel_from_bits_t<double>(uint32_t bits) {
  double r = (double)bits * 2.3283064365386963e-10;  // 2^-32, [0,1)
  double u = r * (1.0 - 2.220446049250313e-16 - 2.2250738585072014e-308) + 2.2250738585072014e-308;
  return -log(-log(u));
}

// g values of the 4-element group q (elements 4q..4q+3) of row (p,b)
template <typename T>
__device__ __forceinline__ void load_group(const GumbelArgs<T> &a, int p, int b, int q, T g[4], T noise[4]) {
  const int n0 = 4 * q;
  const bool full = (n0 + 3 < a.N) && ((a.N & 3) == 0);   // 16-byte aligned, fully inside the row
  if (a.gumbel) {
    const T *src = a.gumbel + ((size_t)p * a.B + b) * a.N + n0;
    if (full && sizeof(T) == 4) {
      const float4 v = *reinterpret_cast<const float4 *>(src);
      noise[0] = v.x; noise[1] = v.y; noise[2] = v.z; noise[3] = v.w;
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) noise[j] = (n0 + j < a.N) ? src[j] : T(0);
    }
  } else {
    uint32_t r[4];
    Philox::gen(a.seed, (uint32_t)q, (uint32_t)(a.sub > 0 ? b % a.sub : b), (uint32_t)p, 0u, r);   // (a.seed: advanced by b / sub at kernel entry)
#pragma unroll
    for (int j = 0; j < 4; ++j) noise[j] = gumbel_from_bits_t<T>(r[j]);
  }
  T l[4];
  if (!a.logits) {
#pragma unroll
    for (int j = 0; j < 4; ++j) l[j] = T(1);
  } else if (full && sizeof(T) == 4) {
    const float4 v = *reinterpret_cast<const float4 *>(a.logits + (size_t)p * a.N + n0);
    l[0] = v.x; l[1] = v.y; l[2] = v.z; l[3] = v.w;
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) l[j] = (n0 + j < a.N) ? a.logits[(size_t)p * a.N + n0 + j] : T(0);
  }
  if (a.tau == T(1)) {   // wave-uniform: x/1 == x exactly, skip the IEEE division sequence
#pragma unroll
    for (int j = 0; j < 4; ++j) g[j] = (n0 + j < a.N) ? (l[j] + noise[j]) : -INFINITY;
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) g[j] = (n0 + j < a.N) ? (l[j] + noise[j]) / a.tau : -INFINITY;
  }
}

template <typename T> __device__ __forceinline__ T exp_t(T x);
template <> __device__ __forceinline__ float exp_t<float>(float x) {
  return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f);   // arguments are <= 0 and flush to 0 for x < -87: fine for soft-max terms
}
template <> __device__ __forceinline__ double exp_t<double>(double x) { return exp(x); }
template <typename T> __device__ __forceinline__ T log_t(T x);
template <> __device__ __forceinline__ float log_t<float>(float x) { return logf(x); }
template <> __device__ __forceinline__ double log_t<double>(double x) { return log(x); }

__device__ __forceinline__ float row_max(float v) { return wave_max_bcast(v); }
__device__ __forceinline__ double row_max(double v) { return wave_max(v); }
__device__ __forceinline__ float row_sum(float v) { return wave_sum_bcast(v); }
__device__ __forceinline__ double row_sum(double v) { return wave_sum(v); }

// kSoft = false: index sets only (y_sel, lse and the dense outputs are all NULL) -- RANSAC test mode consumes nothing
// but `samples != 0` (ransac.py:65), so the soft-max statistics of the row (one exp per element) are skipped.
template <typename T, bool kCache, bool kSoft>
__global__ __launch_bounds__(kRowsPerBlock * 64) void gumbel_topk_kernel(GumbelArgs<T> a, int32_t *__restrict__ idx,
                                                                        T *__restrict__ y_sel, T *__restrict__ lse_out,
                                                                        T *__restrict__ y_soft, T *__restrict__ ret,
                                                                        T *__restrict__ gumbel_out) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  __shared__ int s_win[kRowsPerBlock][kMaxK];
  if (a.seed_ptr) a.seed = *a.seed_ptr;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int p = blockIdx.y, b = blockIdx.x * kRowsPerBlock + wv;
  const int groups = (a.N + 3) >> 2;
  // LDS carve: per wave [cand_val 64][cand_idx 64] then (kCache) g cache [4*groups]
  T *cand_val = reinterpret_cast<T *>(smem_raw) + (size_t)wv * kMaxCand;
  int *cand_idx = reinterpret_cast<int *>(reinterpret_cast<T *>(smem_raw) + (size_t)kRowsPerBlock * kMaxCand) + wv * kMaxCand;
  T *gcache = reinterpret_cast<T *>(reinterpret_cast<int *>(reinterpret_cast<T *>(smem_raw) + (size_t)kRowsPerBlock * kMaxCand) +
                                    kRowsPerBlock * kMaxCand) + (size_t)wv * groups * 4;
  if (b >= a.B) return;  // whole wave exits together (no block-level barrier is used below)
  if (a.sub > 0) a.seed += (uint64_t)(b / a.sub);
  const size_t row = ((size_t)p * a.B + b);

  // ---------------- pass A: online soft-max + lane maximum
  T mx = -INFINITY, sm = T(0), lmax = -INFINITY;
  for (int q = lane; q < groups; q += 64) {
    T g[4], nz[4];
    load_group<T>(a, p, b, q, g, nz);
    if (kCache) {
#pragma unroll
      for (int j = 0; j < 4; ++j) gcache[4 * q + j] = g[j];
    }
    if (gumbel_out) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (4 * q + j < a.N) gumbel_out[row * a.N + 4 * q + j] = nz[j];
    }
    T gm = fmax(fmax(g[0], g[1]), fmax(g[2], g[3]));
    lmax = fmax(lmax, gm);
    if (kSoft) {
      if (gm > mx) { sm *= exp_t<T>(mx - gm); mx = gm; }
#pragma unroll
      for (int j = 0; j < 4; ++j) sm += exp_t<T>(g[j] - mx);
    }
  }
  // wave-wide soft-max statistics
  T wmx = T(0), lse = T(0), inv_sm = T(0);
  if (kSoft) {
    wmx = row_max(mx);
    sm *= (mx == -INFINITY) ? T(0) : exp_t<T>(mx - wmx);
    sm = row_sum(sm);
    lse = wmx + log_t<T>(sm);
    inv_sm = T(1) / sm;   // y = exp(g - max) / sum: exact to rounding even when |g| is huge (lse alone is not)
  }

  // ---------------- threshold: k-th largest lane maximum
  T v = lmax, thr = -INFINITY;
  for (int r = 0; r < a.k; ++r) {
    thr = row_max(v);
    unsigned long long who = __ballot(v == thr);
    if (lane == __ffsll((long long)who) - 1) v = -INFINITY;
  }

  // ---------------- pass B: compact candidates { g >= thr }
  int ncand = 0;
  for (int q0 = 0; q0 < groups; q0 += 64) {
    const int q = q0 + lane;
    T g[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    if (q < groups) {
      if (kCache) {
#pragma unroll
        for (int j = 0; j < 4; ++j) g[j] = gcache[4 * q + j];
      } else {
        T nz[4];
        load_group<T>(a, p, b, q, g, nz);
      }
    }
    // most iterations hold no candidate at all: one ballot on the group maximum instead of four on the elements
    if (!__ballot(fmax(fmax(g[0], g[1]), fmax(g[2], g[3])) >= thr)) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool c = g[j] >= thr;
      unsigned long long bal = __ballot(c);
      if (bal) {
        const int pos = ncand + __popcll(bal & ((1ull << lane) - 1ull));
        if (c && pos < kMaxCand) { cand_val[pos] = g[j]; cand_idx[pos] = 4 * q + j; }
        ncand += __popcll(bal);
      }
    }
  }

  // the candidate list is read across lanes below: order the LDS writes before the reads
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();

  int *winp = &s_win[wv][0];   // winners' point indices, ascending
  if (ncand <= kMaxCand) {
    const bool have = lane < ncand;
    const T cv = have ? cand_val[lane] : -INFINITY;
    const int ci = have ? cand_idx[lane] : 0x7fffffff;
    int rank = 0;
    for (int j = 0; j < ncand; ++j) {
      const T ov = cand_val[j];
      const int oi = cand_idx[j];
      rank += (ov > cv) || (ov == cv && oi < ci);
    }
    const bool win = have && rank < a.k;
    // position among winners by ascending index
    unsigned long long wb = __ballot(win);
    int pos = 0;
    for (int j = 0; j < ncand; ++j) {
      if ((wb >> j) & 1ull) pos += cand_idx[j] < ci;
    }
    if (win) {
      idx[row * a.k + pos] = ci;
      if (kSoft) y_sel[row * a.k + pos] = exp_t<T>(cv - wmx) * inv_sm;
      s_win[wv][pos] = ci;
    }
  } else {
    // slow path: k rounds of (value desc, index asc) arg-max with exclusion of earlier winners
    int won[kMaxK];
    T wong[kMaxK];
    for (int r = 0; r < a.k; ++r) {
      T bv = -INFINITY;
      int bi = 0x7fffffff;
      for (int q = lane; q < groups; q += 64) {
        T g[4], nz[4];
        load_group<T>(a, p, b, q, g, nz);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int n = 4 * q + j;
          bool skip = n >= a.N;
          for (int s = 0; s < r; ++s) skip = skip || (won[s] == n);
          if (!skip && (g[j] > bv || (g[j] == bv && n < bi))) { bv = g[j]; bi = n; }
        }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        T ov = __shfl_xor(bv, o, 64);
        int oi = __shfl_xor(bi, o, 64);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
      }
      won[r] = bi;
      wong[r] = bv;
    }
    if (lane < a.k) {
      int me = 0;
      T mg = T(0);
      for (int r = 0; r < a.k; ++r) if (r == lane) { me = won[r]; mg = wong[r]; }
      int pos = 0;
      for (int r = 0; r < a.k; ++r) pos += won[r] < me;
      idx[row * a.k + pos] = me;
      if (kSoft) y_sel[row * a.k + pos] = exp_t<T>(mg - wmx) * inv_sm;
      s_win[wv][pos] = me;
    }
  }
  if (kSoft && lane == 0) lse_out[row] = lse;

  // ---------------- optional dense outputs (API-faithful mode)
  if (kSoft && (y_soft || ret)) {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    int mine[kMaxK];
    for (int r = 0; r < kMaxK; ++r) mine[r] = r < a.k ? winp[r] : -1;
    for (int q = lane; q < groups; q += 64) {
      T g[4];
      if (kCache) {
#pragma unroll
        for (int j = 0; j < 4; ++j) g[j] = gcache[4 * q + j];
      } else {
        T nz[4];
        load_group<T>(a, p, b, q, g, nz);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = 4 * q + j;
        if (n >= a.N) continue;
        const T y = exp_t<T>(g[j] - wmx) * inv_sm;
        bool sel = false;
#pragma unroll
        for (int r = 0; r < kMaxK; ++r) sel = sel || (mine[r] == n);
        if (y_soft) y_soft[row * a.N + n] = y;
        if (ret) ret[row * a.N + n] = ((sel ? T(1) : T(0)) - y) + y;   // gumbel_sampler.py:38
      }
    }
  }
}

// ---- benchmark-shape specialisation of the kernel above ----------------------------------------------------------------
// log2 of the uniform that gumbel_from_bits (dr_common.hpp) draws from the same bits -- the same convert + fma + v_log_f32
__device__ __forceinline__ float log2_uniform_from_bits(uint32_t bits) {
  constexpr float kScale = 2.3283064365386963e-10f * (1.0f - 1.1920928955078125e-07f - 1.17549435e-38f);   // 2^-32 * c
  return __builtin_amdgcn_logf(__builtin_fmaf((float)bits, kScale, 1.17549435e-38f));
}

// ---- exponential-race form of the index-only sampler (round 6) -------------------------------------------------------------
// The k largest of logit_n + G_n, G_n = -ln(-ln u_n), are the k SMALLEST of (-ln u_n) exp(-logit_n): with w_n = exp(lmax - logit_n)
// per pair (N values, this kernel), a row ranks key_n = w_n log2 u_n (<= 0, larger is better) -- one logarithm and one multiply per
// element instead of two logarithms and an add (-ln(-key_n) = logit_n + G_n - lmax + ln ln 2: the same order, up to the rounding of
// near-ties).  A pair whose logits are not all finite or span more than 80 (w would overflow for points that can still win when
// fewer than k others are in range) is flagged and keeps the two-logarithm form.  ws: w [P,N] floats, then one flag word per pair.
__global__ __launch_bounds__(256) void gumbel_race_weights_kernel(const float *__restrict__ logits, int N, int P, float *__restrict__ ws) {
  race_weights_block(logits, N, P, blockIdx.x, ws);   // (dr_common.hpp: dr_ransac_init runs the same code when it is handed the logits)
}

// f32, in-kernel Philox, logits given, tau = 1, N % 4 == 0 and N <= 2048, no dense outputs: what every RANSAC round of the
// drivers asks for.  Same algorithm, same Philox counters, same comparison order -- the index sets are bit-identical to
// the general kernel's, y_sel / lse agree to rounding (tests/test_gpu_round2.py) -- but a lane keeps its eight 4-element
// groups of g in 32 REGISTERS instead of an LDS row cache, and none of the per-iteration mode tests of load_group exist.
// Measured at 32 x 1024 x 2000 (scratch/ab_k1.py): general kernel 77.7 us in test mode (Philox 29.5, the two logarithms
// 7.7, everything else 42-48), this kernel 60.0 us (everything else: 29); train mode 88.1 -> 74.4 us.
#ifndef DR_K1_SALU_SELECT
#define DR_K1_SALU_SELECT 1   // the selection on wave compare masks (0 = the LDS list of rounds 2-5: A/B); 2 = the one-logarithm form only
#endif
#ifndef DR_K1_WV_SGPR
#define DR_K1_WV_SGPR 1     // the wave index through v_readfirstlane (row, seed, Philox round keys in SGPRs); 2 = index-only mode only
#endif
#ifndef DR_K1_DBG_SELECT
#define DR_K1_DBG_SELECT 0
#endif
constexpr int kFastGroups = 8;   // groups per lane: 64 lanes x 8 groups x 4 elements = 2048

// train mode, K2 fused: the correspondence times the straight-through weight (1 - y) + y (gumbel_sampler.py:40: y_hard -
// y_soft.detach() + y_soft at a selected point), the arithmetic of gather_fwd_kernel
__device__ __forceinline__ float4 straight_through(float4 m, float y) {
  const float st = (1.0f - y) + y;
  return make_float4(m.x * st, m.y * st, m.z * st, m.w * st);
}

// kScreen (round 5, index-only mode): the screening words of the long-row kernel for SHORT rows.  A point can only be among the k
// winners if its score reaches T = logsumexp(logits) - ln(lambda), i.e. if its Philox word reaches a per-point threshold that
// depends on the pair's logits only (gumbel_screen_short_kernel: screen_tb [P,N] words, screen_T [P]).  The wave compares the 64
// words of an element slot against their thresholds -- no logarithm -- and only the lanes that pass (~lambda = 11 + k elements of
// the row) evaluate their score, exactly as the unscreened kernel would have (same word, same logit, same rounding): the winners
// are ranked among them by (value, index).  A row with fewer than k evaluated scores >= T (4e-4 of the rows at lambda = 16), or
// more candidates than the list holds, takes the unscreened path below.
template <bool kSoft, bool kScreen = false>
__global__ __launch_bounds__(kRowsPerBlock * 64) void gumbel_topk_fast_kernel(const float *__restrict__ logits, uint64_t seed,
                                                                             int B, int N, int k, int32_t *__restrict__ idx,
                                                                             float *__restrict__ y_sel,
                                                                             float *__restrict__ lse_out,
                                                                             const uint64_t *__restrict__ seed_ptr,
                                                                             const float4 *__restrict__ gather_src = nullptr,
                                                                             float4 *__restrict__ gather_dst = nullptr,
                                                                             PairGate gate = PairGate(),
                                                                             const uint32_t *__restrict__ screen_tb = nullptr,
                                                                             const float *__restrict__ screen_T = nullptr,
                                                                             int sub = 0, const float *__restrict__ race_ws = nullptr,
                                                                             int P_race = 0) {
  // race_ws (index-only mode, unscreened): the exponential-race form (gumbel_race_weights_kernel), one logarithm per element
  // gather_src / gather_dst (index-only mode): K2 fused -- the winners' correspondences [P,N] x float4 -> samples [P,B,k] x float4
  if (gate.closed(blockIdx.y)) return;   // this pair has terminated (block-uniform): its rows keep what the last round drew
  static_assert(!(kSoft && kScreen), "the soft-max statistics need every element's score");
  __shared__ float s_val[kRowsPerBlock][kMaxCand];
  __shared__ int s_idx[kRowsPerBlock][kMaxCand];
#if DR_K1_PASSB_ATOMIC
  __shared__ int s_cnt[kRowsPerBlock];
#endif
#if DR_K1_PASSB_LDS
  constexpr int kHotMax = 8, kHotStride = 4 * kFastGroups + 4;   // stride 36 words: rows start in different banks, 16 B aligned
  __shared__ __align__(16) float s_stage[kRowsPerBlock][kHotMax * kHotStride];
  __shared__ int s_hot[kRowsPerBlock][kHotMax];
#endif
  if (seed_ptr) seed = *seed_ptr;
  // (the wave index through v_readfirstlane: the row, its seed and the twelve Philox round keys then live in SGPRs)
  const int lane = threadIdx.x & 63;
  int wv = threadIdx.x >> 6;
  if (DR_K1_WV_SGPR == 1 || (DR_K1_WV_SGPR == 2 && !kSoft)) wv = __builtin_amdgcn_readfirstlane(wv);
  const int p = blockIdx.y, b = blockIdx.x * kRowsPerBlock + wv;
  if (b >= B) return;   // whole wave exits together (no block-level barrier is used below)
  int bq;
  sub_batch_row(seed, bq, b, sub);
  const int groups = N >> 2;
  const size_t row = (size_t)p * B + b;
  const float4 *lg = reinterpret_cast<const float4 *>(logits + (size_t)p * N);
  float *cand_val = s_val[wv];
  int *cand_idx = s_idx[wv];

  if constexpr (kScreen) {
    const float Tf = screen_T[p];
    if (Tf != INFINITY) {   // (non-finite logits, or a threshold beyond the margin's reach: the pair is not screened)
      // One element SLOT (group i, component j) at a time: the wave compares the slot's 64 words with their thresholds (one
      // v_cmp; the result is a wave mask in SGPRs) and, only if some lane passed (~40 % of the slots at lambda = 16), those lanes
      // transform THEIR word -- a register with a static index -- add the logit they hold and append the score to the wave's
      // list.  No word is parked anywhere, no load depends on a hit (first version, `s_words`: 32 KiB of LDS per block and a
      // global logit load per evaluation round -- slower than the unscreened kernel although it issued 40 % less).
      const uint4 *tb4 = reinterpret_cast<const uint4 *>(screen_tb + (size_t)p * N);
      int ncand = 0, reach = 0;
      uint4 t_nx = lane < groups ? tb4[lane] : make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu);
      float4 l_nx = lane < groups ? lg[lane] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int i = 0; i < kFastGroups; ++i) {
        if (64 * i >= groups) break;   // wave-uniform
        const int q = lane + 64 * i;
        const bool inb = q < groups;
        const uint4 t4 = t_nx;
        const float4 l4v = l_nx;
        if (i + 1 < kFastGroups) {     // the next group's thresholds and logits are requested before this group's Philox rounds
          const int qn = q + 64;
          t_nx = qn < groups ? tb4[qn] : make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu);
          l_nx = qn < groups ? lg[qn] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        uint32_t r[4];
        Philox::gen(seed, (uint32_t)q, (uint32_t)bq, (uint32_t)p, 0u, r);
        const uint32_t tt[4] = {t4.x, t4.y, t4.z, t4.w};
        const float ll[4] = {l4v.x, l4v.y, l4v.z, l4v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const bool has = inb && r[j] >= tt[j];
          const unsigned long long bal = __builtin_amdgcn_ballot_w64(has);   // (the compare's own mask: __ballot adds a select + compare)
          if (bal) {
            float gv = -INFINITY;
            if (has) {
              gv = ll[j] + gumbel_from_bits(r[j]);
              const int pos = ncand + __popcll(bal & ((1ull << lane) - 1ull));
              if (pos < kMaxCand) { cand_val[pos] = gv; cand_idx[pos] = 4 * q + j; }
            }
            ncand += __popcll(bal);
            reach += __popcll(__builtin_amdgcn_ballot_w64(has && gv >= Tf));
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
      __builtin_amdgcn_wave_barrier();
      if (reach >= k && ncand <= kMaxCand) {
        // the k winners are all >= T, hence all in the list: rank by (value descending, index ascending)
        const bool have = lane < ncand;
        const float cv = have ? cand_val[lane] : -INFINITY;
        const int ci = have ? cand_idx[lane] : 0x7fffffff;
        int rank = 0;
        for (int j = 0; j < ncand; ++j) {
          const float ov = cand_val[j];
          const int oi = cand_idx[j];
          rank += (ov > cv) || (ov == cv && oi < ci);
        }
        const bool win = have && rank < k;
        const unsigned long long wb = __ballot(win);
        int pos = 0;
        for (int j = 0; j < ncand; ++j) {
          if ((wb >> j) & 1ull) pos += cand_idx[j] < ci;
        }
        if (win) {
          idx[row * k + pos] = ci;
          if (gather_dst) gather_dst[row * k + pos] = gather_src[(size_t)p * N + ci];
        }
        return;
      }
      __builtin_amdgcn_wave_barrier();   // too few scores reach T (or too many candidates): the unscreened path
    }
  }

  // ---------------- pass A: g into registers, online soft-max, lane maximum
  float g[kFastGroups][4];
  float mx = -INFINITY, sm = 0.f, lmax = -INFINITY;
  bool race = false;
  if constexpr (!kScreen) race = race_ws != nullptr && reinterpret_cast<const int *>(race_ws + (size_t)P_race * N)[p] != 0;
  if (race) {   // wave-uniform (per pair)
    // kSoft (train mode): exp(g_n - lmax) = exp(logit_n - lmax) / (-ln u_n) = 1 / (ln 2 (-key_n)) -- the soft-max statistics of the
    // row from the SAME keys: e_n = 1 / (-key_n), y_n = e_n / sum e, lse = lmax + ln(sum e) - ln ln 2.  A reciprocal and an add per
    // element instead of the second logarithm, the add, the subtract, the exponential and the running maximum of the form below.
    const float4 *wr = reinterpret_cast<const float4 *>(race_ws + (size_t)p * N);
#pragma unroll
    for (int i = 0; i < kFastGroups; ++i) {
      const int q = lane + 64 * i;
      if (q < groups) {
        uint32_t r[4];
        Philox::gen(seed, (uint32_t)q, (uint32_t)bq, (uint32_t)p, 0u, r);
        const float4 w4 = wr[q];
        g[i][0] = w4.x * log2_uniform_from_bits(r[0]);
        g[i][1] = w4.y * log2_uniform_from_bits(r[1]);
        g[i][2] = w4.z * log2_uniform_from_bits(r[2]);
        g[i][3] = w4.w * log2_uniform_from_bits(r[3]);
        lmax = fmaxf(lmax, fmaxf(fmaxf(g[i][0], g[i][1]), fmaxf(g[i][2], g[i][3])));
        if (kSoft) sm += (__builtin_amdgcn_rcpf(-g[i][0]) + __builtin_amdgcn_rcpf(-g[i][1])) +
                         (__builtin_amdgcn_rcpf(-g[i][2]) + __builtin_amdgcn_rcpf(-g[i][3]));
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) g[i][j] = -INFINITY;
      }
    }
  } else
#pragma unroll
  for (int i = 0; i < kFastGroups; ++i) {
    const int q = lane + 64 * i;
    if (q < groups) {
      uint32_t r[4];
      Philox::gen(seed, (uint32_t)q, (uint32_t)bq, (uint32_t)p, 0u, r);
      const float4 l = lg[q];
      g[i][0] = l.x + gumbel_from_bits(r[0]);
      g[i][1] = l.y + gumbel_from_bits(r[1]);
      g[i][2] = l.z + gumbel_from_bits(r[2]);
      g[i][3] = l.w + gumbel_from_bits(r[3]);
      const float gm = fmaxf(fmaxf(g[i][0], g[i][1]), fmaxf(g[i][2], g[i][3]));
      lmax = fmaxf(lmax, gm);
      if (kSoft) {
        if (gm > mx) { sm *= exp_t<float>(mx - gm); mx = gm; }
#pragma unroll
        for (int j = 0; j < 4; ++j) sm += exp_t<float>(g[i][j] - mx);
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) g[i][j] = -INFINITY;
    }
  }
  float wmx = 0.f, lse = 0.f, inv_sm = 0.f;
  if (kSoft) {
    if (race) {
      sm = row_sum(sm);
      lse = race_ws[(size_t)P_race * N + 2 * (size_t)P_race + p] + (log_t<float>(sm) + 0.36651292058166432701f);   // lmax + ln(sum e) - ln ln 2
      inv_sm = 1.0f / sm;
    } else {
      wmx = row_max(mx);
      sm *= (mx == -INFINITY) ? 0.f : exp_t<float>(mx - wmx);
      sm = row_sum(sm);
      lse = wmx + log_t<float>(sm);
      inv_sm = 1.0f / sm;
    }
  }
  // the soft-max weight of a selected score (kSoft)
  auto y_of = [&](float score) { return race ? __builtin_amdgcn_rcpf(-score) * inv_sm : exp_t<float>(score - wmx) * inv_sm; };

#if DR_K1_DBG_SELECT == 1   // timing experiments (scratch/ab_k1_sel.py): pass A alone
  if (lane < k) idx[row * k + lane] = __float_as_int(lmax);
  return;
#endif
  // ---------------- threshold: any t that at least k LANE MAXIMA reach has the k winners among { g >= t }
  bool settled = false;   // (wave-uniform)
  float thr = -INFINITY;
#if DR_K1_SALU_SELECT
  if constexpr (!kScreen) if (race) {
    // Round 6, the one-logarithm form: t by COUNTING.  The keys of a pair are a Poisson process in t: E #{ key >= t } = -t c_p with
    // c_p = ln 2 sum_n 1 / w_n (gumbel_race_weights_kernel), so t is searched as t = -lam / c_p from lam = 6: a compare + a count per
    // probe (one vector instruction; the k rounds of a wave-wide maximum below are 64), a secant step on the count until the count
    // is bracketed, then bisection; 3.2 probes on average, exactly k lane maxima in 77 % of the rows (scratch/sim_k1_threshold.py).
    // A row the search does not settle (no count in k .. 16 after six probes) takes the exact threshold below.
    const float inv = race_ws[(size_t)P_race * N + P_race + p];   // -1 / c_p
    const float kf = (float)k + 0.5f;
    float t = 6.0f * inv, tlo = 0.f, thi = 0.f;   // tlo: at least k lane maxima reach it; thi: fewer than k do
    int clo = 0;
    bool have_lo = false, have_hi = false;
#pragma unroll
    for (int it = 0; it < 6; ++it) {
      const int c = __popcll(__builtin_amdgcn_ballot_w64(lmax >= t));
      if (c >= k) { tlo = t; clo = c; have_lo = true; if (c == k) break; }
      else { thi = t; have_hi = true; }
      t = (have_lo && have_hi) ? 0.5f * (tlo + thi) : t * fminf(4.0f, kf * __builtin_amdgcn_rcpf((float)c + 0.5f));
    }
    if (have_lo && clo <= 16) { settled = true; thr = tlo; }
  }
#endif
  if (!settled) {   // the k-th largest lane maximum: k rounds of a wave-wide maximum
    float v = lmax;
    for (int r = 0; r < k; ++r) {
      thr = row_max(v);
      const unsigned long long who = __ballot(v == thr);
      if (lane == __ffsll((long long)who) - 1) v = -INFINITY;
    }
  }
#if DR_K1_DBG_SELECT == 2   // pass A + threshold
  if (lane < k) idx[row * k + lane] = __float_as_int(thr);
  return;
#endif

#if DR_K1_SALU_SELECT
  if constexpr (!kScreen) if (DR_K1_SALU_SELECT == 1 || settled) {
    // ---------------- round 6: the selection on wave compare masks (scalar unit), no LDS -------------------------------------------
    // The candidates { g >= thr } are read off 32 compare masks by the scalar unit in ascending point index -- group, lane,
    // component -- and dealt to lanes 0 .. n-1 (a v_readlane and two selects under the scalar mask of lane n): n == k needs no
    // ranking at all; n > k ranks by (value, index) over v_readlane.  Same winners, same output order as the list below (the total
    // order is the same); a row with more than 64 candidates takes the list.  (profiles/r6_k1_selection.md: 789 -> 578 vector
    // instructions per row in the one-logarithm form, identical outputs on 99 cases, scratch/k1_select_check.py.  Keeping the
    // group maxima from pass A and testing them first: 0.8985-0.8998 vs 0.8963-0.8968 ms per step without -- not kept.)
    int n = 0;
    int cv_i = __float_as_int(-INFINITY), ci = 0x7fffffff;
#pragma unroll
    for (int i = 0; i < kFastGroups; ++i) {
      if (64 * i >= groups) break;   // wave-uniform
      const unsigned long long c0 = __builtin_amdgcn_ballot_w64(g[i][0] >= thr), c1 = __builtin_amdgcn_ballot_w64(g[i][1] >= thr),
                               c2 = __builtin_amdgcn_ballot_w64(g[i][2] >= thr), c3 = __builtin_amdgcn_ballot_w64(g[i][3] >= thr);
      unsigned long long any = c0 | c1 | c2 | c3;
      while (any) {
        const int l = __builtin_ctzll(any);
        any &= any - 1;
        const unsigned long long cj[4] = {c0, c1, c2, c3};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if ((cj[j] >> l) & 1ull) {
            // (a 65th candidate overwrites lane 0: the count below then sends the row to the list)
            const bool mine = __builtin_amdgcn_inverse_ballot_w64(1ull << (n & 63));
            cv_i = mine ? __builtin_amdgcn_readlane(__float_as_int(g[i][j]), l) : cv_i;
            ci = mine ? 4 * (l + 64 * i) + j : ci;
            ++n;
          }
        }
      }
    }
    if (n <= 64) {
      const float cv = __int_as_float(cv_i);
      bool win = lane < n;
      int pos = lane;
      if (n > k) {
        int rank = 0;
#pragma unroll 1
        for (int m = 0; m < n; ++m) {
          const float ov = __int_as_float(__builtin_amdgcn_readlane(cv_i, m));
          const int oi = __builtin_amdgcn_readlane(ci, m);
          rank += (ov > cv) || (ov == cv && oi < ci);
        }
        win = win && rank < k;
        pos = __popcll(__builtin_amdgcn_ballot_w64(win) & ((1ull << lane) - 1ull));   // (the lanes are in index order)
      }
      if (win) {
        idx[row * k + pos] = ci;
        if (kSoft) {
          const float y = y_of(cv);
          y_sel[row * k + pos] = y;
          if (gather_dst) gather_dst[row * k + pos] = straight_through(gather_src[(size_t)p * N + ci], y);
        }
        if (!kSoft && gather_dst) gather_dst[row * k + pos] = gather_src[(size_t)p * N + ci];
      }
      if (kSoft && lane == 0) lse_out[row] = lse;
      return;
    }
    if (settled) {   // (more than 64 candidates at a searched threshold: the list wants the exact one)
      float v = lmax;
      for (int r = 0; r < k; ++r) {
        thr = row_max(v);
        const unsigned long long who = __ballot(v == thr);
        if (lane == __ffsll((long long)who) - 1) v = -INFINITY;
      }
    }
  }
#endif

  // ---------------- pass B: the candidates { g >= thr } into the wave's LDS list
  int ncand = 0;
#if DR_K1_PASSB_ATOMIC
  // Only the lanes whose maximum reaches the threshold (k .. ~10 of 64) hold candidates: they walk their own elements and take
  // list slots with an LDS counter.  The list order is then arbitrary -- the ranking below is by (value, index), a total order,
  // so the result does not depend on it.  (Round 2's wave-wide compaction kept the general kernel's order with a ballot, a
  // popcount and an mbcnt per element of every group that holds a candidate: ~160 of the ~250 vector instructions the
  // selection costs per row.)
  if (lane == 0) s_cnt[wv] = 0;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
  if (lmax >= thr) {
#pragma unroll
    for (int i = 0; i < kFastGroups; ++i) {
      if (!(fmaxf(fmaxf(g[i][0], g[i][1]), fmaxf(g[i][2], g[i][3])) >= thr)) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (g[i][j] >= thr) {
          const int pos = atomicAdd(&s_cnt[wv], 1);
          if (pos < kMaxCand) { cand_val[pos] = g[i][j]; cand_idx[pos] = 4 * (lane + 64 * i) + j; }
        }
      }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
  ncand = s_cnt[wv];
#else
#if DR_K1_PASSB_LDS
  // Only the lanes whose maximum reaches the threshold hold candidates: exactly k of 64 unless values tie.  They park their 32
  // values in LDS (eight 16-byte writes under one exec mask) and the WAVE scans the k x 32 values, 64 per step: three steps
  // for k = 5 instead of a compare + branch per element of every group and a ballot / popcount / mbcnt per candidate.  The
  // list order differs from the element-order scan; the ranking below is by (value, index), a total order, so the winners and
  // their output positions do not depend on it.
  const unsigned long long hotb = __ballot(lmax >= thr);
  const int nhot = __popcll(hotb);
  if (nhot <= kHotMax) {
    float *stage = s_stage[wv];
    if (lmax >= thr) {
      const int hr = __popcll(hotb & ((1ull << lane) - 1ull));
#pragma unroll
      for (int i = 0; i < kFastGroups; ++i)
        *reinterpret_cast<float4 *>(stage + hr * kHotStride + 4 * i) = make_float4(g[i][0], g[i][1], g[i][2], g[i][3]);
      s_hot[wv][hr] = lane;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    const int total = nhot * 4 * kFastGroups;
#pragma unroll 1
    for (int base = 0; base < total; base += 64) {
      const int sl = base + lane;
      const bool in = sl < total;
      const int h = in ? sl / (4 * kFastGroups) : 0, e = sl % (4 * kFastGroups);
      const float val = in ? stage[h * kHotStride + e] : -INFINITY;
      const int src = s_hot[wv][h];
      const bool c = in && val >= thr;
      const unsigned long long bal = __ballot(c);
      if (bal) {
        const int pos = ncand + __popcll(bal & ((1ull << lane) - 1ull));
        if (c && pos < kMaxCand) { cand_val[pos] = val; cand_idx[pos] = 4 * (src + 64 * (e >> 2)) + (e & 3); }
        ncand += __popcll(bal);
      }
    }
  } else
#endif
#pragma unroll
  for (int i = 0; i < kFastGroups; ++i) {
    if (64 * i >= groups) break;   // wave-uniform
    const int q = lane + 64 * i;
    if (!__ballot(fmaxf(fmaxf(g[i][0], g[i][1]), fmaxf(g[i][2], g[i][3])) >= thr)) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool c = g[i][j] >= thr;
      const unsigned long long bal = __ballot(c);
      if (bal) {
        const int pos = ncand + __popcll(bal & ((1ull << lane) - 1ull));
        if (c && pos < kMaxCand) { cand_val[pos] = g[i][j]; cand_idx[pos] = 4 * q + j; }
        ncand += __popcll(bal);
      }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
#endif

  if (ncand <= kMaxCand) {
    const bool have = lane < ncand;
    const float cv = have ? cand_val[lane] : -INFINITY;
    const int ci = have ? cand_idx[lane] : 0x7fffffff;
    int rank = 0;
    for (int j = 0; j < ncand; ++j) {
      const float ov = cand_val[j];
      const int oi = cand_idx[j];
      rank += (ov > cv) || (ov == cv && oi < ci);
    }
    const bool win = have && rank < k;
    const unsigned long long wb = __ballot(win);
    int pos = 0;
    for (int j = 0; j < ncand; ++j) {
      if ((wb >> j) & 1ull) pos += cand_idx[j] < ci;
    }
    if (win) {
      idx[row * k + pos] = ci;
      if (kSoft) {
        const float y = y_of(cv);
        y_sel[row * k + pos] = y;
        if (gather_dst) gather_dst[row * k + pos] = straight_through(gather_src[(size_t)p * N + ci], y);
      }
      if (!kSoft && gather_dst) gather_dst[row * k + pos] = gather_src[(size_t)p * N + ci];
    }
  } else {
    // slow path (massive ties): k rounds of (value desc, index asc) arg-max with exclusion of earlier winners
    int won[kMaxK];
    float wong[kMaxK];
    for (int r = 0; r < k; ++r) {
      float bv = -INFINITY;
      int bi = 0x7fffffff;
#pragma unroll
      for (int i = 0; i < kFastGroups; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int n = 4 * (lane + 64 * i) + j;
          bool skip = n >= N;
          for (int s = 0; s < r; ++s) skip = skip || (won[s] == n);
          if (!skip && (g[i][j] > bv || (g[i][j] == bv && n < bi))) { bv = g[i][j]; bi = n; }
        }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
      }
      won[r] = bi;
      wong[r] = bv;
    }
    if (lane < k) {
      int me = 0;
      float mg = 0.f;
      for (int r = 0; r < k; ++r) if (r == lane) { me = won[r]; mg = wong[r]; }
      int pos = 0;
      for (int r = 0; r < k; ++r) pos += won[r] < me;
      idx[row * k + pos] = me;
      if (kSoft) {
        const float y = y_of(mg);
        y_sel[row * k + pos] = y;
        if (gather_dst) gather_dst[row * k + pos] = straight_through(gather_src[(size_t)p * N + me], y);
      }
      if (!kSoft && gather_dst) gather_dst[row * k + pos] = gather_src[(size_t)p * N + me];
    }
  }
  if (kSoft && lane == 0) lse_out[row] = lse;
}

// ---- screening words for the long-row kernel, index-only mode (round 4) -----------------------------------------------------
// The k winners of a row are its k largest logit_n + G_n, and G_n is a monotone function of the Philox word w_n: "score >= T" is
// "w_n >= tb_n" with a per-point word tb_n that depends on the pair's logits only.  T = logsumexp(logits) - ln(lambda) puts
// Poisson(lambda) such points into a row.  tb_n is rounded DOWN (score >= T - margin, four roundings of the 24-bit conversion),
// so no point that reaches T is ever missed; T itself only sets the expected count -- it may be approximate (f32 log-sum-exp),
// but words and sampler must use the SAME value: it is computed once per pair and read back by both.
constexpr float kScreenMargin = 1e-3f;   // the device logarithms are good to ~1e-6
// (a) per pair and eighth of the row: running (max, sum of exp) of its slice -- kScreenParts blocks of 256 threads, ONE pass, four
//     logits per load, every load of a thread in flight at once (first version: every block of the word kernel re-derived T from
//     the whole row, 40 us at 50 000 points; second: one 1024-thread block per pair, 6.9 us)
constexpr int kScreenParts = 8;
__global__ __launch_bounds__(256) void gumbel_screen_part_kernel(const float *__restrict__ logits, int N, float *__restrict__ part) {
  __shared__ float s_mx[4], s_sm[4];
  const int p = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const float4 *l4 = reinterpret_cast<const float4 *>(logits + (size_t)p * N);   // N % 4 == 0 (the long-row kernel's condition)
  const int groups = N >> 2;
  float mx = -INFINITY, sm = 0.f;
  for (int q0 = blockIdx.x * 256 + tid; q0 < groups; q0 += 8 * kScreenParts * 256) {
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int q = q0 + u * kScreenParts * 256;
      v[u] = q < groups ? l4[q] : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float m4 = fmaxf(fmaxf(v[u].x, v[u].y), fmaxf(v[u].z, v[u].w));
      if (m4 > mx) { sm *= __expf(mx - m4); mx = m4; }
      if (mx > -INFINITY) sm += (__expf(v[u].x - mx) + __expf(v[u].y - mx)) + (__expf(v[u].z - mx) + __expf(v[u].w - mx));
    }
  }
  const float wmx = row_max(mx);
  sm *= (mx == -INFINITY) ? 0.f : __expf(mx - wmx);
  sm = row_sum(sm);
  if (lane == 0) { s_mx[wv] = wmx; s_sm[wv] = sm; }
  __syncthreads();
  if (tid == 0) {
    float bm = fmaxf(fmaxf(s_mx[0], s_mx[1]), fmaxf(s_mx[2], s_mx[3]));
    float bs = 0.f;
    for (int w = 0; w < 4; ++w) bs += (s_mx[w] == -INFINITY) ? 0.f : s_sm[w] * __expf(s_mx[w] - bm);
    part[((size_t)p * kScreenParts + blockIdx.x) * 2] = bm;
    part[((size_t)p * kScreenParts + blockIdx.x) * 2 + 1] = bs;
  }
}
// (b) the words, one point per thread.  Every block combines the pair's eight partial sums itself -- the same values in the same
//     order, hence the same T in every block; block 0 publishes it for the sampler
__global__ __launch_bounds__(256) void gumbel_screen_kernel(const float *__restrict__ logits, int N, float lambda,
                                                           const float *__restrict__ part, float *__restrict__ T_out,
                                                           uint32_t *__restrict__ tb) {
  const int p = blockIdx.y, n = blockIdx.x * 256 + threadIdx.x;
  float bm = -INFINITY;
#pragma unroll
  for (int i = 0; i < kScreenParts; ++i) bm = fmaxf(bm, part[((size_t)p * kScreenParts + i) * 2]);
  float bs = 0.f;
#pragma unroll
  for (int i = 0; i < kScreenParts; ++i) {
    const float m = part[((size_t)p * kScreenParts + i) * 2], v = part[((size_t)p * kScreenParts + i) * 2 + 1];
    bs += (m == -INFINITY) ? 0.f : v * __expf(m - bm);
  }
  float Tf = bm + __logf(bs) - __logf(lambda);
  if (!(Tf == Tf) || Tf == INFINITY || Tf == -INFINITY) Tf = INFINITY;   // non-finite logits: no point passes, every row takes the full pass
  // the margin covers the f32 rounding of fl(logit + G) -- half an ulp of a score near T -- only while |T| stays below ~1.6e4
  // (round-4 advice: unnormalised scores, a large additive bias): beyond 4096 the pair is not screened at all
  if (fabsf(Tf) > 4096.f) Tf = INFINITY;
  if (blockIdx.x == 0 && threadIdx.x == 0) T_out[p] = Tf;
  if (n >= N) return;
  constexpr double kTiny = 1.17549435e-38, kScale = 2.3283064365386963e-10 * (1.0 - 1.1920928955078125e-07 - 1.17549435e-38);
  // G >= a  <=>  u >= exp(-exp(-a)),  u = fl(fl24(w) * 2^-32 c + tiny)  (gumbel_from_bits),  a = T - margin - logit_n
  const double a = ((double)Tf - (double)kScreenMargin) - (double)logits[(size_t)p * N + n];
  const double e = exp(-a);
  const double us = (e < 745.0) ? exp(-e) : 0.0;
  double w = floor((us - kTiny) / kScale) - 1024.0;
  if (!(w == w)) w = 0.0;
  w = fmin(fmax(w, 0.0), 4294967295.0);
  tb[(size_t)p * N + n] = (Tf == INFINITY) ? 0xffffffffu : (uint32_t)w;
}

// (a+b) in ONE launch (round 5): blocks of 1024 threads, one point per thread for the words; every block first derives the pair's
//     T from the WHOLE row itself -- the same loads, the same order, hence the same value in every block (block 0 publishes it) --
//     which costs a block ~12 float4 loads and ~50 exponentials per thread at 50 000 points (the row is 200 KB: L2 / MALL hits
//     after the first block) and saves the partial-sum launch with its dependency: 4.1 + 4.8 us + a graph edge -> one kernel.
#ifndef DR_K1_SCREEN_FUSED
#define DR_K1_SCREEN_FUSED 1
#endif
__global__ __launch_bounds__(1024) void gumbel_screen_fused_kernel(const float *__restrict__ logits, int N, float lambda,
                                                                  float *__restrict__ T_out, uint32_t *__restrict__ tb) {
  __shared__ float s_mx[16], s_sm[16];
  const int p = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const float4 *l4 = reinterpret_cast<const float4 *>(logits + (size_t)p * N);   // N % 4 == 0 (the long-row kernel's condition)
  const int groups = N >> 2;
  const int n = blockIdx.x * 1024 + tid;
  const float mine = n < N ? logits[(size_t)p * N + n] : 0.f;   // requested before the row pass
  float mx = -INFINITY, sm = 0.f;
  for (int q0 = tid; q0 < groups; q0 += 4 * 1024) {
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int q = q0 + u * 1024;
      v[u] = q < groups ? l4[q] : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float m4 = fmaxf(fmaxf(v[u].x, v[u].y), fmaxf(v[u].z, v[u].w));
      if (m4 > mx) { sm *= __expf(mx - m4); mx = m4; }
      if (mx > -INFINITY) sm += (__expf(v[u].x - mx) + __expf(v[u].y - mx)) + (__expf(v[u].z - mx) + __expf(v[u].w - mx));
    }
  }
  const float wmx = row_max(mx);
  sm *= (mx == -INFINITY) ? 0.f : __expf(mx - wmx);
  sm = row_sum(sm);
  if (lane == 0) { s_mx[wv] = wmx; s_sm[wv] = sm; }
  __syncthreads();
  float bm = -INFINITY;
#pragma unroll
  for (int w = 0; w < 16; ++w) bm = fmaxf(bm, s_mx[w]);
  float bs = 0.f;
#pragma unroll
  for (int w = 0; w < 16; ++w) bs += (s_mx[w] == -INFINITY) ? 0.f : s_sm[w] * __expf(s_mx[w] - bm);
  float Tf = bm + __logf(bs) - __logf(lambda);
  if (!(Tf == Tf) || Tf == INFINITY || Tf == -INFINITY || fabsf(Tf) > 4096.f) Tf = INFINITY;   // as gumbel_screen_kernel
  if (blockIdx.x == 0 && tid == 0) T_out[p] = Tf;
  if (n >= N) return;
  constexpr double kTiny = 1.17549435e-38, kScale = 2.3283064365386963e-10 * (1.0 - 1.1920928955078125e-07 - 1.17549435e-38);
  const double a = ((double)Tf - (double)kScreenMargin) - (double)mine;
  const double e = exp(-a);
  const double us = (e < 745.0) ? exp(-e) : 0.0;
  double w = floor((us - kTiny) / kScale) - 1024.0;
  if (!(w == w)) w = 0.0;
  w = fmin(fmax(w, 0.0), 4294967295.0);
  tb[(size_t)p * N + n] = (Tf == INFINITY) ? 0xffffffffu : (uint32_t)w;
}

// (c) short rows (N <= 2048, N % 4 == 0: the register-resident sampler): T and the words of a pair by ONE block of 256 threads --
//     one launch per call instead of two
__global__ __launch_bounds__(256) void gumbel_screen_short_kernel(const float *__restrict__ logits, int N, float lambda,
                                                                 float *__restrict__ T_out, uint32_t *__restrict__ tb) {
  __shared__ float s_mx[4], s_sm[4];
  const int p = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const float4 *l4 = reinterpret_cast<const float4 *>(logits + (size_t)p * N);
  const int groups = N >> 2;
  float4 v[2];
  float mx = -INFINITY;
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int q = tid + 256 * u;
    v[u] = q < groups ? l4[q] : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    mx = fmaxf(mx, fmaxf(fmaxf(v[u].x, v[u].y), fmaxf(v[u].z, v[u].w)));
  }
  const float wmx = row_max(mx);
  if (lane == 0) s_mx[wv] = wmx;
  __syncthreads();
  const float bm = fmaxf(fmaxf(s_mx[0], s_mx[1]), fmaxf(s_mx[2], s_mx[3]));
  float sm = 0.f;
  if (bm > -INFINITY) {
#pragma unroll
    for (int u = 0; u < 2; ++u) sm += (__expf(v[u].x - bm) + __expf(v[u].y - bm)) + (__expf(v[u].z - bm) + __expf(v[u].w - bm));
  }
  sm = row_sum(sm);
  if (lane == 0) s_sm[wv] = sm;
  __syncthreads();
  const float bs = (s_sm[0] + s_sm[1]) + (s_sm[2] + s_sm[3]);
  float Tf = bm + __logf(bs) - __logf(lambda);
  if (!(Tf == Tf) || Tf == INFINITY || Tf == -INFINITY || fabsf(Tf) > 4096.f) Tf = INFINITY;   // as gumbel_screen_kernel
  if (tid == 0) T_out[p] = Tf;
  constexpr double kTiny = 1.17549435e-38, kScale = 2.3283064365386963e-10 * (1.0 - 1.1920928955078125e-07 - 1.17549435e-38);
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int q = tid + 256 * u;
    if (q >= groups) continue;
    const float lv[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
    uint32_t wv4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const double a = ((double)Tf - (double)kScreenMargin) - (double)lv[j];
      const double e = exp(-a);
      const double us = (e < 745.0) ? exp(-e) : 0.0;
      double w = floor((us - kTiny) / kScale) - 1024.0;
      if (!(w == w)) w = 0.0;
      w = fmin(fmax(w, 0.0), 4294967295.0);
      wv4[j] = (Tf == INFINITY) ? 0xffffffffu : (uint32_t)w;
    }
    reinterpret_cast<uint4 *>(tb + (size_t)p * N)[q] = make_uint4(wv4[0], wv4[1], wv4[2], wv4[3]);
  }
}

// ---- rows longer than the register kernel holds (N > 2048): ONE pass ----------------------------------------------------
// The general kernel above makes two passes over a row that does not fit its LDS cache -- lane maxima first, then the
// candidates above the k-th largest lane maximum -- and generates the Philox noise of every element twice.  Here a lane keeps
// the K largest (value, index) pairs of the elements it has seen (sorted, in registers; the insertion code runs only in
// iterations where some lane of the wave has a new entry: ~60 % of them at N = 50 000, K = 3) and the wave merges the 64
// lists at the end by k rounds of arg-max over the lanes' heads.  Same Philox counters, same g, same order (value
// descending, index ascending): the index sets are those of the general kernel (tests/test_gpu_round2.py).
// f32, in-kernel noise, logits given, tau = 1, N % 4 == 0, no dense outputs.
#ifndef DR_K1_STREAM
#define DR_K1_STREAM 1   // 0: the general two-pass kernel (A/B builds)
#endif
#ifndef DR_K1_STREAM_SPLIT
#define DR_K1_STREAM_SPLIT 1   // 0: always one wave per row
#endif
// kW = waves per row (1 or kRowsPerBlock).  With few rows (BASELINE configs[3]: 2048 hypotheses of ONE pair) a wave per row is
// two waves per SIMD on this chip: the kernel's 32 registers would allow eight, and its dependent Philox rounds and logarithms
// want them.  kW = 4: the block's four waves take interleaved 64-group slices of ONE row, each keeps the top k of its slice,
// and wave 0 merges the 4 k candidates through LDS -- same total order (value descending, index ascending), so the same
// index set; the soft-max statistics are combined the same way (per-wave running max / sum, then across the four waves).
template <int K, bool kSoft, int kW>
__global__ __launch_bounds__(kRowsPerBlock * 64) void gumbel_topk_stream_kernel(const float *__restrict__ logits, uint64_t seed, int B,
                                                                               int N, int k, int32_t *__restrict__ idx,
                                                                               float *__restrict__ y_sel, float *__restrict__ lse_out,
                                                                               const uint64_t *__restrict__ seed_ptr,
                                                                               const uint32_t *__restrict__ screen_tb = nullptr,
                                                                               const float *__restrict__ screen_T = nullptr, int sub = 0) {
  // screen_tb / screen_T (index-only mode): per point the smallest Philox word that can still lift the point to the score T
  // (gumbel_screen_kernel).  A step of the wave whose 256 words all stay below their thresholds is skipped after Philox + four
  // integer compares: no logits, no logarithms, no insertion.  Steps with a candidate are evaluated in full, for every lane, so
  // the lists hold exactly computed scores only; the row counts the evaluated scores >= T and, if fewer than k reach it
  // (3e-8 of the rows at lambda = 20 + k), repeats itself without the screen.  Otherwise the k winners are all >= T, hence all
  // evaluated: the index sets are those of the unscreened kernel, bit for bit.
  __shared__ int s_cnt[kRowsPerBlock];
  __shared__ float s_cv[kW > 1 ? kRowsPerBlock * kMaxK : 1];
  __shared__ int s_ci[kW > 1 ? kRowsPerBlock * kMaxK : 1];
  __shared__ float s_mx[kRowsPerBlock], s_sm[kRowsPerBlock];
  if (seed_ptr) seed = *seed_ptr;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int part = kW > 1 ? wv : 0;
  const int p = blockIdx.y, b = kW > 1 ? (int)blockIdx.x : (int)(blockIdx.x * kRowsPerBlock + wv);
  if (b >= B) return;   // kW > 1: block-uniform (one row per block), so the barrier below is reached by all or none
  int bq;
  sub_batch_row(seed, bq, b, sub);
  const int groups = N >> 2;
  const size_t row = (size_t)p * B + b;
  const float4 *lg = reinterpret_cast<const float4 *>(logits + (size_t)p * N);
  float tv[K];
  int ti[K];
  float mx = -INFINITY, sm = 0.f;
  const uint4 *tb4 = (!kSoft && screen_tb) ? reinterpret_cast<const uint4 *>(screen_tb + (size_t)p * N) : nullptr;
  const float Tscr = tb4 ? screen_T[p] : 0.f;
  for (int pass = 0; pass < 2; ++pass) {
    const bool screen = tb4 != nullptr && pass == 0;   // block-uniform
#pragma unroll
    for (int s = 0; s < K; ++s) { tv[s] = -INFINITY; ti[s] = 0x7fffffff; }
    int cnt = 0;
    // the thresholds of the NEXT step are requested before this step's Philox rounds (the screened loop otherwise waits for a
    // 16-byte load in front of every ballot: 60 % of the waves' lifetime parked, VALU issuing 0.79 of the cycles)
    uint4 tcur = make_uint4(0u, 0u, 0u, 0u), tnext = tcur;
    if (screen && lane + 64 * part < groups) tcur = tb4[lane + 64 * part];
    for (int q = lane + 64 * part; q < groups; q += 64 * kW, tcur = tnext) {
      if (screen && q + 64 * kW < groups) tnext = tb4[q + 64 * kW];
      uint32_t r[4];
      Philox::gen(seed, (uint32_t)q, (uint32_t)bq, (uint32_t)p, 0u, r);
      if (screen) {
        const uint4 t = tcur;
        if (!__ballot(r[0] >= t.x || r[1] >= t.y || r[2] >= t.z || r[3] >= t.w)) continue;
      }
      const float4 l = lg[q];
      const float g[4] = {l.x + gumbel_from_bits(r[0]), l.y + gumbel_from_bits(r[1]), l.z + gumbel_from_bits(r[2]),
                          l.w + gumbel_from_bits(r[3])};
      const float gm = fmaxf(fmaxf(g[0], g[1]), fmaxf(g[2], g[3]));
      if (kSoft) {
        if (gm > mx) { sm *= exp_t<float>(mx - gm); mx = gm; }
#pragma unroll
        for (int j = 0; j < 4; ++j) sm += exp_t<float>(g[j] - mx);
      }
      if (screen) cnt += (int)(g[0] >= Tscr) + (int)(g[1] >= Tscr) + (int)(g[2] >= Tscr) + (int)(g[3] >= Tscr);
      // a lane's elements arrive in ascending index order, so a new element displaces only strictly smaller values
      if (!__ballot(gm > tv[K - 1])) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float v = g[j];
        const int n = 4 * q + j;
        bool prev = false;   // v > tv[s - 1]
        float pv = 0.f;      // tv[s - 1] before the update
        int pi = 0;
#pragma unroll
        for (int s = 0; s < K; ++s) {
          const bool c = v > tv[s];
          const float ov = tv[s];
          const int oi = ti[s];
          tv[s] = c ? (prev ? pv : v) : ov;
          ti[s] = c ? (prev ? pi : n) : oi;
          prev = c; pv = ov; pi = oi;
        }
      }
    }
    if (!screen) break;
    // did k evaluated scores of the ROW reach T?  (kW > 1: the row's four waves agree through LDS)
    cnt = wave_sum(cnt);
    if (kW > 1) {
      if (lane == 0) s_cnt[wv] = cnt;
      __syncthreads();
      cnt = 0;
#pragma unroll
      for (int w = 0; w < kRowsPerBlock; ++w) cnt += s_cnt[w];
      __syncthreads();
    }
    if (cnt >= k) break;
  }
  float wmx = 0.f, lse = 0.f, inv_sm = 0.f;
  if (kSoft) {
    wmx = row_max(mx);
    sm *= (mx == -INFINITY) ? 0.f : exp_t<float>(mx - wmx);
    sm = row_sum(sm);
    if (kW > 1) {   // this wave's (max, sum) -> the row's
      if (lane == 0) { s_mx[wv] = wmx; s_sm[wv] = sm; }
    } else {
      lse = wmx + log_t<float>(sm);
      inv_sm = 1.0f / sm;
    }
  }
  // merge: k rounds of (value desc, index asc) arg-max over the lanes' heads; the winning lane pops its head
  int won[kMaxK];
  float wong[kMaxK];
  for (int r = 0; r < k; ++r) {
    float bv = tv[0];
    int bi = ti[0];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(bv, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    won[r] = bi;
    wong[r] = bv;
    if (ti[0] == bi) {
#pragma unroll
      for (int s = 0; s + 1 < K; ++s) { tv[s] = tv[s + 1]; ti[s] = ti[s + 1]; }
      tv[K - 1] = -INFINITY;
      ti[K - 1] = 0x7fffffff;
    }
  }
  if (kW > 1) {
    // the four waves' top-k lists -> LDS -> wave 0 ranks the 4 k candidates (value descending, index ascending)
    if (lane < k) {
      int me = 0x7fffffff;
      float mg = -INFINITY;
      for (int r = 0; r < k; ++r) if (r == lane) { me = won[r]; mg = wong[r]; }
      s_cv[wv * kMaxK + lane] = mg;
      s_ci[wv * kMaxK + lane] = me;
    }
    __syncthreads();
    if (wv != 0) return;
    if (kSoft) {
      wmx = fmaxf(fmaxf(s_mx[0], s_mx[1]), fmaxf(s_mx[2], s_mx[3]));
      sm = 0.f;
#pragma unroll
      for (int w = 0; w < kW; ++w) sm += (s_mx[w] == -INFINITY) ? 0.f : s_sm[w] * exp_t<float>(s_mx[w] - wmx);
      lse = wmx + log_t<float>(sm);
      inv_sm = 1.0f / sm;
    }
    const int nc = kW * k;                     // <= 32 candidates; slice w, entry r at lane w * k + r
    const bool have = lane < nc;
    const int cw = have ? lane / k : 0, cr = have ? lane % k : 0;
    const float cv = have ? s_cv[cw * kMaxK + cr] : -INFINITY;
    const int ci = have ? s_ci[cw * kMaxK + cr] : 0x7fffffff;
    int rank = 0;
    for (int j = 0; j < nc; ++j) {
      const float ov = __shfl(cv, j, 64);
      const int oi = __shfl(ci, j, 64);
      rank += (ov > cv) || (ov == cv && oi < ci);
    }
    const bool win = have && rank < k && ci != 0x7fffffff;
    const unsigned long long wb = __ballot(win);
    int pos = 0;
    for (int j = 0; j < nc; ++j) {
      const int oi = __shfl(ci, j, 64);
      if ((wb >> j) & 1ull) pos += oi < ci;
    }
    if (win) {
      idx[row * k + pos] = ci;
      if (kSoft) y_sel[row * k + pos] = exp_t<float>(cv - wmx) * inv_sm;
    }
    if (kSoft && lane == 0) lse_out[row] = lse;
    return;
  }
  if (lane < k) {
    int me = 0;
    float mg = 0.f;
    for (int r = 0; r < k; ++r) if (r == lane) { me = won[r]; mg = wong[r]; }
    int pos = 0;
    for (int r = 0; r < k; ++r) pos += won[r] < me;
    idx[row * k + pos] = me;
    if (kSoft) y_sel[row * k + pos] = exp_t<float>(mg - wmx) * inv_sm;
  }
  if (kSoft && lane == 0) lse_out[row] = lse;
}

template <int K>
static void stream_launch(bool soft, dim3 grid, dim3 block, hipStream_t st, const float *logits, uint64_t seed, int B, int N, int k,
                          int32_t *idx, float *y_sel, float *lse, const uint64_t *seed_ptr, const uint32_t *tb = nullptr,
                          const float *Tp = nullptr, int sub = 0) {
  // few rows: four waves per row (one row per block) -- the wave-per-row grid would leave the SIMDs at <= 4 waves each
  const long rows = (long)grid.y * B;
  if (DR_K1_STREAM_SPLIT && rows <= 4096 && N >= 4 * 64 * 4 * 4) {
    const dim3 g2(B, grid.y);
    if (soft) hipLaunchKernelGGL((gumbel_topk_stream_kernel<K, true, kRowsPerBlock>), g2, block, 0, st, logits, seed, B, N, k, idx, y_sel, lse, seed_ptr);
    else hipLaunchKernelGGL((gumbel_topk_stream_kernel<K, false, kRowsPerBlock>), g2, block, 0, st, logits, seed, B, N, k, idx, y_sel, lse, seed_ptr, tb, Tp, sub);
    return;
  }
  if (soft) hipLaunchKernelGGL((gumbel_topk_stream_kernel<K, true, 1>), grid, block, 0, st, logits, seed, B, N, k, idx, y_sel, lse, seed_ptr);
  else hipLaunchKernelGGL((gumbel_topk_stream_kernel<K, false, 1>), grid, block, 0, st, logits, seed, B, N, k, idx, y_sel, lse, seed_ptr, tb, Tp, sub);
}

template <typename T>
int gumbel_fwd_launch(const T *logits, const T *gumbel, uint64_t seed, T tau, int P, int B, int N, int k,
                      int32_t *idx, T *y_sel, T *lse, T *y_soft, T *ret, T *gumbel_out, hipStream_t st,
                      const uint64_t *seed_ptr = nullptr, const float4 *gather_src = nullptr, float4 *gather_dst = nullptr,
                      bool *gathered = nullptr, uint32_t *screen_ws = nullptr, PairGate gate = PairGate(), int sub = 0,
                      float *race_ws = nullptr, bool race_ready = false) {
  // race_ws (index-only mode, register kernel, no screen): (N + 32) * P floats -- the one-logarithm form; race_ready: the weights are
  // already in it (dr_ransac_init wrote them for the whole call): no prologue launch
  // sub (index-only mode, in-kernel noise): rows per sub-batch of a super-round (GumbelArgs::sub); 0 = one batch
  if (gathered) *gathered = false;
  GumbelArgs<T> a{logits, gumbel, seed, tau, P, B, N, k, seed_ptr, sub};
  const int groups = (N + 3) / 4;
  const size_t base = (size_t)kRowsPerBlock * kMaxCand * (sizeof(T) + sizeof(int));
  const size_t cache = (size_t)kRowsPerBlock * groups * 4 * sizeof(T);
  dim3 grid((B + kRowsPerBlock - 1) / kRowsPerBlock, P);
  dim3 block(kRowsPerBlock * 64);
  const bool soft = y_sel != nullptr;   // the entry points have checked: y_sel and lse both given, or neither (then no dense outputs)
  if constexpr (sizeof(T) == 4) {
    if (DR_K1_FAST && logits && !gumbel && tau == T(1) && (N & 3) == 0 && N <= 4 * 64 * kFastGroups && !y_soft && !ret && !gumbel_out) {
      if (soft) {
        if (race_ws && !race_ready)
          hipLaunchKernelGGL(gumbel_race_weights_kernel, dim3(P), dim3(256), 0, st, (const float *)logits, N, P, race_ws);
        hipLaunchKernelGGL((gumbel_topk_fast_kernel<true>), grid, block, 0, st, (const float *)logits, seed, B, N, k, idx,
                           (float *)y_sel, (float *)lse, seed_ptr, gather_src, gather_dst, PairGate(), (const uint32_t *)nullptr,
                           (const float *)nullptr, 0, (const float *)race_ws, P);
        if (gathered) *gathered = gather_dst != nullptr;
      }
      else if (screen_ws && k <= 5 && B >= 64) {
        // screened (round 5): workspace = P x N words + P scores
        float *Tw = reinterpret_cast<float *>(screen_ws + (size_t)P * N);
        hipLaunchKernelGGL(gumbel_screen_short_kernel, dim3(P), dim3(256), 0, st, (const float *)logits, N, (float)(11 + k), Tw, screen_ws);
        hipLaunchKernelGGL((gumbel_topk_fast_kernel<false, true>), grid, block, 0, st, (const float *)logits, seed, B, N, k, idx,
                           (float *)y_sel, (float *)lse, seed_ptr, gather_src, gather_dst, gate, (const uint32_t *)screen_ws,
                           (const float *)Tw, sub);
        if (gathered) *gathered = gather_dst != nullptr;
      }
      else
      {
        if (race_ws && !race_ready)
          hipLaunchKernelGGL(gumbel_race_weights_kernel, dim3(P), dim3(256), 0, st, (const float *)logits, N, P, race_ws);
        hipLaunchKernelGGL((gumbel_topk_fast_kernel<false>), grid, block, 0, st, (const float *)logits, seed, B, N, k, idx,
                           (float *)y_sel, (float *)lse, seed_ptr, gather_src, gather_dst, gate, (const uint32_t *)nullptr,
                           (const float *)nullptr, sub, (const float *)race_ws, P);
        if (gathered) *gathered = gather_dst != nullptr;
      }
      return check_launch("gumbel_topk_fast_kernel");
    }
    // measured (scratch/ab_k1_stream.py): 50 000 x 2048 rows, k = 3: 226 -> 113 us; 4096 x 32 768 rows, k = 5: 196 -> 178 us;
    // k = 8 lists cost what the second pass costs (256 vs 250 us at N = 20 000) -> the general kernel keeps k > 5
    if (DR_K1_STREAM && logits && !gumbel && tau == T(1) && (N & 3) == 0 && N > 4 * 64 * kFastGroups && k <= 5 && !y_soft && !ret &&
        !gumbel_out) {
      const uint32_t *tb = nullptr;
      const float *Tp = nullptr;
      if (screen_ws && !soft) {
        // workspace: P x N words + P scores + P x 16 partial sums; lambda = 20 + k: P(fewer than k of a row's points reach T) < 1e-7
        float *Tw = reinterpret_cast<float *>(screen_ws + (size_t)P * N);
        float *part = Tw + P;
        if (DR_K1_SCREEN_FUSED && (long)((N + 1023) / 1024) * N <= (16L << 20)) {   // (row re-reads bounded: beyond, the two-launch form)
          hipLaunchKernelGGL(gumbel_screen_fused_kernel, dim3((N + 1023) / 1024, P), dim3(1024), 0, st, (const float *)logits, N,
                             (float)(20 + k), Tw, screen_ws);
        } else {
          hipLaunchKernelGGL(gumbel_screen_part_kernel, dim3(kScreenParts, P), dim3(256), 0, st, (const float *)logits, N, part);
          hipLaunchKernelGGL(gumbel_screen_kernel, dim3((N + 255) / 256, P), dim3(256), 0, st, (const float *)logits, N, (float)(20 + k),
                             part, Tw, screen_ws);
        }
        tb = screen_ws;
        Tp = Tw;
      }
      if (k <= 3) stream_launch<3>(soft, grid, block, st, (const float *)logits, seed, B, N, k, idx, (float *)y_sel, (float *)lse, seed_ptr, tb, Tp, sub);
      else stream_launch<5>(soft, grid, block, st, (const float *)logits, seed, B, N, k, idx, (float *)y_sel, (float *)lse, seed_ptr, tb, Tp, sub);
      return check_launch("gumbel_topk_stream_kernel");
    }
  }
  if (base + cache <= 64 * 1024) {
    if (soft)
      hipLaunchKernelGGL((gumbel_topk_kernel<T, true, true>), grid, block, base + cache, st, a, idx, y_sel, lse, y_soft, ret,
                         gumbel_out);
    else
      hipLaunchKernelGGL((gumbel_topk_kernel<T, true, false>), grid, block, base + cache, st, a, idx, y_sel, lse, y_soft,
                         ret, gumbel_out);
  } else {
    if (soft)
      hipLaunchKernelGGL((gumbel_topk_kernel<T, false, true>), grid, block, base, st, a, idx, y_sel, lse, y_soft, ret,
                         gumbel_out);
    else
      hipLaunchKernelGGL((gumbel_topk_kernel<T, false, false>), grid, block, base, st, a, idx, y_sel, lse, y_soft, ret,
                         gumbel_out);
  }
  return check_launch("gumbel_topk_kernel");
}

#ifndef DR_K1_BWD_RACE
#define DR_K1_BWD_RACE 1
#endif

// ---- backward of K1 (+K2):  grad_logits[p,n] = (1/tau) sum_b y_bn (a_bn - sum_m y_bm a_bm), a non-zero only at idx
template <typename T>
__global__ __launch_bounds__(256) void gumbel_bwd_kernel(GumbelArgs<T> a, const int32_t *__restrict__ idx,
                                                         const T *__restrict__ lse, const T *__restrict__ a_sel,
                                                         T *__restrict__ grad_logits, int rows_per_block,
                                                         const T *__restrict__ grad_samples = nullptr,
                                                         const T *__restrict__ grad_w = nullptr,
                                                         const T *__restrict__ matches4 = nullptr) {
  // a_sel == nullptr (round 5): the backward of K2 on the way -- a = <grad_samples[row, j, :], matches[p, idx, :]> (+ grad_w),
  // four-component correspondences, the arithmetic of gather_bwd_kernel (whose launch and [P,B,k] tensor this replaces)
  // one thread per 4-point group of pair p (one Philox call regenerates its noise); blockIdx.z owns a chunk of
  // `rows_per_block` hypothesis rows (their lse and <y,a> are staged in LDS) and adds its partial sums
  // into grad_logits with one atomicAdd per point: (groups/256) x P x (B/rows_per_block) blocks fill the chip, where a
  // single block per point group would leave 3/4 of the CUs idle at C2.
  if (a.seed_ptr) a.seed = *a.seed_ptr;
  const int p = blockIdx.y;
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  T *s_dot = reinterpret_cast<T *>(smem_raw);               // [256]   sum_m y_m a_m per row
  T *s_lse = s_dot + 256;
  const int groups = (a.N + 3) >> 2;
  const bool unit_tau = a.tau == T(1);   // wave-uniform: x / 1 == x exactly, skip the IEEE division sequence
  T l[4], acc[4] = {T(0), T(0), T(0), T(0)};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int n = 4 * q + j;
    l[j] = (n < a.N) ? (a.logits ? a.logits[(size_t)p * a.N + n] : T(1)) : T(0);
  }
  const int b_lo = blockIdx.z * rows_per_block, b_hi = min(a.B, b_lo + rows_per_block);
#if DR_K1_BWD_RACE
  // f32, in-kernel noise, tau = 1: y = exp(l + G - lse) with exp(G) = 1 / (-ln u) = -1 / (ln2 log2 u), so
  //   y dot = w_j [exp(lref - lse) dot / ln2] (-1 / log2 u),   w_j = exp(l_j - lref) constant over the rows:
  // per element convert, fma, ONE logarithm, a reciprocal and an fma -- the second logarithm, the add, the subtract, the
  // exponential and its scale of the general form (y = exp((l + G) - lse)) are gone; per row and thread one exponential.
  // lref = the thread's largest logit: lref - lse <= -G <= 3.2, no overflow; rows far above the thread's points flush to 0.
  const bool race = sizeof(T) == 4 && !a.gumbel && unit_tau;
  // (over the group's REAL points only: the padding value 0 of a tail group with N % 4 != 0 and logits far below zero would put
  //  lref above the row's lse by more than the bound and overflow the exponential -- round-3 advice)
  float lref = (float)l[0];   // 4 q < N for every thread that accumulates (q < groups)
#pragma unroll
  for (int j = 1; j < 4; ++j) lref = (4 * q + j < a.N) ? fmaxf(lref, (float)l[j]) : lref;
  float racc[4] = {0.f, 0.f, 0.f, 0.f};
#endif
  for (int b0 = b_lo; b0 < b_hi; b0 += 256) {
    const int nb = min(256, b_hi - b0);
    __syncthreads();
    // per row: lse and dot = sum_m y_m a_m over its k selected points.  Eight threads per row (one per selected point, kMaxK = 8),
    // 32 rows per sweep of the block: the k Philox calls, gathers and exponentials of a row run side by side instead of in one
    // thread while three of the block's four waves wait at the barrier (round 3: this prologue was as long as the 32-row main
    // loop it feeds)
    static_assert(kMaxK == 8, "the row prologue maps eight threads to a row");
    for (int rb = 0; rb < nb; rb += 32) {
      const int r = rb + ((int)threadIdx.x >> 3), j = (int)threadIdx.x & 7;
      T term = T(0), ls = T(0);
      if (r < nb) {
        const size_t row = (size_t)p * a.B + b0 + r;
        ls = lse[row];
        if (j < a.k) {
          const int i = idx[row * a.k + j];
          T av;
          if (a_sel) av = a_sel[row * a.k + j];
          else {
            const T *gs = grad_samples + (row * a.k + j) * 4, *m = matches4 + ((size_t)p * a.N + i) * 4;
            av = T(0);
#pragma unroll
            for (int d = 0; d < 4; ++d) av += gs[d] * m[d];
            if (grad_w) av += grad_w[row * a.k + j];
          }
          // y at the selected index is recomputed from (logit, noise): y = exp(g - lse)
          T nz;
          if (a.gumbel) nz = a.gumbel[row * a.N + i];
          else {
            uint32_t rr[4];
            Philox::gen(a.seed, (uint32_t)(i >> 2), (uint32_t)(b0 + r), (uint32_t)p, 0u, rr);
            const uint32_t w = (i & 2) ? ((i & 1) ? rr[3] : rr[2]) : ((i & 1) ? rr[1] : rr[0]);
            nz = gumbel_from_bits_t<T>(w);
          }
          const T li = a.logits ? a.logits[(size_t)p * a.N + i] : T(1);
          const T yi = exp_t<T>((unit_tau ? (li + nz) : (li + nz) / a.tau) - ls);
          term = yi * av;
          // the "+ y a" term exists only at the k selected points of a row: added here, once (by the first point-group
          // block), instead of comparing every point of the row against the k winners in the main loop
          if (blockIdx.x == 0) atomicAdd(grad_logits + (size_t)p * a.N + i, unit_tau ? term : term / a.tau);
        }
      }
      // dot of the row = the eight terms, summed in a fixed order (xor butterfly inside the group of eight lanes)
      term += __shfl_xor(term, 1, 64);
      term += __shfl_xor(term, 2, 64);
      term += __shfl_xor(term, 4, 64);
      if (r < nb && j == 0) {
        s_dot[r] = term;
        s_lse[r] = ls;
      }
    }
    __syncthreads();
#if DR_K1_BWD_RACE
    if (race) {
      if (q < groups) {
        for (int r = 0; r < nb; ++r) {
          uint32_t rr[4];
          Philox::gen(a.seed, (uint32_t)q, (uint32_t)(b0 + r), (uint32_t)p, 0u, rr);
          const float c = __expf(lref - (float)s_lse[r]) * ((float)s_dot[r] * 1.44269504088896340736f);
#pragma unroll
          for (int j = 0; j < 4; ++j) racc[j] = __builtin_fmaf(c, __builtin_amdgcn_rcpf(log2_uniform_from_bits(rr[j])), racc[j]);
        }
      }
      continue;
    }
#endif
    if (q < groups) {
      for (int r = 0; r < nb; ++r) {
        const int b = b0 + r;
        T nz[4];
        if (a.gumbel) {
#pragma unroll
          for (int j = 0; j < 4; ++j) nz[j] = (4 * q + j < a.N) ? a.gumbel[((size_t)p * a.B + b) * a.N + 4 * q + j] : T(0);
        } else {
          uint32_t rr[4];
          Philox::gen(a.seed, (uint32_t)q, (uint32_t)b, (uint32_t)p, 0u, rr);
#pragma unroll
          for (int j = 0; j < 4; ++j) nz[j] = gumbel_from_bits_t<T>(rr[j]);
        }
        const T ls = s_lse[r], dot = s_dot[r];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const T y = exp_t<T>((unit_tau ? (l[j] + nz[j]) : (l[j] + nz[j]) / a.tau) - ls);
          acc[j] -= y * dot;
        }
      }
    }
  }
#if DR_K1_BWD_RACE
  if (race) {
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] += (T)(__expf((float)l[j] - lref) * racc[j]);
  }
#endif
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (4 * q + j < a.N) atomicAdd(grad_logits + (size_t)p * a.N + 4 * q + j, acc[j] / a.tau);
}

// ---- K1 (inference variant): top-down sampling of the Gumbel top-k index set --------------------------------------
// The k largest of {logit_n + G_n}, G_n iid Gumbel(0,1), taken in decreasing order, are distributed as k sequential
// draws WITHOUT replacement from softmax(logits) (Plackett-Luce; the "Gumbel-max trick" read backwards, Kool et al. 2019,
// Maddison et al. 2014).  When only the index set is consumed -- RANSAC test mode, ransac.py:65: `samples != 0` -- the
// N noise values of a row are never needed: k uniforms and k binary searches in the cumulative weights of the pair do.
// O(B k log N) instead of O(B N); tau > 0 does not change the order, so it does not appear.  y_sel / lse / the dense
// outputs (train mode, weighted mode) need the whole noise row and stay with gumbel_topk_kernel.
// Step 1, per pair: cdf[n] = sum_{m <= n} exp(logit_m - max logit) in f64 (one block per pair).
template <typename T>
__global__ __launch_bounds__(1024) void softmax_cdf_kernel(const T *__restrict__ logits, int N, double *__restrict__ cdf) {
  __shared__ double s_part[1024];
  __shared__ double s_max;
  const int p = blockIdx.x, tid = threadIdx.x;
  const T *l = logits ? logits + (size_t)p * N : nullptr;
  double *c = cdf + (size_t)p * N;
  double mx = -INFINITY;
  for (int n = tid; n < N; n += 1024) mx = fmax(mx, l ? (double)l[n] : 1.0);
  s_part[tid] = mx;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if (tid < o) s_part[tid] = fmax(s_part[tid], s_part[tid + o]);
    __syncthreads();
  }
  if (tid == 0) s_max = s_part[0];
  __syncthreads();
  mx = s_max;
  // contiguous chunk per thread, then an exclusive scan of the chunk sums
  const int per = (N + 1023) / 1024, n0 = tid * per, n1 = min(N, n0 + per);
  double acc = 0;
  for (int n = n0; n < n1; ++n) acc += exp((l ? (double)l[n] : 1.0) - mx);
  s_part[tid] = acc;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {   // Hillis-Steele inclusive scan
    const double v = tid >= o ? s_part[tid - o] : 0.0;
    __syncthreads();
    s_part[tid] += v;
    __syncthreads();
  }
  double run = s_part[tid] - acc;
  for (int n = n0; n < n1; ++n) {
    run += exp((l ? (double)l[n] : 1.0) - mx);
    c[n] = run;
  }
}

// Step 2, one lane per (pair, hypothesis): k draws without replacement, ascending output.
// Philox4x32-7(key = seed, counter = (draw pair, b, p, 2)): 64 random bits per draw.
__global__ __launch_bounds__(256) void topdown_sample_kernel(const double *__restrict__ cdf, uint64_t seed, int B, int N, int k,
                                                            int32_t *__restrict__ idx, const uint64_t *__restrict__ seed_ptr) {
  if (seed_ptr) seed = *seed_ptr;
  const int p = blockIdx.y, b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const double *c = cdf + (size_t)p * N;
  const double total = c[N - 1];
  int sel[kMaxK];
  double wsel[kMaxK];
  double removed = 0;
  uint32_t r[4];
  for (int d = 0; d < k; ++d) {
    if ((d & 1) == 0) Philox::gen(seed, (uint32_t)(d >> 1), (uint32_t)b, (uint32_t)p, 2u, r);
    const uint64_t bits = ((uint64_t)r[2 * (d & 1)] << 32) | r[2 * (d & 1) + 1];
    const double u = ((double)(bits >> 11) + 0.5) * (1.0 / 9007199254740992.0);   // (0,1), 53 bits
    const double target = u * (total - removed);
    // smallest n with  cdf[n] - (weight already removed at positions <= n)  >  target
    int lo = 0, hi = N - 1;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      double v = c[mid];
      for (int j = 0; j < d; ++j) v -= (sel[j] <= mid) ? wsel[j] : 0.0;
      if (v > target) hi = mid; else lo = mid + 1;
    }
    // rounding can land on an already-selected point (its remaining weight is 0): move to the next free one
    bool dup = true;
    while (dup) {
      dup = false;
      for (int j = 0; j < d; ++j) dup = dup || (sel[j] == lo);
      if (dup) lo = (lo + 1 < N) ? lo + 1 : 0;
    }
    sel[d] = lo;
    wsel[d] = c[lo] - (lo > 0 ? c[lo - 1] : 0.0);
    removed += wsel[d];
  }
  // insertion sort (k <= 8), ascending point index: the order `points[samples != 0]` yields (ransac.py:65)
  for (int i = 1; i < k; ++i) {
    const int v = sel[i];
    int j = i - 1;
    while (j >= 0 && sel[j] > v) { sel[j + 1] = sel[j]; --j; }
    sel[j + 1] = v;
  }
  for (int d = 0; d < k; ++d) idx[((size_t)p * B + b) * k + d] = sel[d];
}

// ---- K1u: uniform indices in [0, N-2]
__global__ void uniform_sample_kernel(uint64_t seed, int B, int k, int N, int32_t *__restrict__ idx,
                                      const uint64_t *__restrict__ seed_ptr) {
  if (seed_ptr) seed = *seed_ptr;
  const int p = blockIdx.y;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;  // over B*k
  if (t >= B * k) return;
  const int b = t / k, j = t % k;
  uint32_t r[4];
  Philox::gen(seed, (uint32_t)j, (uint32_t)b, (uint32_t)p, 1u, r);
  const uint32_t span = (uint32_t)max(N - 1, 1);
  idx[((size_t)p * B + b) * k + j] = (int32_t)(((uint64_t)r[0] * span) >> 32);
}

// ---- seed of the next call, on the device: state = (base, calls) -> out = base * 0x9E3779B97F4A7C15 + calls ; calls += 1
// (what the drivers compute on the host per call; here a captured graph advances it by itself at every replay)
__global__ void seed_next_kernel(uint64_t *__restrict__ state, uint64_t *__restrict__ out, int n = 1) {
  // n > 1: the seeds of the next n calls at once (a multi-round call draws one per batch: one launch instead of n)
  if (blockIdx.x == 0) seed_next_block(state, out, n);
}

// ---- K2 gather forward / backward
template <typename T>
__global__ void gather_fwd_kernel(const T *__restrict__ matches, const int32_t *__restrict__ idx,
                                  const T *__restrict__ y_sel, int N, int BK, int c, T *__restrict__ samples) {
  const int p = blockIdx.y;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;  // over B*k
  if (t >= BK) return;
  const size_t e = (size_t)p * BK + t;
  const int i = idx[e];
  T st = T(1);
  if (y_sel) { const T y = y_sel[e]; st = (T(1) - y) + y; }
  const T *src = matches + ((size_t)p * N + i) * c;
  T *dst = samples + e * c;
  for (int d = 0; d < c; ++d) dst[d] = src[d] * st;
}

template <typename T>
__global__ void gather_bwd_kernel(const T *__restrict__ matches, const int32_t *__restrict__ idx,
                                  const T *__restrict__ y_sel, const T *__restrict__ grad_samples,
                                  const T *__restrict__ grad_w, int N, int BK, int c, T *__restrict__ a_sel,
                                  T *__restrict__ grad_matches) {
  const int p = blockIdx.y;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= BK) return;
  const size_t e = (size_t)p * BK + t;
  const int i = idx[e];
  T st = T(1);
  if (y_sel) { const T y = y_sel[e]; st = (T(1) - y) + y; }
  const T *m = matches + ((size_t)p * N + i) * c;
  const T *g = grad_samples + e * c;
  T dot = T(0);
  for (int d = 0; d < c; ++d) dot += g[d] * m[d];
  if (grad_w) dot += grad_w[e];
  a_sel[e] = dot;
  if (grad_matches)
    for (int d = 0; d < c; ++d) atomicAdd(grad_matches + ((size_t)p * N + i) * c + d, g[d] * st);
}

}  // namespace dr

template <typename T>
static int gumbel_bwd_impl(const T *logits, const T *gumbel, uint64_t seed, const uint64_t *seed_ptr, T tau,
                           int P, int B, int N, int k, const int32_t *idx, const T *lse, const T *a_sel,
                           T *grad_logits, void *stream, const T *grad_samples = nullptr, const T *grad_w = nullptr,
                           const T *matches4 = nullptr) {
  DR_REQUIRE(idx && lse && grad_logits && (a_sel || (grad_samples && matches4)), "null pointer");
  DR_REQUIRE(P > 0 && B > 0 && N > 0 && P <= 65535 && k >= 1 && k <= dr::kMaxK && tau > 0, "bad sizes");
  dr::GumbelArgs<T> a{logits, gumbel, seed, tau, P, B, N, k, seed_ptr};
  const size_t smem = sizeof(T) * 256 * 2;
  const int gx = ((N + 3) / 4 + 255) / 256;
  // enough row chunks to put >= ~2048 blocks on the chip, at least 32 rows each
#ifndef DR_K1_BWD_BLOCKS
#define DR_K1_BWD_BLOCKS 2048
#endif
#ifndef DR_K1_BWD_MINROWS
#define DR_K1_BWD_MINROWS 32
#endif
  int chunks = (int)std::min<long>((B + DR_K1_BWD_MINROWS - 1) / DR_K1_BWD_MINROWS,
                                   std::max<long>(1, DR_K1_BWD_BLOCKS / std::max<long>(1, (long)gx * P)));
  const int rows_per_block = (B + chunks - 1) / chunks;
  chunks = (B + rows_per_block - 1) / rows_per_block;
  if (hipMemsetAsync(grad_logits, 0, sizeof(T) * (size_t)P * N, (hipStream_t)stream) != hipSuccess)
    return dr::check_launch("memset");
  hipLaunchKernelGGL((dr::gumbel_bwd_kernel<T>), dim3(gx, P, chunks), dim3(256), smem, (hipStream_t)stream, a, idx,
                     lse, a_sel, grad_logits, rows_per_block, grad_samples, grad_w, matches4);
  return dr::check_launch("gumbel_bwd_kernel");
}
static int gumbel_bwd_f32_impl(const float *logits, const float *gumbel, uint64_t seed, const uint64_t *seed_ptr, float tau,
                               int P, int B, int N, int k, const int32_t *idx, const float *lse, const float *a_sel,
                               float *grad_logits, void *stream) {
  return gumbel_bwd_impl<float>(logits, gumbel, seed, seed_ptr, tau, P, B, N, k, idx, lse, a_sel, grad_logits, stream);
}

template <typename T>
static int topdown_impl(const T *logits, uint64_t seed, const uint64_t *seed_ptr, int P, int B, int N, int k, double *cdf_ws,
                        int32_t *idx, void *stream) {
  DR_REQUIRE(P > 0 && B > 0 && N > 0 && k > 0 && k <= dr::kMaxK && k <= N && P <= 65535, "bad sizes");
  DR_REQUIRE(cdf_ws && idx, "null pointer");
  hipLaunchKernelGGL((dr::softmax_cdf_kernel<T>), dim3(P), dim3(1024), 0, (hipStream_t)stream, logits, N, cdf_ws);
  if (int rc = dr::check_launch("softmax_cdf_kernel")) return rc;
  hipLaunchKernelGGL(dr::topdown_sample_kernel, dim3((B + 255) / 256, P), dim3(256), 0, (hipStream_t)stream, cdf_ws, seed, B,
                     N, k, idx, seed_ptr);
  return dr::check_launch("topdown_sample_kernel");
}

static int uniform_impl(uint64_t seed, const uint64_t *seed_ptr, int P, int B, int k, int N, int32_t *idx, void *stream) {
  DR_REQUIRE(idx, "null pointer");
  DR_REQUIRE(P > 0 && B > 0 && k > 0 && N > 1 && P <= 65535, "bad sizes");
  hipLaunchKernelGGL(dr::uniform_sample_kernel, dim3((B * k + 255) / 256, P), dim3(256), 0, (hipStream_t)stream, seed,
                     B, k, N, idx, seed_ptr);
  return dr::check_launch("uniform_sample_kernel");
}

extern "C" {

#define DR_GUMBEL_CHECK()                                                          \
  DR_REQUIRE(idx, "null output pointer");                                          \
  DR_REQUIRE((y_sel != nullptr) == (lse != nullptr), "y_sel and lse: pass both or neither (index sets only)"); \
  DR_REQUIRE(y_sel || (!y_soft && !ret), "the dense outputs need y_sel and lse");    \
  DR_REQUIRE(P > 0 && B > 0 && N > 0 && P <= 65535, "bad sizes");                  \
  DR_REQUIRE(k >= 1 && k <= dr::kMaxK && k <= N, "k must be in [1, 8] and <= N");  \
  DR_REQUIRE(tau > 0, "tau must be positive")

// seed_dev (every sampler entry, round 6: one entry per sampler instead of a `_dseed` twin): != NULL = the Philox key is read from
// device memory when the kernel starts -- what a captured graph needs (a by-value seed would be frozen into it; dr_seed_next_n
// advances such a seed on the device); it serves the in-kernel noise of given logits (no explicit noise, no dense outputs)
#define DR_SEED_DEV_CHECK()                                                                                              \
  DR_REQUIRE(!seed_dev || (logits && !gumbel && !y_soft && !ret && !gumbel_out), "a device seed serves the in-kernel noise of given logits only")
int dr_gumbel_topk_fwd_f32(const float *logits, const float *gumbel, uint64_t seed, const uint64_t *seed_dev, float tau, int P, int B,
                           int N, int k, int32_t *idx, float *y_sel, float *lse, float *y_soft, float *ret,
                           float *gumbel_out, void *stream) {
  DR_GUMBEL_CHECK();
  DR_SEED_DEV_CHECK();
  return dr::gumbel_fwd_launch<float>(logits, gumbel, seed, tau, P, B, N, k, idx, y_sel, lse, y_soft, ret, gumbel_out,
                                      (hipStream_t)stream, seed_dev);
}

int dr_gumbel_topk_fwd_f64(const double *logits, const double *gumbel, uint64_t seed, const uint64_t *seed_dev, double tau, int P,
                           int B, int N, int k, int32_t *idx, double *y_sel, double *lse, double *y_soft, double *ret,
                           double *gumbel_out, void *stream) {
  DR_GUMBEL_CHECK();
  DR_SEED_DEV_CHECK();
  return dr::gumbel_fwd_launch<double>(logits, gumbel, seed, tau, P, B, N, k, idx, y_sel, lse, y_soft, ret,
                                       gumbel_out, (hipStream_t)stream, seed_dev);
}

int dr_gumbel_topk_bwd_f32(const float *logits, const float *gumbel, uint64_t seed, const uint64_t *seed_dev, float tau, int P, int B,
                           int N, int k, const int32_t *idx, const float *lse, const float *a_sel, float *grad_logits,
                           void *stream) {
  DR_REQUIRE(!seed_dev || (logits && !gumbel), "a device seed serves the in-kernel noise of given logits only");
  return gumbel_bwd_f32_impl(logits, gumbel, seed, seed_dev, tau, P, B, N, k, idx, lse, a_sel, grad_logits, stream);
}

// f64 (`-pr 2 -tr 1`, model_cl.py:164-169): the same kernel in double (general form: two logarithms + an exponential per element)
int dr_gumbel_topk_bwd_f64(const double *logits, const double *gumbel, uint64_t seed, double tau, int P, int B, int N,
                           int k, const int32_t *idx, const double *lse, const double *a_sel, double *grad_logits,
                           void *stream) {
  return gumbel_bwd_impl<double>(logits, gumbel, seed, nullptr, tau, P, B, N, k, idx, lse, a_sel, grad_logits, stream);
}

int dr_topdown_sample_f32(const float *logits, uint64_t seed, const uint64_t *seed_dev, int P, int B, int N, int k, double *cdf_ws,
                          int32_t *idx, void *stream) {
  return topdown_impl<float>(logits, seed, seed_dev, P, B, N, k, cdf_ws, idx, stream);
}

int dr_topdown_sample_f64(const double *logits, uint64_t seed, int P, int B, int N, int k, double *cdf_ws, int32_t *idx,
                          void *stream) {
  return topdown_impl<double>(logits, seed, nullptr, P, B, N, k, cdf_ws, idx, stream);
}

int dr_uniform_sample(uint64_t seed, const uint64_t *seed_dev, int P, int B, int k, int N, int32_t *idx, void *stream) {
  return uniform_impl(seed, seed_dev, P, B, k, N, idx, stream);
}

// ---- the same samplers with the Philox key read from device memory (*seed_dev) when the kernel starts: what a captured
//      graph needs (a by-value seed would be frozen into the graph); dr_seed_next advances such a seed on the device
int dr_seed_next_n(uint64_t *state, uint64_t *seeds_out, int n, void *stream) {
  DR_REQUIRE(state && seeds_out, "null pointer");
  DR_REQUIRE(n >= 1 && n <= 65536, "1 <= n <= 65536 seeds per launch");
  hipLaunchKernelGGL(dr::seed_next_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, state, seeds_out, n);
  return dr::check_launch("seed_next_kernel");
}

// K1 (index sets only) + K2 in one call: idx [P,B,k] and samples [P,B,k,4] = matches[p, idx] (test mode: the points themselves,
// ransac.py:65).  One launch when the register kernel serves the shape, sampler + gather launches otherwise.
static int gumbel_topk_gather_impl(const float *logits, const float *matches, uint64_t seed, const uint64_t *seed_dev, float tau, int P,
                                   int B, int N, int k, int32_t *idx, float *samples, dr::PairGate gate, void *stream,
                                   uint32_t *screen_ws = nullptr, int sub = 0, float *race_ws = nullptr, bool race_ready = false) {
  const float *y_sel = nullptr, *lse = nullptr, *y_soft = nullptr, *ret = nullptr;
  DR_REQUIRE(logits && matches && samples, "null pointer");
  DR_REQUIRE((reinterpret_cast<uintptr_t>(matches) & 15) == 0 && (reinterpret_cast<uintptr_t>(samples) & 15) == 0, "16-byte alignment");
  DR_GUMBEL_CHECK();
  bool gathered = false;
  if (int rc = dr::gumbel_fwd_launch<float>(logits, nullptr, seed, tau, P, B, N, k, idx, nullptr, nullptr, nullptr, nullptr, nullptr,
                                            (hipStream_t)stream, seed_dev, reinterpret_cast<const float4 *>(matches),
                                            reinterpret_cast<float4 *>(samples), &gathered, screen_ws, gate, sub, race_ws, race_ready))
    return rc;
  if (gathered) return 0;
  hipLaunchKernelGGL((dr::gather_fwd_kernel<float>), dim3((B * k + 255) / 256, P), dim3(256), 0, (hipStream_t)stream, matches, idx,
                     (const float *)nullptr, N, B * k, 4, samples);
  return dr::check_launch("gather_fwd_kernel");
}

// Train mode (round 5): K1 with the soft-max statistics + K2 in one call -- idx, y_sel [P,B,k], lse [P,B] and samples [P,B,k,4] =
// matches[p, idx] * ((1 - y_sel) + y_sel), the straight-through weights of gumbel_sampler.py:36-40 applied as ransac.py:58-65 does.
// One launch when the register kernel serves the shape (N <= 2048, N % 4 == 0, tau = 1), sampler + gather launches otherwise.
// race_ws (round 6; optional, (N + 32) * P floats, 16-byte aligned): the one-logarithm form of the register kernel in train mode --
// keys, winners and soft-max statistics from ONE logarithm and one reciprocal per element (weights by a prologue launch); y_sel / lse
// agree with the two-logarithm form to rounding, the index sets up to the rounding of near-ties.
int dr_gumbel_topk_gather_soft_f32(const float *logits, const float *matches, uint64_t seed, const uint64_t *seed_dev, float tau,
                                   int P, int B, int N, int k, int32_t *idx, float *y_sel, float *lse, float *samples, float *race_ws,
                                   void *stream) {
  const float *y_soft = nullptr, *ret = nullptr;
  DR_REQUIRE(logits && matches && samples && y_sel && lse, "null pointer");
  DR_REQUIRE((reinterpret_cast<uintptr_t>(matches) & 15) == 0 && (reinterpret_cast<uintptr_t>(samples) & 15) == 0, "16-byte alignment");
  DR_REQUIRE((reinterpret_cast<uintptr_t>(race_ws) & 15) == 0, "workspace alignment");
  DR_GUMBEL_CHECK();
  bool gathered = false;
  if (int rc = dr::gumbel_fwd_launch<float>(logits, nullptr, seed, tau, P, B, N, k, idx, y_sel, lse, nullptr, nullptr, nullptr,
                                            (hipStream_t)stream, seed_dev, reinterpret_cast<const float4 *>(matches),
                                            reinterpret_cast<float4 *>(samples), &gathered, nullptr, dr::PairGate(), 0, race_ws, false))
    return rc;
  if (gathered) return 0;
  hipLaunchKernelGGL((dr::gather_fwd_kernel<float>), dim3((B * k + 255) / 256, P), dim3(256), 0, (hipStream_t)stream, matches, idx,
                     (const float *)y_sel, N, B * k, 4, samples);
  return dr::check_launch("gather_fwd_kernel");
}

// ... and its backward in one launch: grad_logits [P,N] from grad_samples [P,B,k,4] (and grad_w [P,B,k] | null, the gradient of the
// y_sel output) -- dr_gather_bwd's a_sel is formed inside dr_gumbel_topk_bwd's row prologue (no [P,B,k] tensor, no second launch).
// The correspondences get no gradient here (dr_gather_bwd_f32 serves callers that want one).
int dr_gumbel_topk_gather_bwd_f32(const float *logits, const float *matches, uint64_t seed, const uint64_t *seed_dev, float tau,
                                  int P, int B, int N, int k, const int32_t *idx, const float *lse, const float *grad_samples,
                                  const float *grad_w, float *grad_logits, void *stream) {
  DR_REQUIRE(logits && matches && grad_samples, "null pointer");
  return gumbel_bwd_impl<float>(logits, nullptr, seed, seed_dev, tau, P, B, N, k, idx, lse, nullptr, grad_logits, stream,
                                grad_samples, grad_w, matches);
}

// THE test-mode sampler entry (round 6: `_gated` folded in).  gate_iters / gate_max_iters (optional: a round > 1 of a multi-round
// call): pairs whose iteration counter has reached its bound are skipped (gate_iters [P] int32, gate_max_iters [P] f64: the state
// dr_ransac_update keeps; the rows of a skipped pair keep their contents).
// Only the register-resident kernel (N <= 2048, N % 4 == 0, tau = 1) looks at the gate; other shapes simply run.
// screen_ws (optional, (N + 32) * P words, 16-byte aligned; round 5): short rows (N <= 2048, N % 4 == 0, tau = 1, k <= 5, B >= 64)
// then take the SCREENED register kernel -- per point the smallest Philox word that can lift it to the score logsumexp - ln(11 + k),
// only the ~16 points of a row that pass are evaluated; the index sets are those of the unscreened kernel, bit for bit.
// sub (round 6): > 0 = the B rows are consecutive sub-batches of `sub` rows -- row b draws what row b % sub of a call with the seed
// (seed | *seed_dev) + b / sub draws: one launch samples what ceil(B / sub) calls of a batch-by-batch loop sample (dr_ransac_update's
// `sub_models` walks them in order); 0 = one batch.  Soft (train-mode) outputs have no sub-batch form.
int dr_gumbel_topk_gather_f32(const float *logits, const float *matches, uint64_t seed, const uint64_t *seed_dev, float tau,
                              int P, int B, int N, int k, int32_t *idx, float *samples, uint32_t *screen_ws,
                              const int32_t *gate_iters, const double *gate_max_iters, int sub, float *race_ws, int race_ready,
                              void *stream) {
  DR_REQUIRE((gate_iters == nullptr) == (gate_max_iters == nullptr), "gate: both pointers or neither");
  DR_REQUIRE(sub >= 0, "sub-batch size");
  DR_REQUIRE(!(race_ws && screen_ws), "one workspace: the screened or the one-logarithm form");
  DR_REQUIRE((reinterpret_cast<uintptr_t>(race_ws) & 15) == 0, "workspace alignment");
  DR_REQUIRE((reinterpret_cast<uintptr_t>(screen_ws) & 15) == 0, "workspace alignment");
  dr::PairGate gate;
  gate.iters = gate_iters;
  gate.max_iters = gate_max_iters;
  return gumbel_topk_gather_impl(logits, matches, seed, seed_dev, tau, P, B, N, k, idx, samples, gate, stream, screen_ws, sub, race_ws,
                                 race_ready != 0);
}

// K1, index sets only, in-kernel noise, with an optional screening workspace ((N + 32) * P words, 16-byte aligned): long rows
// (N > 2048, N % 4 == 0, tau == 1, k <= 5) then skip the Gumbel transform of every step of a wave that holds no candidate
int dr_gumbel_topk_index_f32(const float *logits, uint64_t seed, const uint64_t *seed_dev, float tau, int P, int B, int N, int k,
                             int32_t *idx, uint32_t *screen_ws, void *stream) {
  const float *y_sel = nullptr, *lse = nullptr, *y_soft = nullptr, *ret = nullptr;
  DR_REQUIRE(logits, "null pointer");
  DR_REQUIRE((reinterpret_cast<uintptr_t>(screen_ws) & 15) == 0, "workspace alignment");
  DR_GUMBEL_CHECK();
  return dr::gumbel_fwd_launch<float>(logits, nullptr, seed, tau, P, B, N, k, idx, nullptr, nullptr, nullptr, nullptr, nullptr,
                                      (hipStream_t)stream, seed_dev, nullptr, nullptr, nullptr, screen_ws);
}

int dr_gather_fwd_f32(const float *matches, const int32_t *idx, const float *y_sel, int P, int N, int B, int k,
                      int c, float *samples, void *stream) {
  DR_REQUIRE(matches && idx && samples, "null pointer");
  DR_REQUIRE(P > 0 && N > 0 && B > 0 && k > 0 && c > 0 && P <= 65535, "bad sizes");
  hipLaunchKernelGGL((dr::gather_fwd_kernel<float>), dim3((B * k + 255) / 256, P), dim3(256), 0, (hipStream_t)stream,
                     matches, idx, y_sel, N, B * k, c, samples);
  return dr::check_launch("gather_fwd_kernel");
}

int dr_gather_fwd_f64(const double *matches, const int32_t *idx, const double *y_sel, int P, int N, int B, int k,
                      int c, double *samples, void *stream) {
  DR_REQUIRE(matches && idx && samples, "null pointer");
  DR_REQUIRE(P > 0 && N > 0 && B > 0 && k > 0 && c > 0 && P <= 65535, "bad sizes");
  hipLaunchKernelGGL((dr::gather_fwd_kernel<double>), dim3((B * k + 255) / 256, P), dim3(256), 0,
                     (hipStream_t)stream, matches, idx, y_sel, N, B * k, c, samples);
  return dr::check_launch("gather_fwd_kernel");
}

int dr_gather_bwd_f32(const float *matches, const int32_t *idx, const float *y_sel, const float *grad_samples,
                      const float *grad_w, int P, int N, int B, int k, int c, float *a_sel, float *grad_matches,
                      void *stream) {
  DR_REQUIRE(matches && idx && grad_samples && a_sel, "null pointer");
  DR_REQUIRE(P > 0 && N > 0 && B > 0 && k > 0 && c > 0 && P <= 65535, "bad sizes");
  hipLaunchKernelGGL((dr::gather_bwd_kernel<float>), dim3((B * k + 255) / 256, P), dim3(256), 0, (hipStream_t)stream,
                     matches, idx, y_sel, grad_samples, grad_w, N, B * k, c, a_sel, grad_matches);
  return dr::check_launch("gather_bwd_kernel");
}

int dr_gather_bwd_f64(const double *matches, const int32_t *idx, const double *y_sel, const double *grad_samples,
                      const double *grad_w, int P, int N, int B, int k, int c, double *a_sel, double *grad_matches,
                      void *stream) {
  DR_REQUIRE(matches && idx && grad_samples && a_sel, "null pointer");
  DR_REQUIRE(P > 0 && N > 0 && B > 0 && k > 0 && c > 0 && P <= 65535, "bad sizes");
  hipLaunchKernelGGL((dr::gather_bwd_kernel<double>), dim3((B * k + 255) / 256, P), dim3(256), 0, (hipStream_t)stream,
                     matches, idx, y_sel, grad_samples, grad_w, N, B * k, c, a_sel, grad_matches);
  return dr::check_launch("gather_bwd_kernel");
}

}  // extern "C"
