// K4, benchmark-shape path -- MSAC scoring (scorings/msac_score.py:12-55) as "matrix-core filter + exact evaluation of the
// candidates + line-aligned mask stream".  Used for f32, N % 16 == 0, N <= 2048, enough (pair x model) work to fill the
// chip (msac_score.hip keeps the general kernels and chooses).
//
// Why.  Of the 39 flops per (model, point) only the sign of  d2/thr2 - 1  and, for the few inliers, its value are wanted:
// an outlier contributes EXACTLY zero to the score and a zero mask byte.  At the benchmark shape 97 % of the evaluations are
// outliers.  So:
//   1. FILTER (matrix cores).  r = x2^T M x1 is bilinear and J = a0^2 + a1^2 + b0^2 + b1^2 is a quadratic form of the point:
//      both are dot products between a per-MODEL vector and a per-POINT feature vector.  Every f32 value is split into two
//      f16 values (hi + lo: 22 significant bits); r uses the four products hh, hl, lh, ll of its 8 variable terms (K = 32, the
//      constant term rides in the accumulator input), J the three products hh, hl, lh of its 10 monomials plus a per-point
//      slack H (K = 32).  Two v_mfma_f32_16x16x32_f16 per (16 points x 16 models).  In normalised units (coordinates scaled
//      by 2^-s to <= 1, model by 2^e to max |m| in [0.5, 1), everything by 2^g with 2^g * theta' in [1, 2))
//            candidate  <=>  rt^2 <= (1 + 1/16) Theta^2 J + H_p ,      H_p = 17 E_p^2 + E^J_p
//      where E_p / E^J_p bound the error of rt / Jt (split representation, f32 accumulation, the exact chain's own rounding);
//      since 2 a b <= eps a^2 + b^2 / eps this contains |rt| <= Theta sqrt(J) + E_p, hence every point the exact f32 chain
//      below calls an inlier (scratch/k4_filter_emul.py checks the inequality on the CPU, degenerate inputs included).
//      Models or thresholds outside the range where those bounds hold run in "all candidates" mode (r operand 0, J = +big).
//   2. EXACT.  Candidates go, as (model, four consecutive points) entries, into a wave-private LDS queue and are evaluated by
//      the same f32 fma chain as the general kernel (sampson_s): masks are bit-identical to it by construction.  The soft
//      score is accumulated as a 64-bit fixed-point integer (2^-29 quantum) with LDS atomics: order-independent, so scores
//      are reproducible run to run and independent of which slots are valid.
//   3. MASK STREAM.  A block assembles the mask of 16 consecutive model slots x all N points in LDS (32 000 B at N = 2000 = the
//      flat image of that [16, N] piece of the output, 250 whole cache lines) and streams it out with 16-byte stores, double
//      buffered, one block barrier per 16 slots.  Zero rows of invalid slots cost nothing extra.  Store-pattern benchmark
//      (round 1): contiguous 16-row flushes 5.5 TB/s against 3.96 TB/s for row-by-row 1 KiB pieces.
//
// Mapping.  512-thread block (8 waves, one block per CU, <= 256 VGPRs), persistent over a contiguous range of 16-slot chunks
// of ONE pair.  A wave owns 256 points for the block's lifetime: their matrix-core operand fragments (2 x 4 VGPRs per
// 16-point tile) are built once in the prologue and stay in registers; the model-side operands of the next chunk are
// prepared by 16 lanes of one (rotating) wave while the current one is filtered.
//
// Algorithmic bytes are those of the general kernel (16N + 36M + 4M + M*N per pair).
#include "dr_common.hpp"
#include "msac_filter.hpp"

// stage timing for profiling builds: accumulated in (scalar) registers, written once per wave at the end -- the generic
// DR_STAGE macro's global atomic per stage perturbs a loop this short beyond use
#ifdef DR_PROFILE_STAGES
#define KF_STAGE_BEGIN() unsigned long long kf_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long kf_prev = __builtin_readcyclecounter()
#define KF_STAGE(i) do { const unsigned long long kf_now = __builtin_readcyclecounter(); kf_acc[i] += kf_now - kf_prev; kf_prev = kf_now; } while (0)
#define KF_STAGE_END() do { if ((threadIdx.x & 63) == 0) { for (int i_ = 0; i_ < 8; ++i_) atomicAdd(&::dr::g_stage_cycles[i_], kf_acc[i_]); \
                                 atomicAdd(&::dr::g_stage_cycles[8 + (threadIdx.x >> 6)], kf_acc[0] + kf_acc[1] + kf_acc[2] + kf_acc[3] + kf_acc[5]); } } while (0)
#else
#define KF_STAGE_BEGIN() do {} while (0)
#define KF_STAGE(i) do {} while (0)
#define KF_STAGE_END() do {} while (0)
#endif

namespace dr {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef uint32_t u4 __attribute__((ext_vector_type(4)));

#ifndef DR_KF_STORE
#define DR_KF_STORE 1       // cache policy of the mask stores: 0 plain, 1 nt, 2 sc1, 3 sc0 sc1, 4 nt sc1, 5 nt sc0 sc1
#endif                      // (write-once stream, never re-read by this kernel: nt measured 128-140 us for the stream alone
                            // against 207-222 us with plain stores)
#ifndef DR_KF_STOREONLY
#define DR_KF_STOREONLY 0   // 1: timing decomposition only -- no filter, no candidates: the mask stream alone (all-zero masks)
#endif
#ifndef DR_KF_RAWBAR
#define DR_KF_RAWBAR 1      // 1: the per-chunk block barrier orders LDS only (s_waitcnt lgkmcnt(0) + s_barrier).  __syncthreads()
                            // also waits for the wave's outstanding global stores (vmcnt(0)): the mask stream of chunk c would
                            // have to be acknowledged by memory before chunk c + 1 may pass its barrier.  No thread of the block
                            // ever reads global memory another thread wrote, so LDS ordering is all the barrier has to give.
#endif
#ifndef DR_KF_SKIP
#define DR_KF_SKIP 0        // timing decomposition only (wrong results): 1 no consumer, 2 no matrix-core filter (and so no
#endif                      // consumer), 4 no per-chunk operand preparation; bits may be combined
#ifndef DR_KF_GROUP
#define DR_KF_GROUP 4       // tiles whose matrix instructions are issued before their compares
#endif

constexpr int kFT = 512, kFW = kFT / 64, kFSlots = 16, kFMaxN = 2048, kFTilesW = 16;
constexpr int kStagePitch = 72;   // halves per staged point row (64 used; 144 B pitch: conflict-free 16-byte reads)
constexpr float kFEps = 1.0f / 16.0f;
constexpr float kFKappaR = 5e-6f, kFEabs = 4e-3f, kFKappaJ = 4e-6f;
constexpr float kFBig = 60000.0f;

struct FilterShared {
  // [0, 65536): two mask buffers; [65536, 98304): the eight wave-private candidate queues.  The prologue stages the
  // point-side operand rows (512 x 144 B) over the same bytes.
  alignas(16) unsigned char raw[98304];
  alignas(16) float4 pts[kFMaxN];
  alignas(16) _Float16 opR[2][kFSlots][32];
  alignas(16) _Float16 opJ[2][kFSlots][32];
  alignas(16) float mraw[2][kFSlots][12];
  float cin[2][kFSlots][2];
  int flags[2][kFSlots];
  unsigned long long acc[2][kFSlots];
  float red[kFW];
};

__device__ __forceinline__ void block_sync_lds() {
#if DR_KF_RAWBAR
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#else
  __syncthreads();
#endif
}

__device__ __forceinline__ void mask_store(u4 *dst, u4 v) {
#if DR_KF_STORE == 0
  *dst = v;
#elif DR_KF_STORE == 1
  __builtin_nontemporal_store(v, dst);
#elif DR_KF_STORE == 2
  asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst), "v"(v) : "memory");
#elif DR_KF_STORE == 3
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(dst), "v"(v) : "memory");
#elif DR_KF_STORE == 4
  asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" ::"v"(dst), "v"(v) : "memory");
#else
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(dst), "v"(v) : "memory");
#endif
}

__device__ __forceinline__ void split2(float v, _Float16 &h, _Float16 &l) {
  h = (_Float16)v;
  l = (_Float16)(v - (float)h);
}

// the exact chain of the general kernel (msac_score.hip: sampson_s / msac_eval16), same association and fma contraction
__device__ __forceinline__ float sampson_exact(const float (&m)[9], float x1, float y1, float x2, float y2, float inv_thr2) {
  const float a0 = fmaf(x2, m[0], fmaf(y2, m[3], m[6]));
  const float a1 = fmaf(x2, m[1], fmaf(y2, m[4], m[7]));
  const float a2 = fmaf(x2, m[2], fmaf(y2, m[5], m[8]));
  const float b0 = fmaf(x1, m[0], fmaf(y1, m[1], m[2]));
  const float b1 = fmaf(x1, m[3], fmaf(y1, m[4], m[5]));
  const float r = fmaf(x1, a0, fmaf(y1, a1, a2));
  const float jj = fmaf(a0, a0, fmaf(a1, a1, fmaf(b0, b0, b1 * b1)));
  const float d2 = (r * r) * __builtin_amdgcn_rcpf(jj);
  return fmaf(d2, inv_thr2, -1.0f);
}

__global__ __launch_bounds__(kFT) void msac_filter_kernel(const float *__restrict__ matches, const float *__restrict__ models,
                                                          const uint8_t *__restrict__ valid, const float *__restrict__ thr,
                                                          int M, int N, float *__restrict__ scores,
                                                          uint8_t *__restrict__ masks) {
  __shared__ FilterShared sh;
  const int p = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: branches on it stay on the scalar unit
  const int row = lane & 15, kg = lane >> 4;
  const int T = N >> 4;                                   // 16-point tiles (N % 16 == 0)
  // wave w owns the tiles 8 t + w (t = 0..15): interleaved, so that spatially sorted inliers (the synthetic pairs keep
  // them in the second half of the point list) load the eight waves evenly
  const int ntw = (T > w) ? min(kFTilesW, (T - w + kFW - 1) / kFW) : 0;   // this wave's tiles
  const int C = (M + kFSlots - 1) / kFSlots;              // 16-slot chunks of the pair
  const int c0 = (int)(((long)C * blockIdx.x) / gridDim.x), c1 = (int)(((long)C * (blockIdx.x + 1)) / gridDim.x);
  if (c0 >= c1) return;
  const float t15 = 1.5f * thr[p];
  const float inv_thr2 = 1.0f / (t15 * t15);
  const float4 *mt = reinterpret_cast<const float4 *>(matches) + (size_t)p * N;
  const float *md = models + (size_t)p * M * 9;
  const uint8_t *vd = valid ? valid + (size_t)p * M : nullptr;
  unsigned char *mask0 = sh.raw;                               // [2][32768]
  uint32_t *queue = reinterpret_cast<uint32_t *>(sh.raw + 65536) + w * 1024;
  _Float16 *stage = reinterpret_cast<_Float16 *>(sh.raw);      // [512][kStagePitch]

  // ---- prologue 1: points -> LDS, coordinate scale 2^-s ---------------------------------------------------------
  float cm = 0.f;
  for (int n = tid; n < kFMaxN; n += kFT) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (n < N) v = mt[n];
    sh.pts[n] = v;
    const float a0 = fabsf(v.x), a1 = fabsf(v.y), a2 = fabsf(v.z), a3 = fabsf(v.w);
    if (is_finite(a0)) cm = fmaxf(cm, a0);
    if (is_finite(a1)) cm = fmaxf(cm, a1);
    if (is_finite(a2)) cm = fmaxf(cm, a2);
    if (is_finite(a3)) cm = fmaxf(cm, a3);
  }
  cm = wave_max_bcast(cm);
  if (lane == 0) sh.red[w] = cm;
  __syncthreads();
  float cmax = sh.red[0];
#pragma unroll
  for (int i = 1; i < kFW; ++i) cmax = fmaxf(cmax, sh.red[i]);
  int s = 0;
  if (cmax > 0.f) (void)frexpf(cmax, &s);                 // cmax * 2^-s in [0.5, 1)
  const float thp = ldexpf(t15, -s);                      // threshold in scaled coordinates
  const bool ok_mode = is_finite(thp) && thp > 6.103515625e-05f && thp < 16.0f && s <= 16 && s >= -16;
  int g = 0;
  if (ok_mode) {
    int eg;
    (void)frexpf(thp, &eg);
    g = min(15, max(-15, 1 - eg));                        // 2^g * thp in [1, 2) unless clamped
  }
  const float Theta = ok_mode ? ldexpf(thp, g) : 1.0f;   // (not ok_mode: every live model runs in all-candidates mode)
  const float A = (1.0f + kFEps + 1e-3f) * Theta * Theta;
  const float sc_g = ldexpf(1.0f, g);

  // ---- prologue 2: point-side operand rows (64 halves per point), staged 512 points at a time, then into registers --
  h8 Ar[kFTilesW], Aj[kFTilesW];
#pragma unroll
  for (int t = 0; t < kFTilesW; ++t) {
#pragma unroll
    for (int q = 0; q < 8; ++q) { Ar[t][q] = (_Float16)0.f; Aj[t][q] = (_Float16)0.f; }
  }
#pragma unroll
  for (int rd = 0; rd < kFMaxN / kFT; ++rd) {
    if (rd * kFT < N) {                                   // block-uniform
      __syncthreads();                                    // previous round's reads are done
      {
        const int n = rd * kFT + tid;
        const float4 v = sh.pts[n];
        const float x1 = ldexpf(v.x, -s), y1 = ldexpf(v.y, -s), x2 = ldexpf(v.z, -s), y2 = ldexpf(v.w, -s);
        const float F[8] = {x1 * x2, x1 * y2, x1, y1 * x2, y1 * y2, y1, x2, y2};
        const float G[10] = {x2 * x2, x2 * y2, y2 * y2, x2, y2, x1 * x1, x1 * y1, y1 * y1, x1, y1};
        float fs = 1.f, gs = 1.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) fs += fabsf(F[j]);
#pragma unroll
        for (int j = 0; j < 10; ++j) gs += fabsf(G[j]);
        const float Ep = kFKappaR * sc_g * fs + kFEabs;
        float H = (1.0f + 1.0f / kFEps) * Ep * Ep + kFKappaJ * 4.0f * A * gs + 1e-5f;
        H = H * (1.0f + 0.001953125f) + 1e-7f;             // the conversion below rounds to nearest: stay above
        _Float16 Hh = (_Float16)H;
        if (!(n < N) || !is_finite(H)) Hh = (_Float16)(-kFBig);   // padding point / non-finite point: never a candidate
        _Float16 *dst = stage + tid * kStagePitch;
        h8 o;
#pragma unroll
        for (int j = 0; j < 8; j += 2) {                  // r operand: (fh, fl, fh, fl) per feature
          _Float16 h0, l0, h1, l1;
          split2(F[j], h0, l0);
          split2(F[j + 1], h1, l1);
          o = (h8){h0, l0, h0, l0, h1, l1, h1, l1};
          *reinterpret_cast<h8 *>(dst + 4 * j) = o;
        }
        _Float16 jrow[32];
#pragma unroll
        for (int j = 0; j < 10; ++j) {                    // J operand: (gh, gl, gh) per monomial
          _Float16 gh, gl;
          split2(G[j], gh, gl);
          jrow[3 * j] = gh; jrow[3 * j + 1] = gl; jrow[3 * j + 2] = gh;
        }
        jrow[30] = Hh;
        jrow[31] = (_Float16)0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          o = (h8){jrow[8 * j], jrow[8 * j + 1], jrow[8 * j + 2], jrow[8 * j + 3], jrow[8 * j + 4], jrow[8 * j + 5],
                   jrow[8 * j + 6], jrow[8 * j + 7]};
          *reinterpret_cast<h8 *>(dst + 32 + 8 * j) = o;
        }
      }
      __syncthreads();
      // the round staged the tiles [32 rd, 32 rd + 32): every wave owns four of them (local tiles 4 rd + u)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const _Float16 *src = stage + (16 * (kFW * u + w) + row) * kStagePitch + 8 * kg;
        Ar[4 * rd + u] = *reinterpret_cast<const h8 *>(src);
        Aj[4 * rd + u] = *reinterpret_cast<const h8 *>(src + 32);
      }
    }
  }
  __syncthreads();
  // ---- prologue 3: clean mask buffers / accumulators ----------------------------------------------------------------
  for (int i = tid; i < 65536 / 16; i += kFT) reinterpret_cast<u4 *>(mask0)[i] = (u4){0u, 0u, 0u, 0u};
  if (tid < 2 * kFSlots) (&sh.acc[0][0])[tid] = 0ull;

  // ---- model-side operands of one chunk: lanes 0-15 of wave (chunk & 7), one model each (straight-line code: a
  // work split finer than a model needs dynamically indexed coefficients, which the compiler sends through scratch) ----
  float pm[9];
#pragma unroll
  for (int q = 0; q < 9; ++q) pm[q] = 0.f;
  bool pv = false;
  auto prep_load = [&](int cc) {
    if (w == (cc & (kFW - 1)) && lane < kFSlots) {
      const int slot = kFSlots * cc + lane;
      const bool inside = slot < M;
      pv = inside && (!vd || vd[slot] != 0);
#pragma unroll
      for (int q = 0; q < 9; ++q) pm[q] = inside ? md[(size_t)slot * 9 + q] : 0.f;
    }
  };
  auto prep_compute = [&](int cc, int nb) {
    if (w == (cc & (kFW - 1)) && lane < kFSlots) {
      const int i = lane;
      bool fin = true, nz = false;
#pragma unroll
      for (int q = 0; q < 9; ++q) { fin = fin && is_finite(pm[q]); nz = nz || (pm[q] != 0.f); }
      // coefficient q multiplies a feature of degree deg[q] in the coordinates
      constexpr int deg[9] = {2, 2, 1, 2, 2, 1, 1, 1, 0};
      float mp[9], mx = 0.f;
#pragma unroll
      for (int q = 0; q < 9; ++q) { mp[q] = ldexpf(pm[q], s * deg[q]); mx = fmaxf(mx, fabsf(mp[q])); }
      int ee = 0;
      if (mx > 0.f && is_finite(mx)) (void)frexpf(mx, &ee);
      const int e = -ee;                                   // mx * 2^e in [0.5, 1)
      const bool live = pv && fin && nz;
      const bool allc = live && (!ok_mode || !(mx > 0.f) || !is_finite(mx) || e > 30 || e < -30);
      const bool filt = live && !allc;
      float mpp[9];
#pragma unroll
      for (int q = 0; q < 9; ++q) mpp[q] = filt ? ldexpf(mp[q], e) : 0.f;
      // r operand: feature j of (x1x2, x1y2, x1, y1x2, y1y2, y1, x2, y2) has coefficient m[jr[j]]: (ch, ch, cl, cl)
      constexpr int jr[8] = {0, 3, 6, 1, 4, 7, 2, 5};
#pragma unroll
      for (int j = 0; j < 8; j += 2) {
        _Float16 h0, l0, h1, l1;
        split2(ldexpf(mpp[jr[j]], g), h0, l0);
        split2(ldexpf(mpp[jr[j + 1]], g), h1, l1);
        *reinterpret_cast<h8 *>(&sh.opR[nb][i][4 * j]) = (h8){h0, h0, l0, l0, h1, h1, l1, l1};
      }
      // J operand: monomial j of (x2^2 x2y2 y2^2 x2 y2 | x1^2 x1y1 y1^2 x1 y1): A q_j as (qh, qh, ql);
      // q_j = k (m_a m_b + m_c m_d)
      constexpr int ia[10] = {0, 0, 3, 0, 3, 0, 0, 1, 0, 1};
      constexpr int ib[10] = {0, 3, 3, 6, 6, 0, 1, 1, 2, 2};
      constexpr int ic[10] = {1, 1, 4, 1, 4, 3, 3, 4, 3, 4};
      constexpr int id[10] = {1, 4, 4, 7, 7, 3, 4, 4, 5, 5};
      constexpr float kk[10] = {1.f, 2.f, 1.f, 2.f, 2.f, 1.f, 2.f, 1.f, 2.f, 2.f};
      _Float16 jrow[32];
#pragma unroll
      for (int j = 0; j < 10; ++j) {
        const float qv = kk[j] * fmaf(mpp[ia[j]], mpp[ib[j]], mpp[ic[j]] * mpp[id[j]]);
        _Float16 qh, ql;
        split2(filt ? A * qv : 0.f, qh, ql);
        jrow[3 * j] = qh; jrow[3 * j + 1] = qh; jrow[3 * j + 2] = ql;
      }
      jrow[30] = (_Float16)1.0f;                           // pairs with the point's slack H_p
      jrow[31] = (_Float16)0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        *reinterpret_cast<h8 *>(&sh.opJ[nb][i][8 * j]) = (h8){jrow[8 * j], jrow[8 * j + 1], jrow[8 * j + 2], jrow[8 * j + 3],
                                                              jrow[8 * j + 4], jrow[8 * j + 5], jrow[8 * j + 6], jrow[8 * j + 7]};
      const float qc = fmaf(mpp[6], mpp[6], fmaf(mpp[7], mpp[7], fmaf(mpp[2], mpp[2], mpp[5] * mpp[5])));
      sh.cin[nb][i][0] = filt ? ldexpf(mpp[8], g) : 0.f;
      sh.cin[nb][i][1] = filt ? A * qc : (allc ? kFBig : -kFBig);
#pragma unroll
      for (int q = 0; q < 9; ++q) sh.mraw[nb][i][q] = pm[q];
      sh.flags[nb][i] = (live ? 1 : 0) | ((pv && (!fin || !nz)) ? 2 : 0);   // bit 1: score NaN (non-finite or all-zero model)
    }
  };
  prep_load(c0);
  prep_compute(c0, 0);
  __syncthreads();

  const uint32_t pay = (uint32_t)row | ((uint32_t)kg << 4);

  KF_STAGE_BEGIN();
  for (int c = c0; c < c1; ++c) {
    const int b = (c - c0) & 1;
    if (c + 1 < c1) prep_load(c + 1);
    const int myflag = (tid < kFSlots) ? sh.flags[b][tid] : 0;
    const h8 Br = *reinterpret_cast<const h8 *>(&sh.opR[b][row][8 * kg]);
    const h8 Bj = *reinterpret_cast<const h8 *>(&sh.opJ[b][row][8 * kg]);
    const float cr = sh.cin[b][row][0], cj = sh.cin[b][row][1];
    const f4 Cr = {cr, cr, cr, cr}, Cj = {cj, cj, cj, cj};

    KF_STAGE(0);
    // ---- filter: per lane one candidate bit per tile (any of the lane's four (point, model) evaluations), no branches;
    // then the set bits become queue entries ----
    uint32_t cbits = 0u;                                   // bit t = tile t
#pragma unroll
    for (int t0 = 0; t0 < kFTilesW; t0 += DR_KF_GROUP) {
      if (!DR_KF_STOREONLY && !(DR_KF_SKIP & 2) && t0 < ntw) {   // wave-uniform
        f4 Dr[DR_KF_GROUP], Dj[DR_KF_GROUP];
#pragma unroll
        for (int u = 0; u < DR_KF_GROUP; ++u) {
          Dr[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ar[t0 + u], Br, Cr, 0, 0, 0);
          Dj[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Aj[t0 + u], Bj, Cj, 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < DR_KF_GROUP; ++u) {
          // d = Jt - rt^2 >= 0  <=>  candidate; NaN (non-finite point) never is: fmaxf drops NaN operands
          const float d0 = fmaf(-Dr[u][0], Dr[u][0], Dj[u][0]), d1 = fmaf(-Dr[u][1], Dr[u][1], Dj[u][1]);
          const float d2 = fmaf(-Dr[u][2], Dr[u][2], Dj[u][2]), d3 = fmaf(-Dr[u][3], Dr[u][3], Dj[u][3]);
          const float mx = fmaxf(fmaxf(fmaxf(d0, d1), d2), d3);
          cbits |= (mx >= 0.f && (t0 + u < ntw)) ? (1u << (t0 + u)) : 0u;
        }
      }
    }
    int qn = 0;
    {
      // exclusive prefix of the per-lane counts (DPP scan), then every lane appends its own entries
      const int cnt = __popc(cbits);
      int incl = cnt;
      incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xF, 0xF, false);   // row_shr:1
      incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xF, 0xF, false);   // row_shr:2
      incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xF, 0xF, false);   // row_shr:4
      incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xF, 0xF, false);   // row_shr:8  -> inclusive scan inside each row of 16
      incl += __builtin_amdgcn_update_dpp(0, incl, 0x142, 0xA, 0xF, false);   // row_bcast:15 -> rows 1 and 3
      incl += __builtin_amdgcn_update_dpp(0, incl, 0x143, 0xC, 0xF, false);   // row_bcast:31 -> rows 2 and 3
      qn = __builtin_amdgcn_readlane(incl, 63);
      int off = incl - cnt;
      uint32_t rem = cbits;
      while (rem) {                                         // lanes drop out as their bits run out
        const int t = __builtin_ctz(rem);
        rem &= rem - 1u;
        queue[off++] = pay | ((uint32_t)t << 6);
      }
    }

    KF_STAGE(1);
#ifdef DR_PROFILE_STAGES
    kf_acc[6] += (unsigned long long)qn; kf_acc[7] += (unsigned long long)((qn + 63) / 64);
#endif
    // ---- exact evaluation of the queued (model, four points) entries ----
    unsigned char *mbuf = mask0 + b * 32768;
#pragma unroll 1
    for (int base = 0; base < ((DR_KF_SKIP & 1) ? 0 : qn); base += 64) {
      const int i = base + lane;
      if (i < qn) {
        const uint32_t e = queue[i];
        const int col = e & 15, qq = (e >> 4) & 3, t = e >> 6;
        const int n0 = 16 * (kFW * t + w) + 4 * qq;
        const float *mm = sh.mraw[b][col];
        const float4 ma = *reinterpret_cast<const float4 *>(mm), mb = *reinterpret_cast<const float4 *>(mm + 4);
        const float m[9] = {ma.x, ma.y, ma.z, ma.w, mb.x, mb.y, mb.z, mb.w, mm[8]};
        uint32_t word = 0u, qs = 0u;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float4 pt = sh.pts[n0 + j];
          const uint32_t bits = __float_as_uint(sampson_exact(m, pt.x, pt.y, pt.z, pt.w, inv_thr2));
          const float v = -__int_as_float(min((int)bits, 0));     // max(-sv, 0); +NaN (0/0 point) -> 0
          qs += (uint32_t)(v * 536870912.0f);                    // 2^29: four of them fit 32 bits
          word |= (bits >> 31) << (8 * j);
        }
        if (word) *reinterpret_cast<uint32_t *>(mbuf + col * N + n0) = word;
        if (qs) atomicAdd(&sh.acc[b][col], (unsigned long long)qs);
      }
    }
    KF_STAGE(2);
    if (c + 1 < c1 && !(DR_KF_SKIP & 4)) prep_compute(c + 1, b ^ 1);
    KF_STAGE(3);
    block_sync_lds();
    KF_STAGE(4);

    // ---- stream the chunk's mask image out, leave the buffer clean; scores ----
    const int rows = min(kFSlots, M - kFSlots * c);
    if (masks) {
      const int nvec = (rows * N) >> 4;                      // <= 2048 vectors: at most four per thread
      u4 *src = reinterpret_cast<u4 *>(mbuf);
      u4 *dst = reinterpret_cast<u4 *>(masks + ((size_t)p * M + (size_t)kFSlots * c) * N);
      u4 v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = tid + kFT * r;
        if (i < nvec) v[r] = src[i];
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = tid + kFT * r;
        if (i < nvec) {
          mask_store(dst + i, v[r]);
          src[i] = (u4){0u, 0u, 0u, 0u};
        }
      }
    }
    if (tid < rows) {
      const unsigned long long a = sh.acc[b][tid];
      sh.acc[b][tid] = 0ull;
      float sc = (float)a * 1.862645149230957e-09f;       // 2^-29
      if (myflag & 2) sc = NAN;                             // valid slot holding a non-finite or all-zero model (as the general kernel)
      scores[(size_t)p * M + kFSlots * c + tid] = sc;
    }
    KF_STAGE(5);
  }
  KF_STAGE_END();
}

bool msac_filter_supported(int N) { return N % 16 == 0 && N >= 16 && N <= kFMaxN; }

bool msac_filter_profitable(int P, int M, int N) {
  return false;   // until the kernel beats the general one on the device (path 2 selects it explicitly)
  if (N < 256) return false;
  const long chunks = (long)P * ((M + kFSlots - 1) / kFSlots);
  return chunks >= 1024;    // at least ~4 chunks per block of a chip-filling grid: the prologue has to amortise
}

int msac_filter_launch(const float *matches, const float *models, const uint8_t *valid, const float *thr, int P, int M,
                       int N, float *scores, uint8_t *masks, hipStream_t st) {
  const int C = (M + kFSlots - 1) / kFSlots;
  int bpp = (256 + P - 1) / P;          // one block per CU when the pairs allow it
  bpp = max(1, min(bpp, C));
  hipLaunchKernelGGL(msac_filter_kernel, dim3(bpp, P), dim3(kFT), 0, st, matches, models, valid, thr, M, N, scores, masks);
  return check_launch("msac_filter_kernel");
}

}  // namespace dr

DR_DEFINE_STAGE_READER(dr_kf_stage_cycles)

#ifdef DR_KF_STANDALONE   // scratch/k4f_check.py builds variants of this file alone (A/B of the knobs above)
extern "C" int dr_kf_run(const float *matches, const float *models, const uint8_t *valid, const float *thr, int P, int M,
                         int N, float *scores, uint8_t *masks, void *stream) {
  return dr::msac_filter_launch(matches, models, valid, thr, P, M, N, scores, masks, (hipStream_t)stream);
}
#endif

