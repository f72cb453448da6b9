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
// timeline of ONE interval of block (0, 0): time stamps of every wave at fixed points (scratch/k4f_check.py prints them)
static __device__ unsigned long long g_kf_trace[8][16];
#define KF_TR(k) do { if (blockIdx.x == 0 && blockIdx.y == 0 && kf_it == 40 && (threadIdx.x & 63) == 0) \
                        ::g_kf_trace[threadIdx.x >> 6][k] = __builtin_readcyclecounter(); } while (0)
extern "C" int dr_kf_trace(unsigned long long *out128) {
  return hipMemcpyFromSymbol(out128, HIP_SYMBOL(::g_kf_trace), sizeof(unsigned long long) * 128) == hipSuccess ? 0 : -2;
}
#else
#define KF_TR(k) do {} while (0)
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
#endif                      // consumer), 4 no per-chunk operand preparation, 8 filter without its matrix instructions, 16 filter
                            // without its vector work; bits may be combined
#ifndef DR_KF_GSHIFT
#define DR_KF_GSHIFT 2      // group A = waves with bit GSHIFT clear (2: waves 0-3; 0: even waves; 1: waves 0,1,4,5): the two groups
#endif                      // must be the two waves of each SIMD for the matrix / vector overlap to exist
#ifndef DR_KF_GROUP
#define DR_KF_GROUP 2       // tiles per software-pipeline stage of the filter (4: ten registers spill; 8: 113)
#endif
#ifndef DR_KF_PIPE
#define DR_KF_PIPE 1        // 1: next group's matrix instructions issued before this group's compares
#endif

constexpr int kFT = 512, kFW = kFT / 64, kFSlots = 16, kFMaxN = 2048, kFTilesW = 16;
constexpr int kStagePitch = 72;   // halves per staged point row (64 used; 144 B pitch: conflict-free 16-byte reads)
constexpr float kFEps = 1.0f / 16.0f;
constexpr float kFKappaR = 5e-6f, kFEabs = 4e-3f, kFKappaJ = 4e-6f;
constexpr float kFBig = 60000.0f;
constexpr int kMaskBuf = kFSlots * kFMaxN;   // bytes of one 16-slot mask image

struct FilterShared {
  alignas(16) unsigned char mask[3][kMaskBuf];   // chunk i lives in buffer i % 3 (filled over two intervals, streamed out in
                                                 // the third); the prologue stages the point-side operand rows here
  alignas(16) uint16_t queue[kFW][1024];         // wave-private candidate entries: model column | quarter << 4 | tile << 6
  alignas(16) float4 pts[kFMaxN + kFMaxN / 16];  // point n at index n + n / 16: the consumer's gathers (four consecutive
                                                 // points of arbitrary tiles per lane) would otherwise all start on four banks
  alignas(16) _Float16 opR[2][kFSlots][32];      // model-side operands of chunk i: buffer i & 1
  alignas(16) _Float16 opJ[2][kFSlots][32];
  alignas(16) float mraw[4][kFSlots][12];        // raw coefficients of chunk i (for the exact chain): buffer i & 3
  alignas(16) float norm[2][kFSlots][12];        // normalised coefficients [0..8], 1 [9], 0 [10] of chunk i: buffer i & 1
  float cin[4][kFSlots][2];                      // accumulator inputs (constant terms) of chunk i: buffer i & 3
  int flags[8][kFSlots];                         // chunk i: buffer i & 7 (bit 1: the score is NaN)
  unsigned long long acc[3][kFSlots];            // fixed-point score sums of chunk i: buffer i % 3
  uint32_t tab_idx[18];                          // operand item u: value = tab_k[u] * (n[a] n[b] + n[c] n[d]), a | b<<4 | c<<8 | d<<12
  float tab_k[18];
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

#ifndef DR_KF_NOLO
#define DR_KF_NOLO 0        // timing experiment only (wrong results): 1 = low halves forced to zero (are f16 denormal operands slow?)
#endif
__device__ __forceinline__ void split2(float v, _Float16 &h, _Float16 &l) {
  h = (_Float16)v;
  l = DR_KF_NOLO ? (_Float16)0.f : (_Float16)(v - (float)h);
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
  const bool grp_a = ((w >> DR_KF_GSHIFT) & 1) == 0;       // see the interval loop
  const int gidx = DR_KF_GSHIFT == 2 ? (w & 3) : (DR_KF_GSHIFT == 0 ? (w >> 1) : ((w & 1) | ((w >> 2) << 1)));   // index in the group
  const int gtid = 64 * gidx + lane;                       // thread index inside the group (0..255)
  const int T = N >> 4;                                   // 16-point tiles (N % 16 == 0)
  // wave w owns the tiles 8 t + w (t = 0..15): interleaved, so that spatially sorted inliers (the synthetic pairs keep
  // them in the second half of the point list) load the eight waves evenly
  const int ntw = (T > w) ? min(kFTilesW, (T - w + kFW - 1) / kFW) : 0;   // this wave's tiles
  const int C = (M + kFSlots - 1) / kFSlots;              // 16-slot chunks of the pair
  const int c0 = (int)(((long)C * blockIdx.x) / gridDim.x), c1 = (int)(((long)C * (blockIdx.x + 1)) / gridDim.x);
  const int n = c1 - c0;                                  // this block's chunks c0 .. c1 - 1
  if (n <= 0) return;
  const float t15 = 1.5f * thr[p];
  const float inv_thr2 = 1.0f / (t15 * t15);
  const float4 *mt = reinterpret_cast<const float4 *>(matches) + (size_t)p * N;
  const float *md = models + (size_t)p * M * 9;
  const uint8_t *vd = valid ? valid + (size_t)p * M : nullptr;
  uint16_t *queue = sh.queue[w];
  _Float16 *stage = reinterpret_cast<_Float16 *>(&sh.mask[0][0]);      // [512][kStagePitch]

  // ---- prologue 1: points -> LDS, coordinate scale 2^-s ---------------------------------------------------------
  float cm = 0.f;
  for (int i = tid; i < kFMaxN; i += kFT) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < N) v = mt[i];
    sh.pts[i + (i >> 4)] = v;
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
        const int i = rd * kFT + tid;
        const float4 v = sh.pts[i + (i >> 4)];
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
        if (!(i < N) || !is_finite(H)) Hh = (_Float16)(-kFBig);   // padding point / non-finite point: never a candidate
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
  // ---- prologue 3: clean mask buffers / accumulators, operand item table, constant operand slots ------------------
  for (int i = tid; i < 3 * kMaskBuf / 16; i += kFT) reinterpret_cast<u4 *>(&sh.mask[0][0])[i] = (u4){0u, 0u, 0u, 0u};
  if (tid < 3 * kFSlots) (&sh.acc[0][0])[tid] = 0ull;
  if (tid < 18) {
    // r operand, feature u of (x1x2, x1y2, x1, y1x2, y1y2, y1, x2, y2): 2^g m[jr[u]]  (n[9] = 1, n[10] = 0)
    // J operand, monomial j of (x2^2 x2y2 y2^2 x2 y2 | x1^2 x1y1 y1^2 x1 y1): A k_j (m_a m_b + m_c m_d)
    static constexpr unsigned char jr[8] = {0, 3, 6, 1, 4, 7, 2, 5};
    static constexpr unsigned char ia[10] = {0, 0, 3, 0, 3, 0, 0, 1, 0, 1};
    static constexpr unsigned char ib[10] = {0, 3, 3, 6, 6, 0, 1, 1, 2, 2};
    static constexpr unsigned char ic[10] = {1, 1, 4, 1, 4, 3, 3, 4, 3, 4};
    static constexpr unsigned char id[10] = {1, 4, 4, 7, 7, 3, 4, 4, 5, 5};
    static constexpr float kk[10] = {1.f, 2.f, 1.f, 2.f, 2.f, 1.f, 2.f, 1.f, 2.f, 2.f};
    if (tid < 8) {
      sh.tab_idx[tid] = (uint32_t)jr[tid] | (9u << 4) | (10u << 8) | (10u << 12);
      sh.tab_k[tid] = sc_g;
    } else {
      const int j = tid - 8;
      sh.tab_idx[tid] = (uint32_t)ia[j] | ((uint32_t)ib[j] << 4) | ((uint32_t)ic[j] << 8) | ((uint32_t)id[j] << 12);
      sh.tab_k[tid] = A * kk[j];
    }
  }
  if (tid < 2 * kFSlots) {                                 // J operand slots 30, 31: (1, 0) -- pairs with the point's slack H_p
    _Float16 *d = &sh.opJ[tid >> 4][tid & 15][30];
    d[0] = (_Float16)1.0f;
    d[1] = (_Float16)0.f;
  }

  // ---- model-side metadata of a chunk, stage A: lanes 0-15 of wave (chunk & 7), one model each: validity, scale,
  // normalised coefficients, constant terms.  Stage B (next interval): 16 x 18 operand items spread over five waves. ----
  float pm[9];
#pragma unroll
  for (int q = 0; q < 9; ++q) pm[q] = 0.f;
  uint32_t pvb = 0u;
  // The loads belong to group B (waves 4-7), which never stores to global memory: loads and stores share one in-order
  // counter (vmcnt), so a wave that also streams masks would have to see its stores acknowledged before it may use the
  // loaded coefficients (measured: +0.6 us per interval).  Nothing is tested here (a test would wait for the load).
  auto prep_wave = [&](int cc) { return !grp_a && gidx == (cc & 3); };
  auto prep_load = [&](int cc) {
    if (prep_wave(cc) && lane < kFSlots) {
      const int slot = min(kFSlots * cc + lane, M - 1);     // clamped: stage A discards slots >= M
      pvb = vd ? (uint32_t)vd[slot] : 1u;
#pragma unroll
      for (int q = 0; q < 9; ++q) pm[q] = md[(size_t)slot * 9 + q];
    }
  };
  auto stage_a = [&](int cc) {
    if (prep_wave(cc) && lane < kFSlots) {
      const int i = lane;
      const bool pv = (kFSlots * cc + lane < M) && pvb != 0u;
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
      float4 *nr = reinterpret_cast<float4 *>(sh.norm[cc & 1][i]);
      nr[0] = make_float4(mpp[0], mpp[1], mpp[2], mpp[3]);
      nr[1] = make_float4(mpp[4], mpp[5], mpp[6], mpp[7]);
      nr[2] = make_float4(mpp[8], 1.0f, 0.f, 0.f);
      float4 *mr = reinterpret_cast<float4 *>(sh.mraw[cc & 3][i]);
      mr[0] = make_float4(pm[0], pm[1], pm[2], pm[3]);
      mr[1] = make_float4(pm[4], pm[5], pm[6], pm[7]);
      mr[2] = make_float4(pm[8], 0.f, 0.f, 0.f);
      const float qc = fmaf(mpp[6], mpp[6], fmaf(mpp[7], mpp[7], fmaf(mpp[2], mpp[2], mpp[5] * mpp[5])));
      sh.cin[cc & 3][i][0] = filt ? ldexpf(mpp[8], g) : 0.f;
      sh.cin[cc & 3][i][1] = filt ? A * qc : (allc ? kFBig : -kFBig);
      sh.flags[cc & 7][i] = (live ? 1 : 0) | ((pv && (!fin || !nz)) ? 2 : 0);   // bit 1: score NaN (non-finite or all-zero model)
    }
  };
  auto stage_b = [&](int cc) {
    const int it = 64 * w + lane;
    if (w < 5 && it < 18 * kFSlots) {
      const int i = it / 18, u = it - 18 * i;
      const uint32_t ti = sh.tab_idx[u];
      const float K = sh.tab_k[u];
      const float *nr = sh.norm[cc & 1][i];
      const float val = K * fmaf(nr[ti & 15u], nr[(ti >> 4) & 15u], nr[(ti >> 8) & 15u] * nr[(ti >> 12) & 15u]);
      _Float16 h, l;
      split2(val, h, l);
      if (u < 8) {
        *reinterpret_cast<h4 *>(&sh.opR[cc & 1][i][4 * u]) = (h4){h, h, l, l};        // pairs with (fh, fl, fh, fl)
      } else {
        _Float16 *d = &sh.opJ[cc & 1][i][3 * (u - 8)];                               // pairs with (gh, gl, gh)
        d[0] = h; d[1] = h; d[2] = l;
      }
    }
  };
  prep_load(c0);
  stage_a(c0);
  if (n > 1) { prep_load(c0 + 1); stage_a(c0 + 1); }
  __syncthreads();
  stage_b(c0);
  __syncthreads();

  const uint32_t pay = (uint32_t)row | ((uint32_t)kg << 4);
  int kf_it = -1;   // interval counter (profiling builds trace one interval)
  (void)kf_it;

  // ---- filter of chunk cc: per lane one candidate bit per tile (any of the lane's four (point, model) evaluations), no
  // branches; then the set bits become queue entries.  Returns the number of entries. ----
  auto filter = [&](int cc) -> int {
    const h8 Br = *reinterpret_cast<const h8 *>(&sh.opR[cc & 1][row][8 * kg]);
    const h8 Bj = *reinterpret_cast<const h8 *>(&sh.opJ[cc & 1][row][8 * kg]);
    const float cr = sh.cin[cc & 3][row][0], cj = sh.cin[cc & 3][row][1];
    const f4 Cr = {cr, cr, cr, cr}, Cj = {cj, cj, cj, cj};
    uint32_t cbits = 0u;                                   // bit t = tile t
    // Software pipeline over groups of DR_KF_GROUP tiles: the matrix instructions of group k + 1 are issued BEFORE the
    // compares of group k (two accumulator sets), so the matrix pipe works while the vector pipe digests the previous group.
    constexpr int G = DR_KF_GROUP, NG = kFTilesW / G;
    f4 Dr[2][G], Dj[2][G];
    auto issue = [&](int k, int set) {
#pragma unroll
      for (int u = 0; u < G; ++u) {
        if (DR_KF_SKIP & 8) {      // timing experiment: no matrix instructions
          Dr[set][u] = Cr + __builtin_bit_cast(f4, __builtin_shufflevector(Ar[k * G + u], Ar[k * G + u], 0, 1, 2, 3, 4, 5, 6, 7));
          Dj[set][u] = Cj + __builtin_bit_cast(f4, __builtin_shufflevector(Aj[k * G + u], Aj[k * G + u], 0, 1, 2, 3, 4, 5, 6, 7));
        } else {
          Dr[set][u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ar[k * G + u], Br, Cr, 0, 0, 0);
          Dj[set][u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Aj[k * G + u], Bj, Cj, 0, 0, 0);
        }
      }
    };
    const bool run = !DR_KF_STOREONLY && !(DR_KF_SKIP & 2);
    if (run && 0 < ntw) issue(0, 0);
#pragma unroll
    for (int k = 0; k < NG; ++k) {
      if (run && k * G < ntw) {                           // wave-uniform
        if (DR_KF_PIPE && k + 1 < NG && (k + 1) * G < ntw) issue(k + 1, (k + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < G; ++u) {
          // d = Jt - rt^2 >= 0  <=>  candidate; NaN (non-finite point) never is: fmaxf drops NaN operands
          const f4 r_ = Dr[k & 1][u], j_ = Dj[k & 1][u];
          if (DR_KF_SKIP & 16) {   // timing experiment: (almost) no vector work behind the matrix instructions
            cbits |= (r_[0] + j_[0] > 1e30f) ? (1u << (k * G + u)) : 0u;
            continue;
          }
          const float d0 = fmaf(-r_[0], r_[0], j_[0]), d1 = fmaf(-r_[1], r_[1], j_[1]);
          const float d2 = fmaf(-r_[2], r_[2], j_[2]), d3 = fmaf(-r_[3], r_[3], j_[3]);
          const float mx = fmaxf(fmaxf(fmaxf(d0, d1), d2), d3);
          cbits |= (mx >= 0.f && (k * G + u < ntw)) ? (1u << (k * G + u)) : 0u;
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!DR_KF_PIPE && k + 1 < NG && (k + 1) * G < ntw) issue(k + 1, (k + 1) & 1);
      }
    }
    KF_TR(9);
    // exclusive prefix of the per-lane counts (DPP scan), then every lane appends its own entries
    const int cnt = __popc(cbits);
    int incl = cnt;
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xF, 0xF, false);   // row_shr:1
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xF, 0xF, false);   // row_shr:2
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xF, 0xF, false);   // row_shr:4
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xF, 0xF, false);   // row_shr:8  -> inclusive scan inside each row of 16
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x142, 0xA, 0xF, false);   // row_bcast:15 -> rows 1 and 3
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x143, 0xC, 0xF, false);   // row_bcast:31 -> rows 2 and 3
    const int total = __builtin_amdgcn_readlane(incl, 63);
    int off = incl - cnt;
    uint32_t rem = cbits;
    while (rem) {                                         // lanes drop out as their bits run out
      const int t = __builtin_ctz(rem);
      rem &= rem - 1u;
      queue[off++] = (uint16_t)(pay | ((uint32_t)t << 6));
    }
    return total;
  };

  // ---- exact evaluation of the queued (model, four points) entries of chunk cc (mask / score buffer bm = cc % 3) ----
  auto consume = [&](int cc, int bm, int qn) {
    unsigned char *mbuf = sh.mask[bm];
    // batch b gives lane l the entry l * nb + b: neighbouring lanes take entries nb apart, i.e. of different producer lanes
    // and so (mostly) of different models -- consecutive entries share their model, and 64 LDS atomics on one address serialise
    const int nb = (qn + 63) >> 6;
#pragma unroll 1
    for (int bt = 0; bt < ((DR_KF_SKIP & 1) ? 0 : nb); ++bt) {
      const int i = lane * nb + bt;
      if (i < qn) {
        const uint32_t e = queue[i];
        const int col = e & 15, qq = (e >> 4) & 3, t = e >> 6;
        const int n0 = 16 * (kFW * t + w) + 4 * qq;
        const float *mm = sh.mraw[cc & 3][col];
        const float4 ma = *reinterpret_cast<const float4 *>(mm), mb = *reinterpret_cast<const float4 *>(mm + 4);
        const float m[9] = {ma.x, ma.y, ma.z, ma.w, mb.x, mb.y, mb.z, mb.w, mm[8]};
        uint32_t word = 0u, qs = 0u;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float4 pt = (DR_KF_SKIP & 128) ? make_float4(0.1f * j, 0.2f, 0.3f, 0.1f * qq) : sh.pts[n0 + (n0 >> 4) + j];   // 128: no point gather
          const uint32_t bits = __float_as_uint(sampson_exact(m, pt.x, pt.y, pt.z, pt.w, inv_thr2));
          const float v = -__int_as_float(min((int)bits, 0));     // max(-sv, 0); +NaN (0/0 point) -> 0
          qs += (uint32_t)(v * 536870912.0f);                    // 2^29: four of them fit 32 bits
          word |= (bits >> 31) << (8 * j);
        }
        if (word) *reinterpret_cast<uint32_t *>(mbuf + col * N + n0) = word;
        if (qs) atomicAdd(&sh.acc[bm][col], (unsigned long long)qs);
      }
    }
  };

  // ---- group A (256 threads) streams chunk cc's mask image out, leaves the buffer clean and writes the scores ----
  auto flush = [&](int cc, int bm) {
    const int rows = min(kFSlots, M - kFSlots * cc);
    if (masks) {
      const int nvec = (rows * N) >> 4;                      // <= 2048 vectors: at most eight per thread
      u4 *src = reinterpret_cast<u4 *>(sh.mask[bm]);
      u4 *dst = reinterpret_cast<u4 *>(masks + ((size_t)p * M + (size_t)kFSlots * cc) * N);
#pragma unroll
      for (int r0 = 0; r0 < 8; r0 += 4) {
        u4 v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = gtid + 256 * (r0 + r);
          if (i < nvec) v[r] = (DR_KF_SKIP & 64) ? (u4){0u, 0u, 0u, 0u} : src[i];   // 64: timing experiment, no LDS read
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = gtid + 256 * (r0 + r);
          if (i < nvec) {
            mask_store(dst + i, v[r]);
            if (!(DR_KF_SKIP & 32)) src[i] = (u4){0u, 0u, 0u, 0u};                       // 32: timing experiment, no re-zeroing
          }
        }
      }
    }
    if (gtid < rows) {
      const unsigned long long a = sh.acc[bm][gtid];
      sh.acc[bm][gtid] = 0ull;
      float sc = (float)a * 1.862645149230957e-09f;       // 2^-29
      if (sh.flags[cc & 7][gtid] & 2) sc = NAN;             // valid slot holding a non-finite or all-zero model (as the general kernel)
      scores[(size_t)p * M + kFSlots * cc + gtid] = sc;
    }
  };

  // ---- intervals.  Waves 0-3 (group A) filter and evaluate chunk i in interval i; their SIMD partners, waves 4-7 (group
  // B), evaluate chunk i - 1 first and filter chunk i afterwards: on every SIMD one wave is on the matrix pipe while the
  // other is on the vector pipe.  Chunk i's image is complete at the end of interval i + 1 and streamed out in interval
  // i + 2 by group A (group B never stores: it owns the model loads, see prep_load).  One LDS-only block barrier per
  // interval. ----
  int qn_b = 0;                    // group B: entries of the chunk it filtered in the previous interval
  int i3 = 0;                      // it % 3
  KF_STAGE_BEGIN();
  for (int it = 0; it < n + 2; ++it) {
    const int ci = c0 + it;
    kf_it = it;
    const int b0 = i3, b1 = (i3 + 2) % 3, b2 = (i3 + 1) % 3;   // buffers of chunks ci, ci - 1, ci - 2
    KF_TR(0);
    if (it + 2 < n) prep_load(ci + 2);
    KF_STAGE(0);
    KF_TR(1);
    if (grp_a) {
      if (it >= 2) flush(ci - 2, b2);
      KF_STAGE(5);
      KF_TR(2);
      if (it < n) {
        const int q = filter(ci);
        KF_STAGE(1);
        KF_TR(3);
#ifdef DR_PROFILE_STAGES
        kf_acc[6] += (unsigned long long)q; kf_acc[7] += (unsigned long long)((q + 63) / 64);
#endif
        consume(ci, b0, q);
        KF_STAGE(2);
        KF_TR(4);
      }
    } else {
      if (it >= 1 && it - 1 < n) consume(ci - 1, b1, qn_b);
      KF_STAGE(2);
      KF_TR(4);
      if (it < n) {
        qn_b = filter(ci);
        KF_TR(3);
#ifdef DR_PROFILE_STAGES
        kf_acc[6] += (unsigned long long)qn_b; kf_acc[7] += (unsigned long long)((qn_b + 63) / 64);
#endif
      }
      KF_STAGE(1);
    }
    KF_TR(5);
    if (!(DR_KF_SKIP & 4)) {
      if (it + 1 < n) stage_b(ci + 1);
      KF_TR(6);
      if (it + 2 < n) stage_a(ci + 2);
    }
    KF_STAGE(3);
    KF_TR(7);
    block_sync_lds();
    KF_STAGE(4);
    KF_TR(8);
    i3 = (i3 == 2) ? 0 : i3 + 1;
  }
  KF_STAGE_END();
}

bool msac_filter_supported(int N) { return N % 16 == 0 && N >= 16 && N <= kFMaxN; }

// Never chosen automatically: measured 262-269 us against 200-216 us for the general kernel at the benchmark shape
// (profiles/r2_k4_filter_experiments.md).  The filter kernel is opt-in through dr_msac_score_path_f32(path = 2) only.
bool msac_filter_profitable(int, int, int) { return false; }

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

