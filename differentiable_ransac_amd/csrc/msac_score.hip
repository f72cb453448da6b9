// K4 -- MSAC soft-inlier scoring on the Sampson distance (reference: MSACScore.score,
// scorings/msac_score.py:12-55) and K6 -- per-pair arg-max / best mask (ransac.py:111-120).
//
// Roofline: per pair bytes = 16N + 36M + 4M + M*N (bool masks are part of the
// score() contract), flops = 39*M*N  =>  38 flop/B at C2: the kernel is f32-VALU bound, the
// masks are the only HBM stream that matters.
//
// Mapping.  One block = (pair p, tile of model slots, chunk range of points).  A lane owns 16 (f32 fast
// path, N % 16 == 0) or 8 CONSECUTIVE points held in VGPRs for the whole block lifetime, so the inner
// loop touches no LDS and no vector memory except one 16-byte (8-byte) mask store per model: the model's
// nine coefficients are wave-uniform and arrive through the scalar cache (s_load, prefetched one model
// ahead) into SGPRs, which VALU instructions read for free.  Slots flagged invalid by the solver are
// never evaluated (the loop walks the set bits of a per-tile validity mask).  Score partials are reduced
// inside the wave and summed across the block's waves through LDS; with one chunk per pair (N <= 2048)
// the result is stored directly, i.e. deterministically; larger N split over blocks use one atomicAdd
// per (block, model).
// Measured A/B (scratch/ab_k4.py, P=32, N=2000, M=10240, 44.5 % valid slots, masks on): 32-slot tiles +
// 8-byte stores 279 us; 64-slot tiles 264 us; zero rows after the evaluations 259 us; 16 points per lane +
// 16-byte stores 229 us (the mask stream is store-issue bound: halving the store count is worth 30 us).
// Tried and dropped: dividing only where the wave holds an inlier (ballot-guarded rcp): 252 us -- the branch costs more
// than the quarter-rate reciprocals it saves, because the VALID models nearly always have inliers somewhere in a wave;
// model coefficients staged in LDS + in-order VGPR prefetch instead of scalar-cache loads: 231 vs 235 us (noise: the
// scalar-load latency is already hidden by occupancy); unpacked v_fma_f32 instead of v_pk_fma_f32: 224 vs 177 us with the
// masks off -- packing is worth 1.27x here; zero rows drip-fed between the evaluations instead of after them: 242 vs
// 241 us (the cost is the store stream itself, not its burstiness).  Store-pattern micro-benchmark
// (scratch/store_patterns.py): this row-after-row 1 KiB pattern streams zeros at 3.96 TB/s, a block flushing 16 rows as
// one contiguous range at 5.5 TB/s; but staging 16-slot windows in LDS to flush them that way costs two block barriers
// per window and un-overlaps the flush: 306 us.  The mask stream therefore keeps costing ~50 us on top of the VALU time.
// Second pass (same harness): soft score accumulated as min_i32(bits(sv), 0) instead of fmaxf(-sv, 0) (the compiler
// canonicalises fmaxf's operand: two instructions per point): 237 -> 223 us with masks, 180 -> 163 us without.  One
// reciprocal per two points (rcp(jj0*jj1) * jj1, * jj0): 221 us -- v_rcp_f32 measures 7 clk per wave and overlaps the FMA
// pipe (scratch/valu_rate.py), not the 16 clk of a quarter-rate op, so there is little to save; dropped (a degenerate
// point would poison its neighbour).  A wave-per-row kernel (32 points per lane, 8-row windows assembled in wave-private
// LDS and flushed as one line-aligned contiguous range: 5.4 TB/s as a pure store pattern) runs at 2 waves/SIMD and loses:
// 278 us.  Mask rows padded to a 2048-byte stride (every row store a whole number of lines): 226 us, no gain -- the row
// alignment is not what the stream costs.  Decomposition at 224 us: arithmetic alone 165 us, + packing/storing the valid
// rows 26 us, + zero rows of the invalid slots 33 us.  Zero rows written by separate store-only blocks interleaved in the
// grid (blockIdx.x & 1): 341 us -- the role branch wrecks the code generated for the evaluating path.
// s = d2 - thr2 by ONE packed fma (score scaled by 1/thr2 once per model) instead of the mul + fma of d2/thr2 - 1: eight
// instructions fewer per 16 points, identical masks, and 206.6 vs 198.1 us on the same box (scratch/ab_k4_pair.py; 208 us
// with the addend in an SGPR): slower -- dropped.  Instruction count is not what bounds this loop any more.
// Round 2: non-temporal mask stores 226 vs 208 us (they help a contiguous LDS-assembled stream, not 1 KiB row pieces);
// a persistent grid (one resident wave of blocks, every block five 32-slot tiles of one pair, points loaded once) 218 vs
// 200 us, with 64-slot tiles 247 us: the hardware's dynamic dispatch of 5 120 unequal blocks balances better than equal
// static shares started in lock step (scratch/ab_k4nt.py); register budgets for 3 / 5 / 6 waves per SIMD instead of 4:
// 233 / 221 / 343 us; 32- and 128-slot tiles at 128 pairs per launch: 793 / 757 against 692 us.  The matrix-core
// candidate filter of round 2 (correct, slower: DESIGN 2b) is kept as scratch/k4_filter_kernel.patch.
#include "dr_common.hpp"

// 0: the empty mask rows of the slots a tile does not evaluate are written by a store-only tail after the model loop (rounds 1-4);
// 1 / 2: one (up to two) of them per evaluated model INSIDE the model loop of the 16-points-per-lane kernel, so that the store issues
// under the next model's ~210 vector instructions; what the loop does not get to is drained by the tail.  Round 5, in-step A/B on one
// box, two rounds (profiles/r5_k4_zero_inloop.md): scoring launch 0.6047 / 0.6051 ms (0) -> 0.5937 / 0.5905 (1) -> 0.5942 / 0.5911 (2),
// masks and scores bit-identical.  1 is the default; the pocket the round-4 review priced at 0.054 ms closes by 0.012.
#ifndef DR_K4_ZERO_INLOOP
#define DR_K4_ZERO_INLOOP 1
#endif

namespace dr {

constexpr int kThreads = 256;
constexpr int kPts = 8;                       // points per lane
constexpr int kChunk = kThreads * kPts;       // 2048 points per block pass
constexpr int kModelsPerBlock = 32;

template <typename T>
struct Pt4 { T x1, y1, x2, y2; };

template <typename T>
__device__ __forceinline__ T sampson_s(const T m[9], T x1, T y1, T x2, T y2, T inv_thr2) {
  // a = M^T x2 ; b = M x1 (first two) ; r = x1 . a       (msac_score.py:33-39)
  T a0 = fma(x2, m[0], fma(y2, m[3], m[6]));
  T a1 = fma(x2, m[1], fma(y2, m[4], m[7]));
  T a2 = fma(x2, m[2], fma(y2, m[5], m[8]));
  T b0 = fma(x1, m[0], fma(y1, m[1], m[2]));
  T b1 = fma(x1, m[3], fma(y1, m[4], m[5]));
  T r = fma(x1, a0, fma(y1, a1, a2));
  T jj = fma(a0, a0, fma(a1, a1, fma(b0, b0, b1 * b1)));
  T d2 = (r * r) * fast_rcp(jj);
  return fma(d2, inv_thr2, T(-1));  // s = d2/thr2 - 1 ; inlier <=> s < 0 ; soft score = max(-s, 0)
}

template <typename T, bool kMask>
__global__ __launch_bounds__(kThreads) void msac_score_kernel(const T *__restrict__ matches,
                                                              const T *__restrict__ models,
                                                              const uint8_t *__restrict__ valid,
                                                              const T *__restrict__ thr, int M, int N,
                                                              T *__restrict__ scores, uint8_t *__restrict__ masks,
                                                              int chunks_per_block, int use_atomic) {
  __shared__ T part[kThreads / kWave][kModelsPerBlock];
  const int p = blockIdx.z;
  const int m0 = blockIdx.x * kModelsPerBlock;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int mcount = min(kModelsPerBlock, M - m0);
  const T t = T(1.5) * thr[p];
  const T inv_thr2 = T(1) / (t * t);
  const T *mt = matches + (size_t)p * N * 4;
  const T *md = models + ((size_t)p * M + m0) * 9;
  const bool row_aligned = (N % 8) == 0 && (reinterpret_cast<uintptr_t>(masks) % 8) == 0;

  for (int i = tid; i < (kThreads / kWave) * kModelsPerBlock; i += kThreads) (&part[0][0])[i] = T(0);
  __syncthreads();

  const int c_begin = blockIdx.y * chunks_per_block;
  for (int c = c_begin; c < c_begin + chunks_per_block; ++c) {
    const int n0 = c * kChunk + tid * kPts;
    if (c * kChunk >= N) break;
    T x1[kPts], y1[kPts], x2[kPts], y2[kPts];
#pragma unroll
    for (int j = 0; j < kPts; ++j) {
      const int n = n0 + j;
      if (n < N) {
        if constexpr (sizeof(T) == 4) {
          const float4 v = reinterpret_cast<const float4 *>(mt)[n];
          x1[j] = v.x; y1[j] = v.y; x2[j] = v.z; y2[j] = v.w;
        } else {
          x1[j] = mt[4 * n]; y1[j] = mt[4 * n + 1]; x2[j] = mt[4 * n + 2]; y2[j] = mt[4 * n + 3];
        }
      } else {  // padding point: contributes exactly zero (s = +inf -> max(-s,0) = 0, mask = 0)
        x1[j] = y1[j] = x2[j] = y2[j] = T(0);
      }
    }
    const int nvalid = min(kPts, max(0, N - n0));

    for (int ml = 0; ml < mcount; ++ml) {
      T m[9];
      bool finite = !valid || valid[(size_t)p * M + m0 + ml];   // invalid slots: score 0, empty mask, no arithmetic
      bool nonzero = false;
#pragma unroll
      for (int q = 0; q < 9; ++q) {
        m[q] = md[ml * 9 + q];
        finite = finite && is_finite(m[q]);
        nonzero = nonzero || (m[q] != T(0));
      }
      finite = finite && nonzero;   // all-zero model: like a non-finite one (reference: NaN score, all-false mask)
      T acc = T(0);
      uint32_t lo = 0, hi = 0;
#pragma unroll
      for (int j = 0; j < kPts; ++j) {
        T s = sampson_s<T>(m, x1[j], y1[j], x2[j], y2[j], inv_thr2);
        const bool in = (j < nvalid) && (s < T(0));
        acc += in ? -s : T(0);
        if (kMask) {
          if (j < 4) lo |= (uint32_t)in << (8 * j);
          else hi |= (uint32_t)in << (8 * (j - 4));
        }
      }
      if (!finite) { acc = T(0); lo = hi = 0; }
      if (kMask && nvalid > 0) {
        uint8_t *row = masks + ((size_t)p * M + m0 + ml) * N + n0;
        if (row_aligned && nvalid == kPts) {
          *reinterpret_cast<uint2 *>(row) = make_uint2(lo, hi);
        } else {
          for (int j = 0; j < nvalid; ++j) row[j] = (uint8_t)(((j < 4 ? lo : hi) >> (8 * (j & 3))) & 1u);
        }
      }
      acc = wave_sum(acc);
      if (lane == 0) part[wv][ml] += (finite || (valid && !valid[(size_t)p * M + m0 + ml])) ? acc : (acc + T(NAN));
    }
  }
  __syncthreads();
  if (tid < mcount) {
    T v = part[0][tid] + part[1][tid] + part[2][tid] + part[3][tid];
    T *dst = scores + (size_t)p * M + m0 + tid;
    if (use_atomic) atomicAdd(dst, v);
    else *dst = v;
  }
}

// ---- f32 fast path ---------------------------------------------------------------------------------------------
// Same mapping as above, hand-shaped for the VALU (the binding unit at this shape):
//  * points are processed in PAIRS so that the 16 FMAs per (model, point) become v_pk_fma_f32 (two lanes-worth of
//    FMAs per issue; the model coefficient is an SGPR broadcast to both halves through op_sel);
//  * the next model's nine coefficients are fetched through the scalar cache while the current one is evaluated;
//  * non-finite models are detected from the exponent bits, once per tile;
//  * the soft-score term max(0, 1 - d2/thr2) comes out of ONE packed FMA with the clamp modifier and the mask byte is
//    an exponent bit of that term, packed four at a time with v_perm_b32.
// The variants measured and dropped in rounds 1-3 (non-temporal stores, LDS quad-sum reduction, contiguous zero-run fill,
// real SGPR pairs, hand-placed scalar prefetch, per-half rendezvous, register budgets for 3 / 5 / 6 waves, 32- / 128-slot
// tiles, ...) are kept as scratch/k4_dropped_variants.patch; their numbers are in profiles/r3_k4_counters_p128.md.
typedef float v2f __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// 16-byte store at (wave-uniform 64-bit base) + (per-lane unsigned 32-bit offset): the SGPR-base form of global_store, so that
// the vector ALU computes no address (hipcc otherwise forms base + row * N + n0 per lane with a quarter-rate v_mad_u64_u32)
__device__ __forceinline__ void mask_row_store_saddr(const uint8_t *base_uniform, uint32_t off, uint32_t a, uint32_t b, uint32_t c,
                                                     uint32_t d) {
  const uint64_t bb = reinterpret_cast<uint64_t>(base_uniform);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)bb), hi = __builtin_amdgcn_readfirstlane((uint32_t)(bb >> 32));
  const uint64_t bs = ((uint64_t)hi << 32) | lo;
  const u32x4 v = {a, b, c, d};
  asm volatile("global_store_dwordx4 %0, %1, %2" : : "v"(off), "v"(v), "s"(bs) : "memory");
}

__device__ __forceinline__ v2f splat(float a) { return (v2f){a, a}; }

constexpr int kFastTile = 64;   // model slots per block in the f32 fast paths (two 32-bit validity words)

// ---- f32, 8 points per lane (rows with N % 16 != 0) ---------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void msac_score_kernel_f32_fast(const float *__restrict__ matches,
                                                                       const float *__restrict__ models,
                                                                       const uint8_t *__restrict__ valid,
                                                                       const float *__restrict__ thr, int M, int N,
                                                                       float *__restrict__ scores,
                                                                       uint8_t *__restrict__ masks, int write_masks,
                                                                       int chunks_per_block, int use_atomic) {
  __shared__ float part[kThreads / kWave][kFastTile];
  const int p = blockIdx.z;
  const int m0 = blockIdx.x * kFastTile;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int mcount = min(kFastTile, M - m0);
  const float t = 1.5f * thr[p];
  const float inv_thr2 = 1.0f / (t * t);
  const float *mt = matches + (size_t)p * N * 4;
  const float *md = models + ((size_t)p * M + m0) * 9;
  const bool row_aligned = (N % 8) == 0 && (reinterpret_cast<uintptr_t>(masks) % 8) == 0;
  for (int i = tid; i < (kThreads / kWave) * kFastTile; i += kThreads) (&part[0][0])[i] = 0.f;
  // Validity of the tile's slots as wave-uniform bit masks.  Slots the solver marked invalid (non-real roots: more than
  // half of the ten five-point slots) are never evaluated: the loop below walks the set bits only, so a block's work is
  // its number of VALID models and there is no per-slot skip cost; their mask rows are zero-filled by a store-only loop.
  uint32_t vword[kFastTile / 32];
#pragma unroll
  for (int w = 0; w < kFastTile / 32; ++w) {
    const int ml = 32 * w + (lane & 31);
    const bool inside = ml < mcount;
    const bool v = inside && (!valid || valid[(size_t)p * M + m0 + ml] != 0);
    vword[w] = (uint32_t)(__ballot(v && lane < 32));
  }
  __syncthreads();

  const int c_begin = blockIdx.y * chunks_per_block;
  for (int c = c_begin; c < c_begin + chunks_per_block; ++c) {
    if (c * kChunk >= N) break;
    const int n0 = c * kChunk + tid * kPts;
    v2f x1[kPts / 2], y1[kPts / 2], x2[kPts / 2], y2[kPts / 2], w[kPts / 2];
#pragma unroll
    for (int j = 0; j < kPts; ++j) {
      const int n = n0 + j;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (n < N) v = reinterpret_cast<const float4 *>(mt)[n];
      x1[j / 2][j & 1] = v.x; y1[j / 2][j & 1] = v.y; x2[j / 2][j & 1] = v.z; y2[j / 2][j & 1] = v.w;
      w[j / 2][j & 1] = (n < N) ? 1.f : 0.f;
    }
    const int nvalid = min(kPts, max(0, N - n0));
    uint32_t vlo = 0, vhi = 0;   // byte j = 1 for valid points
#pragma unroll
    for (int j = 0; j < kPts; ++j) {
      if (j < nvalid) { if (j < 4) vlo |= 1u << (8 * j); else vhi |= 1u << (8 * (j - 4)); }
    }

#pragma unroll 1
    for (int wd = 0; wd < kFastTile / 32; ++wd) {
      uint32_t live = vword[wd];
      if (!live) continue;
      int ml = 32 * wd + __builtin_ctz(live);
      live &= live - 1;
      float mc[9];
#pragma unroll
      for (int q = 0; q < 9; ++q) mc[q] = md[ml * 9 + q];
      while (true) {
        float m[9];
#pragma unroll
        for (int q = 0; q < 9; ++q) m[q] = mc[q];
        const int cur = ml;
        const bool more = live != 0;
        if (more) {   // prefetch the next valid model through the scalar cache
          ml = 32 * wd + __builtin_ctz(live);
          live &= live - 1;
#pragma unroll
          for (int q = 0; q < 9; ++q) mc[q] = md[ml * 9 + q];
        }
        // scalar-unit finiteness test: largest exponent field over the nine coefficients
        uint32_t ex = 0, anybit = 0;
#pragma unroll
        for (int q = 0; q < 9; ++q) {
          const uint32_t mb = __builtin_amdgcn_readfirstlane(__float_as_uint(m[q]));
          ex = max(ex, mb & 0x7f800000u);
          anybit |= mb & 0x7fffffffu;
        }
        // an all-zero model is treated like a non-finite one (score NaN, empty mask): the reference's 0/0 gives NaN
        // scores and `NaN < thr` = False masks (msac_score.py:42-48)
        const bool finite = ex != 0x7f800000u && anybit != 0u;

        v2f acc = splat(0.f);
        uint32_t sb[kPts];
#pragma unroll
        for (int j = 0; j < kPts / 2; ++j) {
          const v2f a0 = x2[j] * splat(m[0]) + (y2[j] * splat(m[3]) + splat(m[6]));
          const v2f a1 = x2[j] * splat(m[1]) + (y2[j] * splat(m[4]) + splat(m[7]));
          const v2f a2 = x2[j] * splat(m[2]) + (y2[j] * splat(m[5]) + splat(m[8]));
          const v2f b0 = x1[j] * splat(m[0]) + (y1[j] * splat(m[1]) + splat(m[2]));
          const v2f b1 = x1[j] * splat(m[3]) + (y1[j] * splat(m[4]) + splat(m[5]));
          const v2f r = x1[j] * a0 + (y1[j] * a1 + a2);
          const v2f jj = a0 * a0 + (a1 * a1 + (b0 * b0 + b1 * b1));
          const v2f rr = r * r;
          v2f rc;
          rc[0] = __builtin_amdgcn_rcpf(jj[0]);
          rc[1] = __builtin_amdgcn_rcpf(jj[1]);
          const v2f sv = (rr * rc) * splat(inv_thr2) - splat(1.0f);   // s = d2/thr2 - 1
          sb[2 * j] = __float_as_uint(sv[0]);
          sb[2 * j + 1] = __float_as_uint(sv[1]);
          // min(sv, 0) by one integer instruction each: for IEEE bit patterns min_i32(bits(sv), 0) is sv when the sign bit is
          // set and +0 otherwise (fmaxf costs two -- the compiler canonicalises its operand first); acc holds the negated sum
          v2f mn;
          mn[0] = __int_as_float(min((int)sb[2 * j], 0));
          mn[1] = __int_as_float(min((int)sb[2 * j + 1], 0));
          acc = mn * w[j] + acc;
        }
        float a = -(acc[0] + acc[1]);
        if (write_masks && nvalid > 0) {
          // top bytes of four s values -> one dword, then sign bit -> bit 0 of each byte
          const uint32_t t01 = __builtin_amdgcn_perm(sb[1], sb[0], 0x0c0c0703u);  // bytes: [s0.b3, s1.b3, 0, 0]
          const uint32_t t23 = __builtin_amdgcn_perm(sb[3], sb[2], 0x07030c0cu);  // bytes: [0, 0, s2.b3, s3.b3]
          const uint32_t t45 = __builtin_amdgcn_perm(sb[5], sb[4], 0x0c0c0703u);
          const uint32_t t67 = __builtin_amdgcn_perm(sb[7], sb[6], 0x07030c0cu);
          uint32_t lo = (((t01 | t23) >> 7) & 0x01010101u) & vlo;
          uint32_t hi = (((t45 | t67) >> 7) & 0x01010101u) & vhi;
          if (!finite) { lo = 0; hi = 0; }
          uint8_t *row = masks + ((size_t)p * M + m0 + cur) * N + n0;
          if (row_aligned && nvalid == kPts) {
            *reinterpret_cast<uint2 *>(row) = make_uint2(lo, hi);
          } else {
            for (int j = 0; j < nvalid; ++j) row[j] = (uint8_t)(((j < 4 ? lo : hi) >> (8 * (j & 3))) & 1u);
          }
        }
        a = wave_sum_lane63(a);
        if (lane == 63) part[wv][cur] += finite ? a : NAN;
        if (!more) break;
      }
    }
    // store-only pass after the evaluations: empty mask rows of the invalid slots
    if (write_masks && nvalid > 0) {
#pragma unroll
      for (int wd = 0; wd < kFastTile / 32; ++wd) {
        uint32_t inv = ~vword[wd];
        if (32 * wd + 32 > mcount) inv &= (mcount > 32 * wd) ? ((1u << (mcount - 32 * wd)) - 1u) : 0u;
        while (inv) {
          const int ml = 32 * wd + __builtin_ctz(inv);
          inv &= inv - 1;
          uint8_t *row = masks + ((size_t)p * M + m0 + ml) * N + n0;
          if (row_aligned && nvalid == kPts) *reinterpret_cast<uint2 *>(row) = make_uint2(0u, 0u);
          else
            for (int j = 0; j < nvalid; ++j) row[j] = 0;
        }
      }
    }
  }
  __syncthreads();
  for (int i = tid; i < mcount; i += kThreads) {
    const float v = part[0][i] + part[1][i] + part[2][i] + part[3][i];
    float *dst = scores + (size_t)p * M + m0 + i;
    if (use_atomic) atomicAdd(dst, v);
    else *dst = v;
  }
}

// One point PAIR against one model: the soft-score terms max(0, 1 - d2/thr2) of both points by the clamp modifier of the
// packed FMA ([0, 1]; NaN -> 0): inlier <=> term > 0 <=> bit 5 of its top byte (biased exponent 64..127) is set.
// ms[q] = the model coefficient q in both halves (an SGPR broadcast through op_sel, or a VGPR pair).
__device__ __forceinline__ v2f msac_pair_term(v2f x1, v2f y1, v2f x2, v2f y2, const v2f (&ms)[9], float inv_thr2) {
  const v2f a0 = x2 * ms[0] + (y2 * ms[3] + ms[6]);
  const v2f a1 = x2 * ms[1] + (y2 * ms[4] + ms[7]);
  const v2f a2 = x2 * ms[2] + (y2 * ms[5] + ms[8]);
  const v2f b0 = x1 * ms[0] + (y1 * ms[1] + ms[2]);
  const v2f b1 = x1 * ms[3] + (y1 * ms[4] + ms[5]);
  const v2f r = x1 * a0 + (y1 * a1 + a2);
  const v2f jj = a0 * a0 + (a1 * a1 + (b0 * b0 + b1 * b1));
  const v2f rr = r * r;
  v2f rc;
  rc[0] = __builtin_amdgcn_rcpf(jj[0]);
  rc[1] = __builtin_amdgcn_rcpf(jj[1]);
  const v2f pq = rr * rc;
  const v2f one = splat(1.0f), ith = splat(inv_thr2);
  v2f c;
  asm("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[1,0,0] neg_hi:[1,0,0] clamp" : "=v"(c) : "v"(pq), "v"(ith), "v"(one));
  return c;
}

// four terms -> one dword of 0/1 mask bytes: top bytes gathered by v_perm_b32, (lo | hi) & 0x20202020 as ONE three-input
// boolean op (v_bitop3_b32, truth table 0xA8), then the shift
__device__ __forceinline__ uint32_t msac_mask_word(v2f c01, v2f c23) {
  const uint32_t lo2 = __builtin_amdgcn_perm(__float_as_uint(c01[1]), __float_as_uint(c01[0]), 0x0c0c0703u);
  const uint32_t hi2 = __builtin_amdgcn_perm(__float_as_uint(c23[1]), __float_as_uint(c23[0]), 0x07030c0cu);
  uint32_t bits;
  asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0xa8" : "=v"(bits) : "v"(lo2), "v"(hi2), "s"(0x20202020u));
  return bits >> 5;
}

// Sixteen points of one lane against one model: mask bytes (uint4) + the lane's NEGATED soft-score partial.
__device__ __forceinline__ uint4 msac_eval16(const v2f (&x1)[8], const v2f (&y1)[8], const v2f (&x2)[8], const v2f (&y2)[8],
                                            const v2f (&ms)[9], float inv_thr2, v2f &nacc) {
  v2f c[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    c[j] = msac_pair_term(x1[j], y1[j], x2[j], y2[j], ms, inv_thr2);
    nacc = nacc - c[j];
  }
  return make_uint4(msac_mask_word(c[0], c[1]), msac_mask_word(c[2], c[3]), msac_mask_word(c[4], c[5]), msac_mask_word(c[6], c[7]));
}

// ---- f32 fast path, 16 points per lane ---------------------------------------------------------------------------
// Same algorithm with a lane owning 16 consecutive points (64 VGPRs), 128-thread halves (two waves cover 2048 points):
// every mask row segment is one 16-byte store per lane (1 KiB per wave instruction) -- the mask stream is store-ISSUE
// bound with 8-byte stores.  Requires N % 16 == 0 (otherwise the 8-point kernel above is used).
// A 256-thread block is two independent 128-thread halves, each with its own 64-slot model tile (nothing shared but the
// workgroup slot): a CU holds at most 8 workgroups, i.e. 16 waves of 128-thread blocks = 4 waves per SIMD.
// tile_slots = model slots per half: 64 when the (pair x tile) grid fills the chip; 16 for calls with few pairs (the reference calls the
// path ONE pair at a time, model_cl.py:488-490: 10 240 slots are 80 workgroups of 64-slot halves on 256 CUs -- 34.6 us per
// launch, each wave alone on its SIMD with the scalar-cache round trip of every model exposed; 16-slot halves are 320 workgroups)
constexpr int kH16 = 128, kHalves = 2, kT16 = kH16 * kHalves, kP16 = 16, kChunk16 = kH16 * kP16, kSmallGridTile = 16;

__global__ __launch_bounds__(kT16) void msac_score_kernel_f32_fast16(const float *__restrict__ matches,
                                                                     const float *__restrict__ models,
                                                                     const uint8_t *__restrict__ valid,
                                                                     const float *__restrict__ thr, int M, int N,
                                                                     float *__restrict__ scores,
                                                                     uint8_t *__restrict__ masks, int write_masks,
                                                                     int chunks_per_block, int use_atomic, int tile_slots,
                                                                     PairGate gate) {
  if (gate.closed(blockIdx.z)) return;   // a terminated pair of a multi-round call (block-uniform): scores and masks keep their contents
  // tile_slots <= kTile: the slots a half actually owns (64, or 16 for grids that would not fill the chip); the kernel is the
  // same either way -- a 16-slot half simply has an empty second validity word
  constexpr int kTile = kFastTile, kWords = kTile / 32;
  __shared__ float part[kT16 / kWave][kTile];
  const int p = blockIdx.z;
  // wave-uniform BY CONSTRUCTION (a half is two whole waves) -- and it must look so to the compiler: with `half` in a VGPR the
  // tile's model coefficients stop being scalar loads and the kernel is 40 % slower
  const int half = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / kH16));
  const int m0 = (blockIdx.x * kHalves + half) * tile_slots;
  const int tid = threadIdx.x % kH16, lane = tid & 63, wv = threadIdx.x >> 6;
  const int mcount = max(0, min(tile_slots, M - m0));
  const float t = 1.5f * thr[p];
  const float inv_thr2 = 1.0f / (t * t);
  const float *mt = matches + (size_t)p * N * 4;
  const float *md = models + ((size_t)p * M + m0) * 9;
  for (int i = threadIdx.x; i < (kT16 / kWave) * kTile; i += kT16) (&part[0][0])[i] = 0.f;
  // the first chunk's points are requested BEFORE the model check: its strided model reads, the ballots and the block barrier
  // then run under the latency of the sixteen point loads instead of in front of it
  v2f x1[kP16 / 2], y1[kP16 / 2], x2[kP16 / 2], y2[kP16 / 2];
  {
    const int n0p = blockIdx.y * chunks_per_block * kChunk16 + tid * kP16;
    const bool havep = n0p < N;
#pragma unroll
    for (int j = 0; j < kP16; ++j) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (havep) v = reinterpret_cast<const float4 *>(mt)[n0p + j];
      x1[j / 2][j & 1] = v.x; y1[j / 2][j & 1] = v.y; x2[j / 2][j & 1] = v.z; y2[j / 2][j & 1] = v.w;
    }
  }
  // non-finite / all-zero models of the tile are found here, once (lane l looks at slot l), instead of in every wave of
  // every chunk: they leave the evaluated set (empty mask row like an invalid slot) and their score is NaN
  uint32_t vword[kWords], nanword[kWords];
#pragma unroll
  for (int w = 0; w < kWords; ++w) {
    const int ml = 32 * w + (lane & 31);
    const bool v = (ml < mcount) && (!valid || valid[(size_t)p * M + m0 + ml] != 0);
    uint32_t ex = 0, anybit = 0;
    if (v) {
#pragma unroll
      for (int q = 0; q < 9; ++q) {
        const uint32_t mb = __float_as_uint(md[ml * 9 + q]);
        ex = max(ex, mb & 0x7f800000u);
        anybit |= mb & 0x7fffffffu;
      }
    }
    const bool bad = v && (ex == 0x7f800000u || anybit == 0u);
    vword[w] = (uint32_t)(__ballot(v && !bad && lane < 32));
    nanword[w] = (uint32_t)(__ballot(bad && lane < 32));
  }
  __syncthreads();
  if ((wv & 1) == 0 && lane < 32) {
#pragma unroll
    for (int w = 0; w < kWords; ++w)
      if ((nanword[w] >> lane) & 1u) part[wv][32 * w + lane] = NAN;
  }

  const int c_begin = blockIdx.y * chunks_per_block;
  for (int c = c_begin; c < c_begin + chunks_per_block; ++c) {
    if (c * kChunk16 >= N) break;
    const int n0 = c * kChunk16 + tid * kP16;
    const bool have = n0 < N;   // N % 16 == 0: a lane's 16 points are all inside or all outside
    if (c != c_begin) {
#pragma unroll
      for (int j = 0; j < kP16; ++j) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (have) v = reinterpret_cast<const float4 *>(mt)[n0 + j];
        x1[j / 2][j & 1] = v.x; y1[j / 2][j & 1] = v.y; x2[j / 2][j & 1] = v.z; y2[j / 2][j & 1] = v.w;
      }
    }

#if DR_K4_ZERO_INLOOP
    // the not-evaluated slots of the tile (invalid, or non-finite), as a bit scan carried next to `live`: one of their empty rows
    // is issued per evaluated model, so that the store rides under the next model's ~210 vector instructions
    uint32_t zrow[kWords];
#pragma unroll
    for (int wd = 0; wd < kWords; ++wd) {
      zrow[wd] = ~vword[wd];
      if (32 * wd + 32 > mcount) zrow[wd] &= (mcount > 32 * wd) ? ((1u << (mcount - 32 * wd)) - 1u) : 0u;
    }
#endif
#pragma unroll 1
    for (int wd = 0; wd < kWords; ++wd) {
      uint32_t live = vword[wd];
      if (!live) continue;
      int ml = 32 * wd + __builtin_ctz(live);
      live &= live - 1;
      float mc[9];
#pragma unroll
      for (int q = 0; q < 9; ++q) mc[q] = md[ml * 9 + q];
#if DR_K4_ZERO_INLOOP
      uint32_t zr = zrow[wd];
#endif
      while (true) {
        v2f ms[9];
#pragma unroll
        for (int q = 0; q < 9; ++q) ms[q] = splat(mc[q]);
        const int cur = ml;
        const bool more = live != 0;
        if (more) {   // the next valid model through the scalar cache
          ml = 32 * wd + __builtin_ctz(live);
          live &= live - 1;
#pragma unroll
          for (int q = 0; q < 9; ++q) mc[q] = md[ml * 9 + q];
        }
        v2f nacc = splat(0.f);
        const uint4 q = msac_eval16(x1, y1, x2, y2, ms, inv_thr2, nacc);
        float a = have ? -(nacc[0] + nacc[1]) : 0.f;
        // row base = wave-uniform 64-bit address (scalar ALU), lane part = unsigned 32-bit offset: the store takes the SGPR-base form
        // and the vector ALU computes no address at all (the 64-bit multiply-add per store was a quarter-rate instruction)
        if (write_masks && have) mask_row_store_saddr(masks + ((size_t)p * M + m0 + cur) * N, (uint32_t)n0, q.x, q.y, q.z, q.w);
#if DR_K4_ZERO_INLOOP
        // DR_K4_ZERO_INLOOP = 1: one empty row per evaluated model; 2: a second one while the empty rows outnumber the models left
        for (int rep = 0; rep < DR_K4_ZERO_INLOOP; ++rep) {
          if (zr && (rep == 0 || __builtin_popcount(zr) > __builtin_popcount(live) + 1)) {
            const int mz = 32 * wd + __builtin_ctz(zr);
            zr &= zr - 1;
            if (write_masks && have) mask_row_store_saddr(masks + ((size_t)p * M + m0 + mz) * N, (uint32_t)n0, 0u, 0u, 0u, 0u);
          }
        }
#endif
        a = wave_sum_lane63(a);   // DPP only: 212 vs 229 us with the ds_bpermute butterfly (no reduction at all: 204)
        if (lane == 63) atomicAdd(&part[wv][cur], a);   // ds_add_f32, no return: nothing to wait for (-2 %)
        if (!more) break;
      }
#if DR_K4_ZERO_INLOOP
      zrow[wd] = zr;
#endif
    }
    // empty mask rows of the invalid slots (store-only)
    if (write_masks && have) {
#pragma unroll
      for (int wd = 0; wd < kWords; ++wd) {
#if DR_K4_ZERO_INLOOP
        uint32_t inv = zrow[wd];   // what the model loop did not get to
#else
        uint32_t inv = ~vword[wd];
        if (32 * wd + 32 > mcount) inv &= (mcount > 32 * wd) ? ((1u << (mcount - 32 * wd)) - 1u) : 0u;
#endif
        while (inv) {
          const int ml = 32 * wd + __builtin_ctz(inv);
          inv &= inv - 1;
          mask_row_store_saddr(masks + ((size_t)p * M + m0 + ml) * N, (uint32_t)n0, 0u, 0u, 0u, 0u);
        }
      }
    }
  }
  __syncthreads();
  for (int i = tid; i < mcount; i += kH16) {
    const float v = part[2 * half][i] + part[2 * half + 1][i];
    float *dst = scores + (size_t)p * M + m0 + i;
    if (use_atomic) atomicAdd(dst, v);
    else *dst = v;
  }
}

// ---- f32, short rows (N <= 256): a WAVE per model ------------------------------------------------------------------------
// The 16-points-per-lane mapping above gives a block's 128 lanes to 2048 points: at N = 128 (BASELINE configs[0]: 8-point F,
// 128 correspondences, 64 hypotheses) eight lanes of a block hold all the points, the other 120 idle, and the tile's 64 models
// run one after the other -- 47 us for 2.1 M evaluations, 40x below the kernel's own rate at N = 2000.  Here one wave covers
// the whole row with kPts = 1, 2 or 4 CONSECUTIVE points per lane (smallest kPts with 64 kPts >= N), keeps them in VGPRs, and
// the block's four waves take the tile's models in turn (model index wave-uniform -> coefficients through the scalar cache).
// Every (pair, model) is owned by exactly one wave: the score is stored directly (deterministic, no LDS, no atomics) and the
// mask row is one 1-, 2- or 4-byte store per lane.  Same arithmetic per (model, point) as the generic kernel (sampson_s).
constexpr int kSmallMaxN = 256, kSmallWaves = 4, kSmallPerWave = 4, kSmallTile = kSmallWaves * kSmallPerWave;
template <int kPtsS>
__global__ __launch_bounds__(kSmallWaves * 64) void msac_score_kernel_f32_small(const float *__restrict__ matches,
                                                                               const float *__restrict__ models,
                                                                               const uint8_t *__restrict__ valid,
                                                                               const float *__restrict__ thr, int M, int N,
                                                                               float *__restrict__ scores,
                                                                               uint8_t *__restrict__ masks) {
  const int p = blockIdx.z;
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const float t = 1.5f * thr[p];
  const float inv_thr2 = 1.0f / (t * t);
  const float4 *mt = reinterpret_cast<const float4 *>(matches) + (size_t)p * N;
  const int n0 = lane * kPtsS;
  const int nvalid = min(kPtsS, max(0, N - n0));
  float x1[kPtsS], y1[kPtsS], x2[kPtsS], y2[kPtsS];
#pragma unroll
  for (int j = 0; j < kPtsS; ++j) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (j < nvalid) v = mt[n0 + j];
    x1[j] = v.x; y1[j] = v.y; x2[j] = v.z; y2[j] = v.w;
  }
  // every lane's segment is whole and its row offset naturally aligned (the base pointer too: a caller may pass a slice)
  const bool row_word = (N % kPtsS) == 0 && (reinterpret_cast<uintptr_t>(masks) % kPtsS) == 0;
#pragma unroll 1
  for (int r = 0; r < kSmallPerWave; ++r) {
    const int m = blockIdx.x * kSmallTile + r * kSmallWaves + wv;   // wave-uniform
    if (m >= M) break;
    const size_t slot = (size_t)p * M + m;
    float mc[9];
    uint32_t ex = 0, anybit = 0;
#pragma unroll
    for (int q = 0; q < 9; ++q) {
      mc[q] = models[slot * 9 + q];
      const uint32_t mb = __float_as_uint(mc[q]);
      ex = max(ex, mb & 0x7f800000u);
      anybit |= mb & 0x7fffffffu;
    }
    const bool live = !valid || valid[slot] != 0;                       // invalid slot: score 0, empty mask row
    const bool finite = ex != 0x7f800000u && anybit != 0u;              // non-finite / all-zero model: score NaN, empty row
    float acc = 0.f;
    uint32_t bits = 0;
    if (live && finite) {
#pragma unroll
      for (int j = 0; j < kPtsS; ++j) {
        const float sv = sampson_s<float>(mc, x1[j], y1[j], x2[j], y2[j], inv_thr2);
        const bool in = (j < nvalid) && (sv < 0.f);                     // a 0/0 point (NaN) is no inlier and contributes 0
        acc += in ? -sv : 0.f;
        bits |= (uint32_t)in << (8 * j);
      }
    }
    if (masks && nvalid > 0) {
      uint8_t *row = masks + slot * N + n0;
      if (row_word && kPtsS == 4) *reinterpret_cast<uint32_t *>(row) = bits;
      else if (row_word && kPtsS == 2) *reinterpret_cast<uint16_t *>(row) = (uint16_t)bits;
      else
        for (int j = 0; j < nvalid; ++j) row[j] = (uint8_t)((bits >> (8 * j)) & 1u);
    }
    acc = wave_sum_lane63(acc);
    if (lane == 63) scores[slot] = live ? (finite ? acc : NAN) : 0.f;
  }
}

// ---- K6: per-pair first arg-max over valid, non-NaN scores; recompute the winner's mask -------------
template <typename T>
__global__ __launch_bounds__(kThreads) void select_best_kernel(const T *__restrict__ matches,
                                                               const T *__restrict__ models,
                                                               const uint8_t *__restrict__ valid,
                                                               const T *__restrict__ scores,
                                                               const T *__restrict__ thr, int M, int N,
                                                               int32_t *__restrict__ best_idx,
                                                               T *__restrict__ best_score, T *__restrict__ best_model,
                                                               uint8_t *__restrict__ best_mask,
                                                               int32_t *__restrict__ inliers) {
  __shared__ T s_val[kThreads / kWave];
  __shared__ int s_idx[kThreads / kWave];
  __shared__ int s_cnt[kThreads / kWave];
  const int p = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const T *sc = scores + (size_t)p * M;
  const uint8_t *vd = valid ? valid + (size_t)p * M : nullptr;
  T bv = -INFINITY;
  int bi = 0x7fffffff;
  for (int m = tid; m < M; m += kThreads) {
    T v = sc[m];
    bool ok = (v == v) && (!vd || vd[m]);
    if (ok && (v > bv || (v == bv && m < bi))) { bv = v; bi = m; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    T ov = __shfl_xor(bv, o, 64);
    int oi = __shfl_xor(bi, o, 64);
    if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
  }
  if (lane == 0) { s_val[wv] = bv; s_idx[wv] = bi; }
  __syncthreads();
  bv = s_val[0]; bi = s_idx[0];
#pragma unroll
  for (int w = 1; w < kThreads / kWave; ++w)
    if (s_val[w] > bv || (s_val[w] == bv && s_idx[w] < bi)) { bv = s_val[w]; bi = s_idx[w]; }
  const bool none = bi == 0x7fffffff;
  T m[9];
#pragma unroll
  for (int q = 0; q < 9; ++q) m[q] = none ? T(q % 4 == 0) : models[((size_t)p * M + bi) * 9 + q];
  if (tid == 0) {
    best_idx[p] = none ? -1 : bi;
    best_score[p] = none ? T(0) : bv;
    if (best_model)
      for (int q = 0; q < 9; ++q) best_model[(size_t)p * 9 + q] = m[q];
  }
  const T t = T(1.5) * thr[p];
  const T inv_thr2 = T(1) / (t * t);
  int cnt = 0;
  for (int n = tid; n < N; n += kThreads) {
    const T *q = matches + ((size_t)p * N + n) * 4;
    T s = sampson_s<T>(m, q[0], q[1], q[2], q[3], inv_thr2);
    bool in = !none && (s < T(0));
    if (best_mask) best_mask[(size_t)p * N + n] = in;
    cnt += in;
  }
  cnt = wave_sum(cnt);
  if (lane == 0) s_cnt[wv] = cnt;
  __syncthreads();
  if (tid == 0 && inliers) inliers[p] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
}

// ---- K7 acceptance: score the (few) refit candidates of every pair and keep the best of them if it beats the RANSAC
// result (ransac.py:173-185: `scores = score(points, models); if scores.max() > best_score: best_model, best_score =
// ...`; the mask is left as it is, like there).  One block per pair, a wave per candidate in turn: replaces an MSAC launch,
// select_best and four torch kernels (50 us of device time per call) by one launch.
template <typename T>
__global__ __launch_bounds__(kThreads) void refit_accept_kernel(const T *__restrict__ matches, const T *__restrict__ cand,
                                                               const uint8_t *__restrict__ cvalid, const T *__restrict__ thr,
                                                               int S, int N, T *__restrict__ best_score,
                                                               T *__restrict__ best_model) {
  __shared__ T s_part[kThreads / kWave];
  __shared__ T s_best;
  __shared__ int s_which;
  const int p = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const T t = T(1.5) * thr[p];
  const T inv_thr2 = T(1) / (t * t);
  if (tid == 0) { s_best = best_score[p]; s_which = -1; }
  __syncthreads();
  for (int c = 0; c < S; ++c) {
    if (cvalid && !cvalid[(size_t)p * S + c]) continue;   // block-uniform
    T m[9];
    bool finite = true;
#pragma unroll
    for (int q = 0; q < 9; ++q) {
      m[q] = cand[((size_t)p * S + c) * 9 + q];
      finite = finite && is_finite(m[q]);
    }
    T acc = T(0);
    for (int n = tid; n < N; n += kThreads) {
      const T *q = matches + ((size_t)p * N + n) * 4;
      const T sv = sampson_s<T>(m, q[0], q[1], q[2], q[3], inv_thr2);
      acc += (sv < T(0)) ? -sv : T(0);      // a 0/0 point (NaN) contributes 0, as in the scoring kernel
    }
    acc = wave_sum(acc);
    if (lane == 0) s_part[wv] = acc;
    __syncthreads();
    if (tid == 0) {
      T sc = T(0);
      for (int w = 0; w < kThreads / kWave; ++w) sc += s_part[w];
      if (finite && sc > s_best) { s_best = sc; s_which = c; }   // strict: the first maximum wins, like torch.argmax
    }
    __syncthreads();
  }
  const int w = s_which;
  if (w >= 0) {
    if (tid < 9) best_model[(size_t)p * 9 + tid] = cand[((size_t)p * S + w) * 9 + tid];
    if (tid == 0) best_score[p] = s_best;
  }
}

// ---- K6 fused: arg-max + "is it better" + best mask + adaptive stop, all per-pair state on the device ----------
// (ransac.py:109-144 and adaptive_iteration_number :202-215).  One 1024-thread block per pair.
constexpr int kUpdThreads = 1024;

// sub_models (round 6): 0 / >= M = the M models are ONE batch.  Otherwise they are ceil(M / sub_models) consecutive SUB-BATCHES of
// sub_models models (= B hypotheses each) -- the batches the loop of ransac.py:55-144 would have scored one call after the other --
// and the kernel WALKS them in order with that loop's own rule: stop when iters >= max_iters, arg-max of the sub-batch, better-test,
// best mask / inlier count, new max_iters, iters += B.  State after the launch = state after that many iterations of the loop.
constexpr int kUpdMaxSub = 512;   // sub-batches per launch

template <typename T>
__device__ __forceinline__ void argmax_merge(T &bv, int &bi, T ov, int oi) {
  if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
}

template <typename T>
__global__ __launch_bounds__(kUpdThreads) void ransac_update_kernel(
    const T *__restrict__ matches, const T *__restrict__ models, const uint8_t *__restrict__ valid,
    const T *__restrict__ scores, const T *__restrict__ thr, int M, int N, int B, int k, double confidence,
    double eps, int max_iterations, T *__restrict__ best_score, T *__restrict__ best_model,
    uint8_t *__restrict__ best_mask, int32_t *__restrict__ best_inliers, int32_t *__restrict__ iters,
    double *__restrict__ max_iters, int sub_models) {
  __shared__ T s_val[kUpdThreads / kWave];
  __shared__ int s_idx[kUpdThreads / kWave];
  __shared__ int s_cnt[kUpdThreads / kWave];
  __shared__ T s_sub_val[kUpdMaxSub];
  __shared__ int s_sub_idx[kUpdMaxSub];
  __shared__ double s_mi;
  const int p = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  int it = iters[p];
  double mi = max_iters[p];
  if ((double)it >= mi) return;  // this pair has terminated (uniform across the block)
  const T *sc = scores + (size_t)p * M;
  const uint8_t *vd = valid ? valid + (size_t)p * M : nullptr;
  const int msub = (sub_models > 0 && sub_models < M) ? sub_models : M;
  const int R = (M + msub - 1) / msub;
  constexpr int kWaves = kUpdThreads / kWave;
  T one_val = -INFINITY;
  int one_idx = 0x7fffffff;
  if (R <= kWaves) {
    // few sub-batches (R = 1: the one batch): kWaves / R waves share a sub-batch, every load of the launch in flight at once --
    // two sub-batches of 10 240 scores cost what one costs
    const int wps = kWaves / R, j = wv / wps, part = wv - j * wps;
    T bv = -INFINITY;
    int bi = 0x7fffffff;
    if (j < R) {
      const int lo = j * msub, hi = min(M, lo + msub), step = wps * kWave;
      for (int m0 = lo + part * kWave + lane; m0 < hi; m0 += 4 * step) {
        T v[4];
        bool ok[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int m = m0 + u * step;
          v[u] = m < hi ? sc[m] : T(0);
          ok[u] = m < hi && (!vd || vd[m]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int m = m0 + u * step;
          if (ok[u] && v[u] == v[u] && (v[u] > bv || (v[u] == bv && m < bi))) { bv = v[u]; bi = m; }
        }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) argmax_merge(bv, bi, __shfl_xor(bv, o, 64), __shfl_xor(bi, o, 64));
    }
    if (lane == 0) { s_val[wv] = bv; s_idx[wv] = bi; }
    __syncthreads();
    if (R == 1) {
      // one batch (the common launch): every thread merges the sixteen wave results itself -- no second barrier (round 5's form)
      one_val = s_val[0];
      one_idx = s_idx[0];
#pragma unroll
      for (int w = 1; w < kWaves; ++w) argmax_merge(one_val, one_idx, s_val[w], s_idx[w]);
    } else if (tid < R) {
      bv = s_val[tid * wps]; bi = s_idx[tid * wps];
      for (int w = 1; w < wps; ++w) argmax_merge(bv, bi, s_val[tid * wps + w], s_idx[tid * wps + w]);
      s_sub_val[tid] = bv;
      s_sub_idx[tid] = bi;
    }
  } else {
    // many small sub-batches: a wave per sub-batch (first arg-max over its valid, non-NaN models), four scores in flight per lane
    for (int j = wv; j < R; j += kWaves) {
      const int lo = j * msub, hi = min(M, lo + msub);
      T bv = -INFINITY;
      int bi = 0x7fffffff;
      for (int m0 = lo + lane; m0 < hi; m0 += 4 * kWave) {
        T v[4];
        bool ok[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int m = m0 + u * kWave;
          v[u] = m < hi ? sc[m] : T(0);
          ok[u] = m < hi && (!vd || vd[m]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int m = m0 + u * kWave;
          if (ok[u] && v[u] == v[u] && (v[u] > bv || (v[u] == bv && m < bi))) { bv = v[u]; bi = m; }
        }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) argmax_merge(bv, bi, __shfl_xor(bv, o, 64), __shfl_xor(bi, o, 64));
      if (lane == 0) { s_sub_val[j] = bv; s_sub_idx[j] = bi; }
    }
  }
  if (R > 1) __syncthreads();
  T bs = best_score[p];
  const T t = T(1.5) * thr[p];
  const T inv_thr2 = T(1) / (t * t);
  for (int j = 0; j < R; ++j) {
    if ((double)it >= mi) break;                                 // ransac.py:55 (block-uniform: every thread holds the same it / mi)
    const T bv = R == 1 ? one_val : s_sub_val[j];
    const int bi = R == 1 ? one_idx : s_sub_idx[j];
    const bool have = bi != 0x7fffffff;
    const bool better = have && (bv > bs || it == 0);           // ransac.py:116
    if (better) {
      T m[9];
#pragma unroll
      for (int q = 0; q < 9; ++q) m[q] = models[((size_t)p * M + bi) * 9 + q];
      int cnt = 0;
      for (int n = tid; n < N; n += kUpdThreads) {
        const T *q = matches + ((size_t)p * N + n) * 4;
        const bool in = sampson_s<T>(m, q[0], q[1], q[2], q[3], inv_thr2) < T(0);
        best_mask[(size_t)p * N + n] = in;
        cnt += in;
      }
      cnt = wave_sum(cnt);
      __syncthreads();                                           // (s_cnt / s_mi of the previous sub-batch have been read)
      if (lane == 0) s_cnt[wv] = cnt;
      __syncthreads();
      if (tid == 0) {
        int inl = 0;
#pragma unroll
        for (int w = 0; w < kUpdThreads / kWave; ++w) inl += s_cnt[w];
        best_inliers[p] = inl;
        best_score[p] = bv;
#pragma unroll
        for (int q = 0; q < 9; ++q) best_model[(size_t)p * 9 + q] = m[q];
        // adaptive_iteration_number, ransac.py:202-215
        const double ratio = (double)inl / (double)N;
        const double rk = pow(ratio, (double)k);
        const double prob = 1.0 - rk;
        double nmi = (double)max_iterations;
        // (round 5, measured and dropped: the thread's correspondences requested before the arg-max, twelve scores in flight instead
        //  of four -- 15.9 -> 15.9 / 16.1 us at 128 pairs, 10.1 at one pair; without this f64 pow / log10 tail: 15.1 / 9.2 us)
        if (!(prob >= 1.0 - eps)) nmi = fmax(0.0, log10(1.0 - confidence) / log10(1.0 - rk + eps));
        nmi = fmin((double)max_iterations, nmi);
        max_iters[p] = nmi;
        s_mi = nmi;
      }
      bs = bv;
      if (R > 1) {                                               // the next sub-batch's stop test needs the new bound
        __syncthreads();
        mi = s_mi;
      }
    }
    it += B;
  }
  if (tid == 0) iters[p] = it;
}

// ---- K6 state set-up: the normalised threshold of ransac.py:49-53 and the per-pair test-mode state, one launch ----
// (as torch ops this is ~17 tiny kernels per call: 75 us of the 0.49 ms benchmark step)
template <typename T>
__global__ __launch_bounds__(256) void ransac_init_kernel(const T *__restrict__ K1, const T *__restrict__ K2, int k_stride,
                                                         T threshold, int N, int max_iterations, T *__restrict__ thr,
                                                         T *__restrict__ best_score, T *__restrict__ best_model,
                                                         uint8_t *__restrict__ best_mask,
                                                         int32_t *__restrict__ best_inliers, int32_t *__restrict__ iters,
                                                         double *__restrict__ max_iters, uint64_t *__restrict__ seed_state,
                                                         uint64_t *__restrict__ seeds_out, int n_seeds,
                                                         const float *__restrict__ race_logits, float *__restrict__ race_ws, int P) {
  const int p = blockIdx.x;
  if (threadIdx.x == 0) {
    T th = threshold;
    if (K1) {
      const T *a = K1 + (size_t)p * k_stride, *b = K2 + (size_t)p * k_stride;
      // (K1[0,0] + K1[1,1] + K1[0,0] + K2[1,1]) / 4 -- K1[0,0] twice, K2[0,0] never (ransac.py:52);
      // threshold / f evaluated as torch does for scalar / tensor: reciprocal, then multiply
      const T f = (((a[0] + a[4]) + a[0]) + b[4]) / T(4);
      th = (T(1) / f) * threshold;
    }
    thr[p] = th;
    best_score[p] = T(0);
    best_inliers[p] = 0;
    iters[p] = 0;
    max_iters[p] = (double)max_iterations;
  }
  if (threadIdx.x < 9) best_model[(size_t)p * 9 + threadIdx.x] = (threadIdx.x % 4 == 0) ? T(1) : T(0);
  for (int n = threadIdx.x; n < N; n += blockDim.x) best_mask[(size_t)p * N + n] = 0;
  // round 6: the call's sampler keys from the same launch (dr_seed_next_n's work: one node fewer in a replayed call)
  if (p == 0 && seed_state) seed_next_block(seed_state, seeds_out, n_seeds);
  // ... and the per-pair weights of the one-logarithm sampler (the logits do not change between the rounds of a call)
  if (race_ws) race_weights_block(race_logits, N, P, p, race_ws);
}

template <typename T>
int msac_score_launch(const T *matches, const T *models, const uint8_t *valid, const T *thr, int P, int M, int N,
                      T *scores, uint8_t *masks, hipStream_t st, PairGate gate = PairGate()) {
  constexpr bool kFast = sizeof(T) == 4;
  if constexpr (kFast) {
    if (N <= kSmallMaxN) {   // short rows: a wave per model (BASELINE configs[0])
      const dim3 g((M + kSmallTile - 1) / kSmallTile, 1, P), b(kSmallWaves * 64);
      const float *mt = (const float *)matches, *md = (const float *)models, *th = (const float *)thr;
      if (N <= 64) hipLaunchKernelGGL(msac_score_kernel_f32_small<1>, g, b, 0, st, mt, md, valid, th, M, N, (float *)scores, masks);
      else if (N <= 128) hipLaunchKernelGGL(msac_score_kernel_f32_small<2>, g, b, 0, st, mt, md, valid, th, M, N, (float *)scores, masks);
      else hipLaunchKernelGGL(msac_score_kernel_f32_small<4>, g, b, 0, st, mt, md, valid, th, M, N, (float *)scores, masks);
      return check_launch("msac_score_kernel_f32_small");
    }
  }
  // (16-byte mask stores: whole rows and a 16-byte aligned base -- a caller may pass a slice of a larger buffer)
  const bool fast16 = kFast && (N % 16 == 0) && (reinterpret_cast<uintptr_t>(masks) % 16 == 0);
  // few pairs: 16-slot halves so that the grid covers the chip (one-pair calls)
  const bool small_grid = fast16 && (long)P * ((M + kFastTile * kHalves - 1) / (kFastTile * kHalves)) < 512;   // measured: 10.6 us per pair at 3 pairs (16-slot) against 12.4 at 4 pairs (64-slot)
  // (round 5: 8- and 4-slot halves for one to three pairs measured -- one pair 21.6 / 20.7 / 21.7 us with 16 / 8 / 4 slots, three pairs
  //  32.5 / 33.9 / 44.3: the one-pair launch is not bound by a half's serial models; 16 stays)
  const int tile = fast16 ? (small_grid ? kSmallGridTile : kFastTile) * kHalves : (kFast ? kFastTile : kModelsPerBlock);
  const int tiles = (M + tile - 1) / tile;
  const int chunks = (N + kChunk - 1) / kChunk;   // kChunk16 == kChunk
  // split the point range over blocks only when the (pair x model-tile) grid cannot fill the chip
  int ny = 1;
  const long base = (long)P * tiles;
  if (chunks > 1 && base < 2048) ny = (int)min((long)chunks, (2048 + base - 1) / base);
  const int cpb = (chunks + ny - 1) / ny;
  ny = (chunks + cpb - 1) / cpb;
  const int use_atomic = ny > 1;
  if (use_atomic) {
    if (hipMemsetAsync(scores, 0, sizeof(T) * (size_t)P * M, st) != hipSuccess) return check_launch("memset");
  }
  dim3 grid(tiles, ny, P);
  if constexpr (kFast) {
    if (fast16) {
      hipLaunchKernelGGL(msac_score_kernel_f32_fast16, grid, dim3(kT16), 0, st, (const float *)matches,
                         (const float *)models, valid, (const float *)thr, M, N, (float *)scores, masks, masks ? 1 : 0,
                         cpb, use_atomic, small_grid ? kSmallGridTile : kFastTile, gate);
      return check_launch("msac_score_kernel");
    }
    hipLaunchKernelGGL(msac_score_kernel_f32_fast, grid, dim3(kThreads), 0, st, (const float *)matches,
                       (const float *)models, valid, (const float *)thr, M, N, (float *)scores, masks, masks ? 1 : 0,
                       cpb, use_atomic);
  } else {
    if (masks)
      hipLaunchKernelGGL((msac_score_kernel<T, true>), grid, dim3(kThreads), 0, st, matches, models, valid, thr, M, N,
                         scores, masks, cpb, use_atomic);
    else
      hipLaunchKernelGGL((msac_score_kernel<T, false>), grid, dim3(kThreads), 0, st, matches, models, valid, thr, M, N,
                         scores, masks, cpb, use_atomic);
  }
  return check_launch("msac_score_kernel");
}

}  // namespace dr

extern "C" {

// gate_iters / gate_max_iters (optional; round 6: the `_gated` twin folded in): a round > 1 of a multi-round test-mode call -- the blocks
// of pairs whose iteration counter has reached its bound return at once (their scores / masks keep their contents; dr_ransac_update
// ignores such pairs).  Only the 16-points-per-lane kernel looks at the gate; other shapes simply run.
int dr_msac_score_f32(const float *matches, const float *models, const uint8_t *valid, const float *thr, int P,
                      int M, int N, float *scores, uint8_t *masks, const int32_t *gate_iters, const double *gate_max_iters,
                      void *stream) {
  DR_REQUIRE(P > 0 && M > 0 && N > 0 && P <= 65535, "bad sizes");
  DR_REQUIRE(matches && models && thr && scores, "null pointer");
  DR_REQUIRE((gate_iters == nullptr) == (gate_max_iters == nullptr), "gate: both pointers or neither");
  dr::PairGate gate;
  gate.iters = gate_iters;
  gate.max_iters = gate_max_iters;
  return dr::msac_score_launch<float>(matches, models, valid, thr, P, M, N, scores, masks, (hipStream_t)stream, gate);
}

int dr_msac_score_f64(const double *matches, const double *models, const uint8_t *valid, const double *thr, int P,
                      int M, int N, double *scores, uint8_t *masks, void *stream) {
  DR_REQUIRE(P > 0 && M > 0 && N > 0 && P <= 65535, "bad sizes");
  DR_REQUIRE(matches && models && thr && scores, "null pointer");
  return dr::msac_score_launch<double>(matches, models, valid, thr, P, M, N, scores, masks, (hipStream_t)stream);
}

int dr_select_best_f32(const float *matches, const float *models, const uint8_t *valid, const float *scores,
                       const float *thr, int P, int M, int N, int32_t *best_idx, float *best_score,
                       float *best_model, uint8_t *best_mask, int32_t *inliers, void *stream) {
  DR_REQUIRE(matches && models && scores && thr && best_idx && best_score, "null pointer");
  DR_REQUIRE(P > 0 && M > 0 && N > 0, "bad sizes");
  hipLaunchKernelGGL((dr::select_best_kernel<float>), dim3(P), dim3(dr::kThreads), 0, (hipStream_t)stream, matches,
                     models, valid, scores, thr, M, N, best_idx, best_score, best_model, best_mask, inliers);
  return dr::check_launch("select_best_kernel");
}

int dr_refit_accept_f32(const float *matches, const float *cand, const uint8_t *cand_valid, const float *thr, int P, int S,
                        int N, float *best_score, float *best_model, void *stream) {
  DR_REQUIRE(P > 0 && S > 0 && N > 0, "bad sizes");
  DR_REQUIRE(matches && cand && thr && best_score && best_model, "null pointer");
  hipLaunchKernelGGL((dr::refit_accept_kernel<float>), dim3(P), dim3(dr::kThreads), 0, (hipStream_t)stream, matches, cand,
                     cand_valid, thr, S, N, best_score, best_model);
  return dr::check_launch("refit_accept_kernel");
}

int dr_refit_accept_f64(const double *matches, const double *cand, const uint8_t *cand_valid, const double *thr, int P,
                        int S, int N, double *best_score, double *best_model, void *stream) {
  DR_REQUIRE(P > 0 && S > 0 && N > 0, "bad sizes");
  DR_REQUIRE(matches && cand && thr && best_score && best_model, "null pointer");
  hipLaunchKernelGGL((dr::refit_accept_kernel<double>), dim3(P), dim3(dr::kThreads), 0, (hipStream_t)stream, matches, cand,
                     cand_valid, thr, S, N, best_score, best_model);
  return dr::check_launch("refit_accept_kernel");
}

int dr_ransac_init_f32(const float *K1, const float *K2, int k_stride, double threshold, int P, int N,
                       int max_iterations, float *thr, float *best_score, float *best_model, uint8_t *best_mask,
                       int32_t *best_inliers, int32_t *iters, double *max_iters, uint64_t *seed_state, uint64_t *seeds_out,
                       int n_seeds, const float *race_logits, float *race_ws, void *stream) {
  DR_REQUIRE(P > 0 && N > 0 && (k_stride == 0 || k_stride == 9), "bad sizes");
  DR_REQUIRE((race_logits == nullptr) == (race_ws == nullptr), "race weights: logits and workspace, or neither");
  DR_REQUIRE(thr && best_score && best_model && best_mask && best_inliers && iters && max_iters && (!K1 == !K2),
             "null pointer");
  DR_REQUIRE(!seed_state || (seeds_out && n_seeds >= 1 && n_seeds <= 65536), "seed state: need seeds_out and 1 <= n_seeds <= 65536");
  hipLaunchKernelGGL((dr::ransac_init_kernel<float>), dim3(P), dim3(256), 0, (hipStream_t)stream, K1, K2, k_stride,
                     (float)threshold, N, max_iterations, thr, best_score, best_model, best_mask, best_inliers, iters,
                     max_iters, seed_state, seeds_out, n_seeds, race_logits, race_ws, P);
  return dr::check_launch("ransac_init_kernel");
}

int dr_ransac_init_f64(const double *K1, const double *K2, int k_stride, double threshold, int P, int N,
                       int max_iterations, double *thr, double *best_score, double *best_model, uint8_t *best_mask,
                       int32_t *best_inliers, int32_t *iters, double *max_iters, uint64_t *seed_state, uint64_t *seeds_out,
                       int n_seeds, const float *race_logits, float *race_ws, void *stream) {
  DR_REQUIRE(P > 0 && N > 0 && (k_stride == 0 || k_stride == 9), "bad sizes");
  DR_REQUIRE((race_logits == nullptr) == (race_ws == nullptr), "race weights: logits and workspace, or neither");
  DR_REQUIRE(thr && best_score && best_model && best_mask && best_inliers && iters && max_iters && (!K1 == !K2),
             "null pointer");
  DR_REQUIRE(!seed_state || (seeds_out && n_seeds >= 1 && n_seeds <= 65536), "seed state: need seeds_out and 1 <= n_seeds <= 65536");
  hipLaunchKernelGGL((dr::ransac_init_kernel<double>), dim3(P), dim3(256), 0, (hipStream_t)stream, K1, K2, k_stride,
                     threshold, N, max_iterations, thr, best_score, best_model, best_mask, best_inliers, iters,
                     max_iters, seed_state, seeds_out, n_seeds, race_logits, race_ws, P);
  return dr::check_launch("ransac_init_kernel");
}

int dr_ransac_update_f32(const float *matches, const float *models, const uint8_t *valid, const float *scores,
                         const float *thr, int P, int M, int N, int B, int k, double confidence, double eps,
                         int max_iterations, float *best_score, float *best_model, uint8_t *best_mask,
                         int32_t *best_inliers, int32_t *iters, double *max_iters, int sub_models, void *stream) {
  DR_REQUIRE(matches && models && scores && thr && best_score && best_model && best_mask && best_inliers && iters &&
                 max_iters, "null pointer");
  DR_REQUIRE(P > 0 && M > 0 && N > 0 && B > 0 && k > 0, "bad sizes");
  DR_REQUIRE(sub_models >= 0 && (sub_models == 0 || (M + sub_models - 1) / sub_models <= dr::kUpdMaxSub), "sub-batches per launch");
  hipLaunchKernelGGL((dr::ransac_update_kernel<float>), dim3(P), dim3(dr::kUpdThreads), 0, (hipStream_t)stream,
                     matches, models, valid, scores, thr, M, N, B, k, confidence, eps, max_iterations, best_score,
                     best_model, best_mask, best_inliers, iters, max_iters, sub_models);
  return dr::check_launch("ransac_update_kernel");
}

int dr_ransac_update_f64(const double *matches, const double *models, const uint8_t *valid, const double *scores,
                         const double *thr, int P, int M, int N, int B, int k, double confidence, double eps,
                         int max_iterations, double *best_score, double *best_model, uint8_t *best_mask,
                         int32_t *best_inliers, int32_t *iters, double *max_iters, int sub_models, void *stream) {
  DR_REQUIRE(matches && models && scores && thr && best_score && best_model && best_mask && best_inliers && iters &&
                 max_iters, "null pointer");
  DR_REQUIRE(P > 0 && M > 0 && N > 0 && B > 0 && k > 0, "bad sizes");
  DR_REQUIRE(sub_models >= 0 && (sub_models == 0 || (M + sub_models - 1) / sub_models <= dr::kUpdMaxSub), "sub-batches per launch");
  hipLaunchKernelGGL((dr::ransac_update_kernel<double>), dim3(P), dim3(dr::kUpdThreads), 0, (hipStream_t)stream,
                     matches, models, valid, scores, thr, M, N, B, k, confidence, eps, max_iterations, best_score,
                     best_model, best_mask, best_inliers, iters, max_iters, sub_models);
  return dr::check_launch("ransac_update_kernel");
}

int dr_select_best_f64(const double *matches, const double *models, const uint8_t *valid, const double *scores,
                       const double *thr, int P, int M, int N, int32_t *best_idx, double *best_score,
                       double *best_model, uint8_t *best_mask, int32_t *inliers, void *stream) {
  DR_REQUIRE(matches && models && scores && thr && best_idx && best_score, "null pointer");
  DR_REQUIRE(P > 0 && M > 0 && N > 0, "bad sizes");
  hipLaunchKernelGGL((dr::select_best_kernel<double>), dim3(P), dim3(dr::kThreads), 0, (hipStream_t)stream, matches,
                     models, valid, scores, thr, M, N, best_idx, best_score, best_model, best_mask, inliers);
  return dr::check_launch("select_best_kernel");
}

}  // extern "C"
