// K3r -- rigid-transformation solver (RigidTransformationSVDBasedSolver.estimate_model, rigid…:11-74),
// K4r -- squared residual of rigid models (squared_residual, rigid…:76-89, called at ransac.py:380),
// K5  -- train-mode best-of-S selection (ransac.py:87-96).
#include "solver_common.hpp"

namespace dr {

// ---- K3r: one lane = one sample of n >= 3 correspondences (p, q), f64, registers only ----------------------
template <typename T>
__global__ __launch_bounds__(64) void rigid_kernel(const T *__restrict__ samples, const T *__restrict__ weights,
                                                   int Bt, int n, int flag, T *__restrict__ models,
                                                   T *__restrict__ Rout, T *__restrict__ tout, T *__restrict__ sout,
                                                   uint8_t *__restrict__ valid, const int32_t *__restrict__ gidx = nullptr,
                                                   int gB = 0, int gN = 0, T *__restrict__ zero_sums = nullptr) {
  // gidx != NULL (round 4, K2 fused): `samples` is the pair's correspondence array [P, gN, 6] and row r of sample s is
  // matches[s / gB, gidx[s n + r]] -- the gather launch and the [Bt, n, 6] tensor disappear.  zero_sums != NULL: entry s of the
  // residual sums is cleared here, so that the residual kernel that follows can accumulate without a memset launch.
  const int s = blockIdx.x * 64 + threadIdx.x;
  if (s >= Bt) return;
  if (zero_sums) zero_sums[s] = T(0);
  const T *base = gidx ? samples + (size_t)(s / gB) * gN * 6 : samples + (size_t)s * n * 6;
  const int32_t *gi = gidx ? gidx + (size_t)s * n : nullptr;
  auto rowp = [&](int r) -> const T * { return gi ? base + (size_t)gi[r] * 6 : base + 6 * r; };
  const T *wts = weights ? weights + (size_t)s * n : nullptr;
  double c[6] = {0, 0, 0, 0, 0, 0};
  for (int r = 0; r < n; ++r) {
    const T *pts = rowp(r);
#pragma unroll
    for (int d = 0; d < 6; ++d) c[d] += (double)pts[d];
  }
  const double rn = frcp((double)n);   // (round 6: rcp / rsq + Newton for this kernel's divisions and square roots -- one lane's dependent chain)
#pragma unroll
  for (int d = 0; d < 6; ++d) c[d] *= rn;
  double a0 = 0, a1 = 0;
  double cov[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
  for (int r = 0; r < n; ++r) {
    const T *pts = rowp(r);
    double dp[3], dq[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      dp[d] = (double)pts[d] - c[d];
      dq[d] = (double)pts[3 + d] - c[3 + d];
    }
    a0 += fsqrt(dp[0] * dp[0] + dp[1] * dp[1] + dp[2] * dp[2]);
    a1 += fsqrt(dq[0] * dq[0] + dq[1] * dq[1] + dq[2] * dq[2]);
    const double w = wts ? (double)wts[r] : 1.0;
    const double w2 = w * w;  // the reference scales both coordinate blocks by the weight (:34-35)
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) cov[i][j] += w2 * dp[i] * dq[j];
  }
  a0 *= rn;
  a1 *= rn;
  const double sc = 3.0 * frcp(a0 * a1);  // both sides scaled to mean distance sqrt(3) (:37-41): (sqrt 3 / a0)(sqrt 3 / a1)
  bool ok = true;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      cov[i][j] *= sc;
      ok = ok && is_finite(cov[i][j]);
    }
  // tgt = cov^T cov (flag, the reference default, Q9) or cov^T ; SVD tgt = U S V^T ; R = V U^T with det fix
  double tg[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      if (flag) tg[i][j] = cov[0][i] * cov[0][j] + cov[1][i] * cov[1][j] + cov[2][i] * cov[2][j];
      else tg[i][j] = cov[j][i];
    }
  double ata[3][3], V[3][3], ev[3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) ata[i][j] = tg[0][i] * tg[0][j] + tg[1][i] * tg[1][j] + tg[2][i] * tg[2][j];
  jacobi_eig3(ata, V, ev);
  // two dominant right-singular vectors v0, v1 (columns of V with the largest eigenvalues)
  int i0 = 0, i1 = 1;
  {
    int order[3] = {0, 1, 2};
    if (ev[order[0]] < ev[order[1]]) { int t = order[0]; order[0] = order[1]; order[1] = t; }
    if (ev[order[1]] < ev[order[2]]) { int t = order[1]; order[1] = order[2]; order[2] = t; }
    if (ev[order[0]] < ev[order[1]]) { int t = order[0]; order[0] = order[1]; order[1] = t; }
    i0 = order[0];
    i1 = order[1];
  }
  double v0[3], v1[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    v0[k] = (i0 == 0) ? V[k][0] : (i0 == 1) ? V[k][1] : V[k][2];
    v1[k] = (i1 == 0) ? V[k][0] : (i1 == 1) ? V[k][1] : V[k][2];
  }
  double u0[3], u1[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    u0[k] = tg[k][0] * v0[0] + tg[k][1] * v0[1] + tg[k][2] * v0[2];
    u1[k] = tg[k][0] * v1[0] + tg[k][1] * v1[1] + tg[k][2] * v1[2];
  }
  const double n0 = frsqrt(u0[0] * u0[0] + u0[1] * u0[1] + u0[2] * u0[2]);
#pragma unroll
  for (int k = 0; k < 3; ++k) u0[k] *= n0;
  const double dt = u0[0] * u1[0] + u0[1] * u1[1] + u0[2] * u1[2];
#pragma unroll
  for (int k = 0; k < 3; ++k) u1[k] -= dt * u0[k];
  const double n1 = frsqrt(u1[0] * u1[0] + u1[1] * u1[1] + u1[2] * u1[2]);
#pragma unroll
  for (int k = 0; k < 3; ++k) u1[k] *= n1;
  // third vectors by cross product: V' = [v0 v1 v0xv1], U' = [u0 u1 u0xu1]  => R = V' U'^T is a proper rotation,
  // identical to the reference's V U^T after its det(R) < 0 column flip (:59-62)
  const double v2[3] = {v0[1] * v1[2] - v0[2] * v1[1], v0[2] * v1[0] - v0[0] * v1[2], v0[0] * v1[1] - v0[1] * v1[0]};
  const double u2[3] = {u0[1] * u1[2] - u0[2] * u1[1], u0[2] * u1[0] - u0[0] * u1[2], u0[0] * u1[1] - u0[1] * u1[0]};
  double R[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      R[i][j] = v0[i] * u0[j] + v1[i] * u1[j] + v2[i] * u2[j];
      ok = ok && is_finite(R[i][j]);
    }
  // t_j = c1_j - c0_j * sum_i R_ij   (:66 -- an element-wise product summed over rows, not -R c0)
  double t[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) t[j] = c[3 + j] - c[j] * (R[0][j] + R[1][j] + R[2][j]);
  if (!ok) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      t[i] = 0;
#pragma unroll
      for (int j = 0; j < 3; ++j) R[i][j] = (i == j);
    }
  }
  T *m = models + (size_t)s * 16;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
#pragma unroll
    for (int j = 0; j < 3; ++j) m[4 * i + j] = (T)R[i][j];
    m[4 * i + 3] = (T)t[i];
  }
  m[12] = m[13] = m[14] = T(0);
  m[15] = T(1);
  if (Rout)
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) Rout[(size_t)s * 9 + 3 * i + j] = (T)R[i][j];
  if (tout)
#pragma unroll
    for (int i = 0; i < 3; ++i) tout[(size_t)s * 3 + i] = (T)t[i];
  if (sout) sout[s] = (T)fdiv(a1, a0);
  valid[s] = ok;
}

// ---- K4r: same mapping as the MSAC kernel (lane owns 8 consecutive points in VGPRs; model in SGPRs) ------------
constexpr int kRThreads = 256, kRPts = 8, kRChunk = kRThreads * kRPts, kRModels = 32;

template <typename T, bool kMask>
__global__ __launch_bounds__(kRThreads) void rigid_residual_kernel(const T *__restrict__ pts,
                                                                   const T *__restrict__ models, T threshold, int M,
                                                                   int N, T *__restrict__ res_sum,
                                                                   uint8_t *__restrict__ masks, int chunks_per_block,
                                                                   int use_atomic) {
  __shared__ T part[kRThreads / 64][kRModels];
  const int p = blockIdx.z, m0 = blockIdx.x * kRModels;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int mcount = min(kRModels, M - m0);
  const T *pt = pts + (size_t)p * N * 6;
  const T *md = models + ((size_t)p * M + m0) * 16;
  const bool row_aligned = (N % 8) == 0;
  for (int i = tid; i < (kRThreads / 64) * kRModels; i += kRThreads) (&part[0][0])[i] = T(0);
  __syncthreads();
  const int c_begin = blockIdx.y * chunks_per_block;
  for (int c = c_begin; c < c_begin + chunks_per_block; ++c) {
    if (c * kRChunk >= N) break;
    const int n0 = c * kRChunk + tid * kRPts;
    T x[kRPts][6];
#pragma unroll
    for (int j = 0; j < kRPts; ++j)
#pragma unroll
      for (int d = 0; d < 6; ++d) x[j][d] = (n0 + j < N) ? pt[(size_t)(n0 + j) * 6 + d] : T(0);
    const int nvalid = min(kRPts, max(0, N - n0));
    for (int ml = 0; ml < mcount; ++ml) {
      T m[12];
#pragma unroll
      for (int q = 0; q < 12; ++q) m[q] = md[ml * 16 + q];
      T acc = T(0);
      uint32_t lo = 0, hi = 0;
#pragma unroll
      for (int j = 0; j < kRPts; ++j) {
        T d2 = T(0);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const T pred = fma(m[4 * i], x[j][0], fma(m[4 * i + 1], x[j][1], fma(m[4 * i + 2], x[j][2], m[4 * i + 3])));
          const T e = x[j][3 + i] - pred;
          d2 = fma(e, e, d2);
        }
        const bool live = j < nvalid;
        acc += live ? d2 : T(0);
        const bool in = live && d2 < threshold;
        if (kMask) {
          if (j < 4) lo |= (uint32_t)in << (8 * j);
          else hi |= (uint32_t)in << (8 * (j - 4));
        }
      }
      if (kMask && nvalid > 0) {
        uint8_t *row = masks + ((size_t)p * M + m0 + ml) * N + n0;
        if (row_aligned && nvalid == kRPts) *reinterpret_cast<uint2 *>(row) = make_uint2(lo, hi);
        else
          for (int j = 0; j < nvalid; ++j) row[j] = (uint8_t)(((j < 4 ? lo : hi) >> (8 * (j & 3))) & 1u);
      }
      acc = wave_sum(acc);
      if (lane == 0) part[wv][ml] += acc;
    }
  }
  __syncthreads();
  if (tid < mcount) {
    const T v = part[0][tid] + part[1][tid] + part[2][tid] + part[3][tid];
    T *dst = res_sum + (size_t)p * M + m0 + tid;
    if (use_atomic) atomicAdd(dst, v);
    else *dst = v;
  }
}

// f32, N % 16 == 0, masks wanted: a lane owns SIXTEEN consecutive points (96 VGPRs), 128-thread blocks, so that every mask
// row segment is one 16-byte store per lane (1 KiB per wave instruction) and the per-model bookkeeping (reduction, model
// fetch) is shared by twice as many points; DPP-only wave reduction.  Same arithmetic per (model, point) as the kernel
// above, in the same order: identical masks, sums equal to rounding of the reduction order.
// Round 3 (BASELINE config 4, 50 000 points x 2048 models, one pair; 63 us = 0.20 of the HBM roofline before): the ISA
// of the round-2 loop showed (i) the twelve model coefficients fetched by s_load at the top of every iteration with an
// immediate s_waitcnt -- a scalar-cache round trip per model, exposed at three waves per SIMD; (ii) sixteen v_cmp + v_cndmask
// pairs with s_nop bubbles + twelve v_or for the mask bytes; (iii) the lane-63 LDS atomicAdd expanded by the compiler into a
// readlane loop.  Now: the next model is prefetched into a second SGPR set while the current one is evaluated (as in the MSAC
// kernel), the inlier test is the clamped packed difference max(0, min(1, thr - d2)) whose exponent bit 6 says "> 0"
// (NaN -> 0 = outlier), packed with v_perm_b32 like the MSAC masks (1.25 instructions per point instead of 3.5), plain
// LDS read-modify-write of a wave-private partial, and 16-model tiles (twice as many, half as long blocks: the 3 200 waves
// of a 32-model grid were 3.1 per SIMD -- SIMDs with four waves set the time, the others idled a quarter of it).
#ifndef DR_K4R_16
#define DR_K4R_16 1
#endif
#ifndef DR_K4R_PTS
#define DR_K4R_PTS 8    // points per lane of the packed kernel: 8 (round 5) or 16 (rounds 2-4).  Config 4, same box, in the step
                        // (scratch/runs/r5_gpu_ab.sh, r5_gpu_ac.sh): launch 46.1 -> 43.4 us under rocprofv3 (0.281 -> 0.298 of HBM),
                        // step 0.1335 -> 0.1297 ms; 76 registers = six waves per SIMD instead of three; an occupancy hint of
                        // seven waves: 0.1323, of eight (spills): 0.153
#endif
#ifndef DR_K4R_GROUP8
#define DR_K4R_GROUP8 1  // models per scalar fetch of the 8-point form (1: 0.1296, 2: 0.1300-0.1305, 3: 0.1305, 4: 0.1306, 8: 0.1402 ms):
                         // at six waves per SIMD the scalar-cache round trip of a model hides behind the other waves
#endif
#ifndef DR_K4R_TILE
#define DR_K4R_TILE 0    // > 0: fixed models per block (A/B builds); 0: chosen per launch so that the grid is whole rounds of blocks
#endif
constexpr int kR16Threads = 128, kR16Pts = 16, kR16Chunk = kR16Threads * kR16Pts, kR16MaxTile = 64;
typedef float v2r __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2r rsplat(float a) { return (v2r){a, a}; }
#ifndef DR_K4R_GROUP
#define DR_K4R_GROUP 4   // models whose coefficients are requested from the scalar cache together
#endif
#ifndef DR_K4R_WAVES
#define DR_K4R_WAVES 0   // > 0: register budget pinned to that many waves per SIMD (A/B builds; 4 spills two point registers)
#endif
#if DR_K4R_WAVES > 0
#define DR_K4R_OCC __attribute__((amdgpu_waves_per_eu(DR_K4R_WAVES, DR_K4R_WAVES)))
#else
#define DR_K4R_OCC
#endif
// kPts (round 5): 16 points per lane (96 point registers, 138 in all: three waves per SIMD) or 8 (48 / ~90: five waves per SIMD, an
// 8-byte mask store per lane and model instead of a 16-byte one, the per-model reduction shared by half as many points)
template <int kPts>
__global__ __launch_bounds__(kR16Threads) DR_K4R_OCC void rigid_residual_kernel_f32_pk(const float *__restrict__ pts, const float *__restrict__ models,
                                                                           float threshold, int M, int N, float *__restrict__ res_sum,
                                                                           uint8_t *__restrict__ masks, int chunks_per_block,
                                                                           int use_atomic, int tile) {
  constexpr int kChunk = kR16Threads * kPts;
  __shared__ float part[kR16Threads / 64][kR16MaxTile];
  const int p = blockIdx.z, m0 = blockIdx.x * tile;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int mcount = min(tile, M - m0);
  const float *pt = pts + (size_t)p * N * 6;
  const float *md = models + ((size_t)p * M + m0) * 16;
  if (tid < (kR16Threads / 64) * kR16MaxTile) (&part[0][0])[tid] = 0.f;
  __syncthreads();
  const v2r thr2 = rsplat(threshold);
  const int c_begin = blockIdx.y * chunks_per_block;
  for (int c = c_begin; c < c_begin + chunks_per_block; ++c) {
    if (c * kChunk >= N) break;
    const int n0 = c * kChunk + tid * kPts;
    const bool have = n0 < N;   // N % kPts == 0: a lane's points are all inside or all outside
    // points in PAIRS (component 0 / 1 = points 2 j / 2 j + 1 of the lane): the loop below is written in packed form, so the
    // compiler has no operand pairs to assemble per model (its own vectoriser spent 178 v_mov per model doing that)
    v2r xp[kPts / 2][6];
    {
      float x[kPts * 6];   // kPts points x 6 floats = 24 / 12 float4 per lane, contiguous
      // lanes past the end read the row's first points instead (unconditional loads issue back to back; a predicated load
      // each waited for the one before: 24 memory round trips per chunk) -- their results are never stored or summed
      const float4 *src = reinterpret_cast<const float4 *>(pt + (size_t)(have ? n0 : 0) * 6);
#pragma unroll
      for (int v = 0; v < kPts * 6 / 4; ++v) {
        const float4 w = src[v];
        x[4 * v] = w.x; x[4 * v + 1] = w.y; x[4 * v + 2] = w.z; x[4 * v + 3] = w.w;
      }
#pragma unroll
      for (int j = 0; j < kPts / 2; ++j)
#pragma unroll
        for (int d = 0; d < 6; ++d) xp[j][d] = (v2r){x[12 * j + d], x[12 * j + 6 + d]};
    }
    uint8_t *mrow = masks + ((size_t)p * M + m0) * N;   // wave-uniform row base; the lane part is the 32-bit offset n0
    // models in groups of kGroup: the 12 coefficients of all of them are requested from the scalar cache together (one
    // round trip per group instead of one per model: the compiler puts `s_waitcnt lgkmcnt(0)` right behind any s_load whose
    // destination shares an SGPR pair with a live splat operand, so a hand-written "prefetch the next model" is not one)
    constexpr int kGroup = kPts == 8 ? DR_K4R_GROUP8 : DR_K4R_GROUP;
#pragma unroll 1
    for (int mg = 0; mg < mcount; mg += kGroup) {
      float mm[kGroup][12];
#pragma unroll
      for (int u = 0; u < kGroup; ++u) {
        const int mu = min(mg + u, mcount - 1);
#pragma unroll
        for (int q = 0; q < 12; ++q) mm[u][q] = md[mu * 16 + q];
      }
#pragma unroll
      for (int u = 0; u < kGroup; ++u) {
        const int ml = mg + u;
        const bool live = ml < mcount;    // wave-uniform; a tail slot re-evaluates the tile's last model and drops the result
        const float (&m)[12] = mm[u];
        v2r acc2 = (v2r){0.f, 0.f};
        uint32_t wq[kPts / 4];
#pragma unroll
        for (int g = 0; g < kPts / 4; ++g) {
          uint32_t sb[4];
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int j = 2 * g + h;
            v2r d2 = (v2r){0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 3; ++i) {
              // e = (q_i - t_i) - R_i . p, every step with ONE scalar (SGPR) coefficient operand: no model coefficient has to
              // be copied into a VGPR pair to serve as an FMA addend (three v_mov_b64 and six registers per model otherwise)
              v2r e = xp[j][3 + i] - rsplat(m[4 * i + 3]);
              e = e - xp[j][2] * rsplat(m[4 * i + 2]);
              e = e - xp[j][1] * rsplat(m[4 * i + 1]);
              e = e - xp[j][0] * rsplat(m[4 * i]);
              d2 = e * e + d2;
            }
            acc2 = acc2 + d2;
            // inlier <=> d2 < thr <=> fl(thr - d2) > 0 (a difference of two floats is zero only if they are equal); clamped
            // to [0, 1], NaN -> 0: "positive" = biased exponent >= 64 = bit 5 of the top byte (values below 2^-63 cannot
            // occur as a difference of f32 numbers of the sizes thresholds and squared distances have)
            v2r cl;
            asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1] clamp" : "=v"(cl) : "v"(thr2), "v"(d2));
            sb[2 * h] = __float_as_uint(cl[0]);
            sb[2 * h + 1] = __float_as_uint(cl[1]);
          }
          const uint32_t lo2 = __builtin_amdgcn_perm(sb[1], sb[0], 0x0c0c0703u);      // [s0.b3, s1.b3, 0, 0]
          const uint32_t hi2 = __builtin_amdgcn_perm(sb[3], sb[2], 0x07030c0cu);      // [0, 0, s2.b3, s3.b3]
          wq[g] = ((lo2 | hi2) >> 5) & 0x01010101u;   // top byte = sign, exponent bits 7..1: exponent bit 6 sits at bit 5
        }
        float acc = acc2[0] + acc2[1];
        if (have && live) {
          if constexpr (kPts == 16) *reinterpret_cast<uint4 *>(mrow + (size_t)ml * N + (uint32_t)n0) = make_uint4(wq[0], wq[1], wq[2], wq[3]);
          else *reinterpret_cast<uint2 *>(mrow + (size_t)ml * N + (uint32_t)n0) = make_uint2(wq[0], wq[1]);
        }
        acc = wave_sum_lane63(have ? acc : 0.f);
        if (lane == 63 && live) atomicAdd(&part[wv][ml], acc);   // ds_add_f32, no return: the wave does not wait for its own LDS round trip (the plain += was a ds_read + s_waitcnt + ds_write per model)
      }
    }
  }
  __syncthreads();
  if (tid < mcount) {
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < kR16Threads / 64; ++w) v += part[w][tid];
    float *dst = res_sum + (size_t)p * M + m0 + tid;
    if (use_atomic) atomicAdd(dst, v);
    else *dst = v;
  }
}

template <typename T>
int rigid_residual_launch(const T *pts, const T *models, T threshold, int P, int M, int N, T *res_sum, uint8_t *masks,
                          hipStream_t st, bool sums_zeroed = false) {
  if constexpr (sizeof(T) == 4) {
    // (thresholds below 1e-12 would put `thr - d2` near the bit-6 exponent test's blind spot: general kernel)
    if (DR_K4R_16 && masks && N % 16 == 0 && (reinterpret_cast<uintptr_t>(pts) & 15) == 0 && threshold > T(1e-12)) {
      // One block = (pair, model tile, point chunk); every block of a launch takes the same time (tile x chunk evaluations), so
      // the launch takes ceil(blocks / resident blocks) block times: 3 200 blocks on 1 536 resident ones (138 registers: three
      // waves per SIMD, six 2-wave blocks per CU) are THREE rounds for 2.08 rounds of work.  The tile is therefore chosen per
      // launch: the largest grid that is a whole number of rounds -- at C4 (one pair, 25 chunks of 2048 points) 61 tiles of 34
      // models = 1 525 blocks, one round.
      constexpr int kPts = DR_K4R_PTS;
      constexpr int kChunkL = kR16Threads * kPts;
      static int resident = 0;
      if (!resident) {
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        // resident 2-wave blocks per CU: 16 points = 138 registers = three waves per SIMD; 8 points = 76 registers = six
        resident = (DR_K4R_WAVES > 0 ? 2 * DR_K4R_WAVES : (kPts == 8 ? 12 : 6)) * max(cus, 1);
      }
      const int chunks = (N + kChunkL - 1) / kChunkL;
      int ny = chunks;                      // one chunk per block whenever the row is longer than a chunk (atomics across them)
      int tile = DR_K4R_TILE;
      if (tile <= 0) {
        const long per_tile = (long)P * ny;                       // blocks added by one more tile
        const long min_tiles = (M + kR16MaxTile - 1) / kR16MaxTile;
        long rounds = max(1L, (per_tile * min_tiles + resident - 1) / resident);
        long tiles_t = max(min_tiles, rounds * resident / per_tile);      // as many tiles as fit in `rounds` whole rounds
        tile = (int)((M + tiles_t - 1) / tiles_t);
        tile = min(kR16MaxTile, max(4, (tile + 1) & ~1));
      }
      const int tiles = (M + tile - 1) / tile;
      const int cpb = 1;
      const int use_atomic = ny > 1 || sums_zeroed;   // (pre-zeroed sums are added to, never stored over)
      if (use_atomic && !sums_zeroed && hipMemsetAsync(res_sum, 0, sizeof(T) * (size_t)P * M, st) != hipSuccess) return check_launch("memset");
      hipLaunchKernelGGL((rigid_residual_kernel_f32_pk<kPts>), dim3(tiles, ny, P), dim3(kR16Threads), 0, st, (const float *)pts,
                         (const float *)models, (float)threshold, M, N, (float *)res_sum, masks, cpb, use_atomic, tile);
      return check_launch("rigid_residual_kernel_f32_pk");
    }
  }
  const int tiles = (M + kRModels - 1) / kRModels;
  const int chunks = (N + kRChunk - 1) / kRChunk;
  int ny = 1;
  const long base = (long)P * tiles;
  if (chunks > 1 && base < 2048) ny = (int)min((long)chunks, (2048 + base - 1) / base);
  const int cpb = (chunks + ny - 1) / ny;
  ny = (chunks + cpb - 1) / cpb;
  const int use_atomic = ny > 1 || sums_zeroed;
  if (use_atomic && !sums_zeroed && hipMemsetAsync(res_sum, 0, sizeof(T) * (size_t)P * M, st) != hipSuccess)
    return check_launch("memset");
  dim3 grid(tiles, ny, P);
  if (masks)
    hipLaunchKernelGGL((rigid_residual_kernel<T, true>), grid, dim3(kRThreads), 0, st, pts, models, threshold, M, N,
                       res_sum, masks, cpb, use_atomic);
  else
    hipLaunchKernelGGL((rigid_residual_kernel<T, false>), grid, dim3(kRThreads), 0, st, pts, models, threshold, M, N,
                       res_sum, masks, cpb, use_atomic);
  return check_launch("rigid_residual_kernel");
}

// ---- K6 of the 3-D path: per-pair arg-min of the residual sums, "is it better", best model and best mask on the device ----
// (RANSAC3D's test branch, ransac.py:383-406, is dead code upstream -- SURVEY Q4 -- so the selection rule is the documented
// stand-in: keep the valid model with the smallest residual sum.)  grid = (point slices, pairs): every block repeats the
// arg-min over the M sums (8 KB at C4) and recomputes the winner's mask for ITS slice of the points, so one pair of 50 000
// points is served by many CUs; the state is ping-ponged (best_*_in -> best_*_out) because block 0 cannot overwrite the
// value the other blocks of the pair still compare against.  Replaces ~10 torch launches per round (where / min / gather).
constexpr int kU3Threads = 256;
template <typename T>
__global__ __launch_bounds__(kU3Threads) void ransac3d_update_kernel(
    const T *__restrict__ pts, const T *__restrict__ models, const uint8_t *__restrict__ valid, const T *__restrict__ res,
    T threshold, int M, int N, int pts_per_block, const T *__restrict__ best_res_in, const T *__restrict__ best_model_in,
    T *__restrict__ best_res_out, T *__restrict__ best_model_out, uint8_t *__restrict__ best_mask,
    int32_t *__restrict__ best_idx) {
  __shared__ T s_val[kU3Threads / 64];
  __shared__ int s_idx[kU3Threads / 64];
  const int p = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const T *rs = res + (size_t)p * M;
  const uint8_t *vd = valid ? valid + (size_t)p * M : nullptr;
  // the thread's first two points are requested BEFORE the arg-min (they do not depend on the winner): the kernel is one
  // dependent chain -- sums -> winner -> its model -> points -> mask bytes -- and this takes the longest link out of it
  // (round 5, config 4: 9.1 -> see profiles/r5_kernel_stats_c4.md)
  const int n_begin = blockIdx.x * pts_per_block, n_end = min(N, n_begin + pts_per_block);
  T xa[2][6];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int n = n_begin + tid + u * kU3Threads;
    if (best_mask && n < n_end) {
      const T *x = pts + ((size_t)p * N + n) * 6;
      if constexpr (sizeof(T) == 4) {   // 24-byte records: three 8-byte loads
        const float2 a = *reinterpret_cast<const float2 *>(x), b = *reinterpret_cast<const float2 *>(x + 2),
                     c = *reinterpret_cast<const float2 *>(x + 4);
        xa[u][0] = a.x; xa[u][1] = a.y; xa[u][2] = b.x; xa[u][3] = b.y; xa[u][4] = c.x; xa[u][5] = c.y;
      } else {
#pragma unroll
        for (int q = 0; q < 6; ++q) xa[u][q] = x[q];
      }
    } else {
#pragma unroll
      for (int q = 0; q < 6; ++q) xa[u][q] = T(0);
    }
  }
  T bv = INFINITY;
  int bi = 0x7fffffff;
  for (int m = tid; m < M; m += kU3Threads) {
    const T v = rs[m];
    const bool ok = (!vd || vd[m]) && v == v;
    if (ok && (v < bv || (v == bv && m < bi))) { bv = v; bi = m; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const T ov = __shfl_xor(bv, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ov < bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
  }
  if (lane == 0) { s_val[wv] = bv; s_idx[wv] = bi; }
  __syncthreads();
  bv = s_val[0]; bi = s_idx[0];
#pragma unroll
  for (int w = 1; w < kU3Threads / 64; ++w)
    if (s_val[w] < bv || (s_val[w] == bv && s_idx[w] < bi)) { bv = s_val[w]; bi = s_idx[w]; }
  // best_res_in == NULL: the first round of a call -- no previous state (residual +inf, model = identity, mask = empty), so the
  // driver allocates and fills nothing before its first round (three torch fill / copy launches per call otherwise)
  const bool first = best_res_in == nullptr;
  const T old = first ? T(INFINITY) : best_res_in[p];
  const bool better = bi != 0x7fffffff && bv < old;            // strict: an equal later round does not replace the model
  if (blockIdx.x == 0) {
    if (tid < 16) best_model_out[(size_t)p * 16 + tid] = better ? models[((size_t)p * M + bi) * 16 + tid]
                                                       : (first ? T(tid % 5 == 0 ? 1 : 0) : best_model_in[(size_t)p * 16 + tid]);
    if (tid == 0) {
      best_res_out[p] = better ? bv : old;
      if (best_idx) best_idx[p] = better ? bi : -1;            // winner of THIS round, -1 when the state was kept
    }
  }
  if (!best_mask) return;
  if (!better) {
    if (first) {   // nothing selected in the first round: the mask is defined (empty), not left uninitialised
      for (int n = n_begin + tid; n < n_end; n += kU3Threads) best_mask[(size_t)p * N + n] = 0;
    }
    return;
  }
  T m[12];
#pragma unroll
  for (int q = 0; q < 12; ++q) m[q] = models[((size_t)p * M + bi) * 16 + q];
  auto inlier = [&](const T *x) {
    T d2 = T(0);
#pragma unroll
    for (int i = 0; i < 3; ++i) {     // the operation order of rigid_residual_kernel: the mask equals the winner's row of K4r
      const T pred = fma(m[4 * i], x[0], fma(m[4 * i + 1], x[1], fma(m[4 * i + 2], x[2], m[4 * i + 3])));
      const T e = x[3 + i] - pred;
      d2 = fma(e, e, d2);
    }
    return d2 < threshold;
  };
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int n = n_begin + tid + u * kU3Threads;
    if (n < n_end) best_mask[(size_t)p * N + n] = inlier(xa[u]);
  }
  for (int n = n_begin + tid + 2 * kU3Threads; n < n_end; n += kU3Threads) best_mask[(size_t)p * N + n] = inlier(pts + ((size_t)p * N + n) * 6);
}

template <typename T>
int ransac3d_update_launch(const T *pts, const T *models, const uint8_t *valid, const T *res, T threshold, int P, int M, int N,
                           const T *best_res_in, const T *best_model_in, T *best_res_out, T *best_model_out,
                           uint8_t *best_mask, int32_t *best_idx, hipStream_t st) {
  // enough blocks per pair to spread a long point row over the chip: 512 points each (two per thread, both requested before the
  // arg-min) while the launch stays below ~1024 blocks, longer slices beyond
  int nblk = 1;
  if (best_mask) nblk = max(1, min((N + 511) / 512, max(1, 1024 / P)));
  const int ppb = (N + nblk - 1) / nblk;
  hipLaunchKernelGGL((ransac3d_update_kernel<T>), dim3(nblk, P), dim3(kU3Threads), 0, st, pts, models, valid, res, threshold, M,
                     N, ppb, best_res_in, best_model_in, best_res_out, best_model_out, best_mask, best_idx);
  return check_launch("ransac3d_update_kernel");
}

// ---- K5: chosen[p,b] = the valid slot closest (Frobenius) to gt[p] -----------------------------------------
template <typename T>
__global__ void select_closest_kernel(const T *__restrict__ models, const uint8_t *__restrict__ valid,
                                      const T *__restrict__ gt, int B, int S, T *__restrict__ chosen,
                                      int32_t *__restrict__ which, uint8_t *__restrict__ keep) {
  const int p = blockIdx.y;
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const size_t e = (size_t)p * B + b;
  T g[9];
#pragma unroll
  for (int q = 0; q < 9; ++q) g[q] = gt[(size_t)p * 9 + q];
  int best = -1;
  T bd = INFINITY;
  if (S == 10) {
    // the five-point case, branch-free: all ten slots' loads are independent of the validity flags and of each other (the
    // data-dependent `continue` of the general loop below made the ten slots ten dependent memory round trips: 10.8 us for
    // 32 768 samples, a launch that only half fills the chip)
    T d[10];
    bool ok[10];
#pragma unroll
    for (int s = 0; s < 10; ++s) {
      ok[s] = !valid || valid[e * 10 + s] != 0;
      const T *m = models + (e * 10 + s) * 9;
      T acc = T(0);
#pragma unroll
      for (int q = 0; q < 9; ++q) {
        const T t = m[q] - g[q];
        acc = fma(t, t, acc);
      }
      d[s] = acc;
    }
#pragma unroll
    for (int s = 0; s < 10; ++s)
      if (ok[s] && d[s] < bd) { bd = d[s]; best = s; }
  } else {
    for (int s = 0; s < S; ++s) {
      if (valid && !valid[e * S + s]) continue;
      const T *m = models + (e * S + s) * 9;
      T d = T(0);
#pragma unroll
      for (int q = 0; q < 9; ++q) {
        const T t = m[q] - g[q];
        d = fma(t, t, d);
      }
      if (d < bd) { bd = d; best = s; }
    }
  }
  which[e] = best;
  if (keep) keep[e] = best >= 0;   // the nan_filter of ransac.py:104-106 as a flag (was a torch compare kernel per step)
#pragma unroll
  for (int q = 0; q < 9; ++q) chosen[e * 9 + q] = best >= 0 ? models[(e * S + best) * 9 + q] : T(q % 4 == 0 ? 1 : 0);
}

// backward of K5: the gradient of chosen[p,b] goes to slot which[p,b]; every other slot gets 0 (one launch instead of
// zeros + clamp + compare + mul + scatter)
template <typename T>
__global__ void select_closest_bwd_kernel(const T *__restrict__ grad_chosen, const int32_t *__restrict__ which, size_t total,
                                          int S, T *__restrict__ grad_models) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // one thread per (p, b, s, q)
  if (i >= total) return;
  const int q = (int)(i % 9);
  const size_t es = i / 9;
  const int s = (int)(es % S);
  const size_t e = es / S;
  grad_models[i] = (which[e] == s) ? grad_chosen[e * 9 + q] : T(0);
}

}  // namespace dr

extern "C" {

int dr_solve_rigid_f32(const float *samples, const float *weights, int Bt, int n, int flag, float *models, float *R,
                       float *t, float *scale, uint8_t *valid, void *stream) {
  DR_REQUIRE(samples && models && valid, "null pointer");
  DR_REQUIRE(Bt > 0 && n >= 3, "need Bt > 0 and n >= 3 correspondences per sample");
  hipLaunchKernelGGL((dr::rigid_kernel<float>), dim3((Bt + 63) / 64), dim3(64), 0, (hipStream_t)stream, samples,
                     weights, Bt, n, flag, models, R, t, scale, valid);
  return dr::check_launch("rigid_kernel");
}
int dr_solve_rigid_f64(const double *samples, const double *weights, int Bt, int n, int flag, double *models,
                       double *R, double *t, double *scale, uint8_t *valid, void *stream) {
  DR_REQUIRE(samples && models && valid, "null pointer");
  DR_REQUIRE(Bt > 0 && n >= 3, "need Bt > 0 and n >= 3 correspondences per sample");
  hipLaunchKernelGGL((dr::rigid_kernel<double>), dim3((Bt + 63) / 64), dim3(64), 0, (hipStream_t)stream, samples,
                     weights, Bt, n, flag, models, R, t, scale, valid);
  return dr::check_launch("rigid_kernel");
}

// K2 + K3r in one launch (test mode of the 3-D driver): samples are read through the index sets, and (optionally) the residual
// sums of the round are cleared on the way, so that dr_rigid_residual_acc_f32 needs no memset launch
int dr_solve_rigid_gather_f32(const float *matches, const int32_t *idx, int P, int B, int N, int k, int flag, float *models,
                              float *R, float *t, float *scale, uint8_t *valid, float *zero_sums, void *stream) {
  DR_REQUIRE(matches && idx && models && valid, "null pointer");
  DR_REQUIRE(P > 0 && B > 0 && N > 0 && k >= 3 && (long)P * B < (1l << 31), "need P, B, N > 0 and k >= 3 correspondences per sample");
  const int Bt = P * B;
  hipLaunchKernelGGL((dr::rigid_kernel<float>), dim3((Bt + 63) / 64), dim3(64), 0, (hipStream_t)stream, matches,
                     (const float *)nullptr, Bt, k, flag, models, R, t, scale, valid, idx, B, N, zero_sums);
  return dr::check_launch("rigid_kernel");
}

// dr_rigid_residual_f32 with the sums ADDED to res_sum, which the caller (or dr_solve_rigid_gather_f32) has cleared
// accumulate != 0 (round 6: the `_acc` twin folded in): res_sum holds zeros (dr_solve_rigid_gather_f32 clears it) and the sums are
// ADDED to it -- no memset launch
int dr_rigid_residual_f32(const float *pts, const float *models, float threshold, int P, int M, int N,
                          float *res_sum, uint8_t *masks, int accumulate, void *stream) {
  DR_REQUIRE(pts && models && res_sum, "null pointer");
  DR_REQUIRE(P > 0 && M > 0 && N > 0 && P <= 65535, "bad sizes");
  return dr::rigid_residual_launch<float>(pts, models, threshold, P, M, N, res_sum, masks, (hipStream_t)stream, accumulate != 0);
}
int dr_rigid_residual_f64(const double *pts, const double *models, double threshold, int P, int M, int N,
                          double *res_sum, uint8_t *masks, void *stream) {
  DR_REQUIRE(pts && models && res_sum, "null pointer");
  DR_REQUIRE(P > 0 && M > 0 && N > 0 && P <= 65535, "bad sizes");
  return dr::rigid_residual_launch<double>(pts, models, threshold, P, M, N, res_sum, masks, (hipStream_t)stream);
}

int dr_ransac3d_update_f32(const float *pts, const float *models, const uint8_t *valid, const float *res, float threshold,
                           int P, int M, int N, const float *best_res_in, const float *best_model_in, float *best_res_out,
                           float *best_model_out, uint8_t *best_mask, int32_t *best_idx, void *stream) {
  DR_REQUIRE(pts && models && res && best_res_out && best_model_out, "null pointer");
  DR_REQUIRE((best_res_in == nullptr) == (best_model_in == nullptr), "first round: pass NULL for BOTH input states");
  DR_REQUIRE(best_res_in != best_res_out && best_model_in != best_model_out, "the state is ping-ponged: in and out must differ");
  DR_REQUIRE(P > 0 && M > 0 && N > 0 && P <= 65535, "bad sizes");
  return dr::ransac3d_update_launch<float>(pts, models, valid, res, threshold, P, M, N, best_res_in, best_model_in,
                                           best_res_out, best_model_out, best_mask, best_idx, (hipStream_t)stream);
}
int dr_ransac3d_update_f64(const double *pts, const double *models, const uint8_t *valid, const double *res, double threshold,
                           int P, int M, int N, const double *best_res_in, const double *best_model_in, double *best_res_out,
                           double *best_model_out, uint8_t *best_mask, int32_t *best_idx, void *stream) {
  DR_REQUIRE(pts && models && res && best_res_out && best_model_out, "null pointer");
  DR_REQUIRE((best_res_in == nullptr) == (best_model_in == nullptr), "first round: pass NULL for BOTH input states");
  DR_REQUIRE(best_res_in != best_res_out && best_model_in != best_model_out, "the state is ping-ponged: in and out must differ");
  DR_REQUIRE(P > 0 && M > 0 && N > 0 && P <= 65535, "bad sizes");
  return dr::ransac3d_update_launch<double>(pts, models, valid, res, threshold, P, M, N, best_res_in, best_model_in,
                                            best_res_out, best_model_out, best_mask, best_idx, (hipStream_t)stream);
}

int dr_select_closest_f32(const float *models, const uint8_t *valid, const float *gt, int P, int B, int S,
                          float *chosen, int32_t *which, uint8_t *keep, void *stream) {
  DR_REQUIRE(models && gt && chosen && which, "null pointer");
  DR_REQUIRE(P > 0 && B > 0 && S > 0 && P <= 65535, "bad sizes");
  hipLaunchKernelGGL((dr::select_closest_kernel<float>), dim3((B + 255) / 256, P), dim3(256), 0, (hipStream_t)stream,
                     models, valid, gt, B, S, chosen, which, keep);
  return dr::check_launch("select_closest_kernel");
}
int dr_select_closest_f64(const double *models, const uint8_t *valid, const double *gt, int P, int B, int S,
                          double *chosen, int32_t *which, uint8_t *keep, void *stream) {
  DR_REQUIRE(models && gt && chosen && which, "null pointer");
  DR_REQUIRE(P > 0 && B > 0 && S > 0 && P <= 65535, "bad sizes");
  hipLaunchKernelGGL((dr::select_closest_kernel<double>), dim3((B + 255) / 256, P), dim3(256), 0,
                     (hipStream_t)stream, models, valid, gt, B, S, chosen, which, keep);
  return dr::check_launch("select_closest_kernel");
}
int dr_select_closest_bwd_f32(const float *grad_chosen, const int32_t *which, int P, int B, int S, float *grad_models,
                              void *stream) {
  DR_REQUIRE(P > 0 && B > 0 && S > 0, "bad sizes");
  DR_REQUIRE(grad_chosen && which && grad_models, "null pointer");
  const size_t total = (size_t)P * B * S * 9;
  hipLaunchKernelGGL((dr::select_closest_bwd_kernel<float>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, grad_chosen, which, total, S, grad_models);
  return dr::check_launch("select_closest_bwd_kernel");
}
int dr_select_closest_bwd_f64(const double *grad_chosen, const int32_t *which, int P, int B, int S, double *grad_models,
                              void *stream) {
  DR_REQUIRE(P > 0 && B > 0 && S > 0, "bad sizes");
  DR_REQUIRE(grad_chosen && which && grad_models, "null pointer");
  const size_t total = (size_t)P * B * S * 9;
  hipLaunchKernelGGL((dr::select_closest_bwd_kernel<double>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, grad_chosen, which, total, S, grad_models);
  return dr::check_launch("select_closest_bwd_kernel");
}

}  // extern "C"
