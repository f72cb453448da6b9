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
                                                   uint8_t *__restrict__ valid) {
  const int s = blockIdx.x * 64 + threadIdx.x;
  if (s >= Bt) return;
  const T *pts = samples + (size_t)s * n * 6;
  const T *wts = weights ? weights + (size_t)s * n : nullptr;
  double c[6] = {0, 0, 0, 0, 0, 0};
  for (int r = 0; r < n; ++r)
#pragma unroll
    for (int d = 0; d < 6; ++d) c[d] += (double)pts[6 * r + d];
#pragma unroll
  for (int d = 0; d < 6; ++d) c[d] /= (double)n;
  double a0 = 0, a1 = 0;
  double cov[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
  for (int r = 0; r < n; ++r) {
    double dp[3], dq[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      dp[d] = (double)pts[6 * r + d] - c[d];
      dq[d] = (double)pts[6 * r + 3 + d] - c[3 + d];
    }
    a0 += sqrt(dp[0] * dp[0] + dp[1] * dp[1] + dp[2] * dp[2]);
    a1 += sqrt(dq[0] * dq[0] + dq[1] * dq[1] + dq[2] * dq[2]);
    const double w = wts ? (double)wts[r] : 1.0;
    const double w2 = w * w;  // the reference scales both coordinate blocks by the weight (:34-35)
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) cov[i][j] += w2 * dp[i] * dq[j];
  }
  a0 /= (double)n;
  a1 /= (double)n;
  const double sc = (sqrt(3.0) / a0) * (sqrt(3.0) / a1);  // both sides scaled to mean distance sqrt(3) (:37-41)
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
  double n0 = sqrt(u0[0] * u0[0] + u0[1] * u0[1] + u0[2] * u0[2]);
#pragma unroll
  for (int k = 0; k < 3; ++k) u0[k] /= n0;
  const double dt = u0[0] * u1[0] + u0[1] * u1[1] + u0[2] * u1[2];
#pragma unroll
  for (int k = 0; k < 3; ++k) u1[k] -= dt * u0[k];
  double n1 = sqrt(u1[0] * u1[0] + u1[1] * u1[1] + u1[2] * u1[2]);
#pragma unroll
  for (int k = 0; k < 3; ++k) u1[k] /= n1;
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
  if (sout) sout[s] = (T)(a1 / a0);
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
// fetch) is shared by twice as many points; DPP-only wave reduction, LDS atomic for the per-model partial.  Same arithmetic
// per (model, point) as the kernel above, in the same order: identical masks, sums equal to rounding of the reduction order.
// BASELINE config 4 (50 000 points x 2048 models): 80-86 -> see DESIGN.md section 6.
#ifndef DR_K4R_16
#define DR_K4R_16 1
#endif
constexpr int kR16Threads = 128, kR16Pts = 16, kR16Chunk = kR16Threads * kR16Pts;
typedef float v2r __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2r rsplat(float a) { return (v2r){a, a}; }
__global__ __launch_bounds__(kR16Threads) void rigid_residual_kernel_f32_16(const float *__restrict__ pts, const float *__restrict__ models,
                                                                           float threshold, int M, int N, float *__restrict__ res_sum,
                                                                           uint8_t *__restrict__ masks, int chunks_per_block,
                                                                           int use_atomic) {
  __shared__ float part[kRModels];
  const int p = blockIdx.z, m0 = blockIdx.x * kRModels;
  const int tid = threadIdx.x, lane = tid & 63;
  const int mcount = min(kRModels, M - m0);
  const float *pt = pts + (size_t)p * N * 6;
  const float *md = models + ((size_t)p * M + m0) * 16;
  if (tid < kRModels) part[tid] = 0.f;
  __syncthreads();
  const int c_begin = blockIdx.y * chunks_per_block;
  for (int c = c_begin; c < c_begin + chunks_per_block; ++c) {
    if (c * kR16Chunk >= N) break;
    const int n0 = c * kR16Chunk + tid * kR16Pts;
    const bool have = n0 < N;   // N % 16 == 0: a lane's points are all inside or all outside
    // points in PAIRS (component 0 / 1 = points 2 j / 2 j + 1 of the lane): the loop below is written in packed form, so the
    // compiler has no operand pairs to assemble per model (its own vectoriser spent 178 v_mov per model doing that)
    v2r xp[kR16Pts / 2][6];
    {
      float x[kR16Pts * 6];   // 16 points x 6 floats = 24 float4 per lane, contiguous
      // lanes past the end read the row's first points instead (unconditional loads issue back to back; a predicated load
      // each waited for the one before: 24 memory round trips per chunk) -- their results are never stored or summed
      const float4 *src = reinterpret_cast<const float4 *>(pt + (size_t)(have ? n0 : 0) * 6);
#pragma unroll
      for (int v = 0; v < 24; ++v) {
        const float4 w = src[v];
        x[4 * v] = w.x; x[4 * v + 1] = w.y; x[4 * v + 2] = w.z; x[4 * v + 3] = w.w;
      }
#pragma unroll
      for (int j = 0; j < kR16Pts / 2; ++j)
#pragma unroll
        for (int d = 0; d < 6; ++d) xp[j][d] = (v2r){x[12 * j + d], x[12 * j + 6 + d]};
    }
    for (int ml = 0; ml < mcount; ++ml) {
      float m[12];
#pragma unroll
      for (int q = 0; q < 12; ++q) m[q] = md[ml * 16 + q];
      v2r acc2 = (v2r){0.f, 0.f};
      uint32_t wq[4] = {0u, 0u, 0u, 0u};
#pragma unroll
      for (int j = 0; j < kR16Pts / 2; ++j) {
        v2r d2 = (v2r){0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const v2r pred = xp[j][0] * rsplat(m[4 * i]) + (xp[j][1] * rsplat(m[4 * i + 1]) + (xp[j][2] * rsplat(m[4 * i + 2]) + rsplat(m[4 * i + 3])));
          const v2r e = xp[j][3 + i] - pred;
          d2 = e * e + d2;
        }
        acc2 = acc2 + d2;
        wq[j >> 1] |= ((uint32_t)(d2[0] < threshold) | ((uint32_t)(d2[1] < threshold) << 8)) << (16 * (j & 1));
      }
      float acc = acc2[0] + acc2[1];
      if (have) {
        const uint4 q = make_uint4(wq[0], wq[1], wq[2], wq[3]);
        *reinterpret_cast<uint4 *>(masks + ((size_t)p * M + m0 + ml) * N + n0) = q;
      }
      acc = wave_sum_lane63(have ? acc : 0.f);
      if (lane == 63) atomicAdd(&part[ml], acc);
    }
  }
  __syncthreads();
  if (tid < mcount) {
    float *dst = res_sum + (size_t)p * M + m0 + tid;
    if (use_atomic) atomicAdd(dst, part[tid]);
    else *dst = part[tid];
  }
}

template <typename T>
int rigid_residual_launch(const T *pts, const T *models, T threshold, int P, int M, int N, T *res_sum, uint8_t *masks,
                          hipStream_t st) {
  const int tiles = (M + kRModels - 1) / kRModels;
  if constexpr (sizeof(T) == 4) {
    if (DR_K4R_16 && masks && N % 16 == 0 && (reinterpret_cast<uintptr_t>(pts) & 15) == 0) {
      const int chunks = (N + kR16Chunk - 1) / kR16Chunk;
      int ny = 1;
      const long base = (long)P * tiles;
      if (chunks > 1 && base < 4096) ny = (int)min((long)chunks, (4096 + base - 1) / base);
      const int cpb = (chunks + ny - 1) / ny;
      ny = (chunks + cpb - 1) / cpb;
      const int use_atomic = ny > 1;
      if (use_atomic && hipMemsetAsync(res_sum, 0, sizeof(T) * (size_t)P * M, st) != hipSuccess) return check_launch("memset");
      hipLaunchKernelGGL(rigid_residual_kernel_f32_16, dim3(tiles, ny, P), dim3(kR16Threads), 0, st, (const float *)pts,
                         (const float *)models, (float)threshold, M, N, (float *)res_sum, masks, cpb, use_atomic);
      return check_launch("rigid_residual_kernel_f32_16");
    }
  }
  const int chunks = (N + kRChunk - 1) / kRChunk;
  int ny = 1;
  const long base = (long)P * tiles;
  if (chunks > 1 && base < 2048) ny = (int)min((long)chunks, (2048 + base - 1) / base);
  const int cpb = (chunks + ny - 1) / ny;
  ny = (chunks + cpb - 1) / cpb;
  const int use_atomic = ny > 1;
  if (use_atomic && hipMemsetAsync(res_sum, 0, sizeof(T) * (size_t)P * M, st) != hipSuccess)
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

// ---- K5: chosen[p,b] = the valid slot closest (Frobenius) to gt[p] -----------------------------------------
template <typename T>
__global__ void select_closest_kernel(const T *__restrict__ models, const uint8_t *__restrict__ valid,
                                      const T *__restrict__ gt, int B, int S, T *__restrict__ chosen,
                                      int32_t *__restrict__ which) {
  const int p = blockIdx.y;
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const size_t e = (size_t)p * B + b;
  T g[9];
#pragma unroll
  for (int q = 0; q < 9; ++q) g[q] = gt[(size_t)p * 9 + q];
  int best = -1;
  T bd = INFINITY;
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
  which[e] = best;
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

int dr_rigid_residual_f32(const float *pts, const float *models, float threshold, int P, int M, int N,
                          float *res_sum, uint8_t *masks, void *stream) {
  DR_REQUIRE(pts && models && res_sum, "null pointer");
  DR_REQUIRE(P > 0 && M > 0 && N > 0 && P <= 65535, "bad sizes");
  return dr::rigid_residual_launch<float>(pts, models, threshold, P, M, N, res_sum, masks, (hipStream_t)stream);
}
int dr_rigid_residual_f64(const double *pts, const double *models, double threshold, int P, int M, int N,
                          double *res_sum, uint8_t *masks, void *stream) {
  DR_REQUIRE(pts && models && res_sum, "null pointer");
  DR_REQUIRE(P > 0 && M > 0 && N > 0 && P <= 65535, "bad sizes");
  return dr::rigid_residual_launch<double>(pts, models, threshold, P, M, N, res_sum, masks, (hipStream_t)stream);
}

int dr_select_closest_f32(const float *models, const uint8_t *valid, const float *gt, int P, int B, int S,
                          float *chosen, int32_t *which, void *stream) {
  DR_REQUIRE(models && gt && chosen && which, "null pointer");
  DR_REQUIRE(P > 0 && B > 0 && S > 0 && P <= 65535, "bad sizes");
  hipLaunchKernelGGL((dr::select_closest_kernel<float>), dim3((B + 255) / 256, P), dim3(256), 0, (hipStream_t)stream,
                     models, valid, gt, B, S, chosen, which);
  return dr::check_launch("select_closest_kernel");
}
int dr_select_closest_f64(const double *models, const uint8_t *valid, const double *gt, int P, int B, int S,
                          double *chosen, int32_t *which, void *stream) {
  DR_REQUIRE(models && gt && chosen && which, "null pointer");
  DR_REQUIRE(P > 0 && B > 0 && S > 0 && P <= 65535, "bad sizes");
  hipLaunchKernelGGL((dr::select_closest_kernel<double>), dim3((B + 255) / 256, P), dim3(256), 0,
                     (hipStream_t)stream, models, valid, gt, B, S, chosen, which);
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
