// Experiment: MSAC scores (no masks) with the residual r = x2^T E x1 (K = 9) and jj = |(Ex1)_{0,1}|^2 + |(E^T x2)_{0,1}|^2
// (K = 11, expanded quadratic forms) as f32 MFMA dot products over per-point feature vectors; VALU epilogue only.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// features [P][24][Npad]: k 0..8 = x2_i * x1_j (i = k/3, j = k%3), 9..11 = 0;
// 12..22 = x2^2, x2 y2, y2^2, x2, y2, x1^2, x1 y1, y1^2, x1, y1, 1 ; 23 = 0
extern "C" __global__ void features_kernel(const float *__restrict__ matches, int N, int Npad, float *__restrict__ phi) {
  const int p = blockIdx.y, n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= Npad) return;
  float x1 = 0, y1 = 0, x2 = 0, y2 = 0, one = 0;
  if (n < N) {
    const float4 v = reinterpret_cast<const float4 *>(matches)[(size_t)p * N + n];
    x1 = v.x; y1 = v.y; x2 = v.z; y2 = v.w; one = 1.f;
  }
  float *o = phi + (size_t)p * 24 * Npad + n;
  const float X1[3] = {x1, y1, one}, X2[3] = {x2, y2, one};
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) o[(3 * i + j) * Npad] = X2[i] * X1[j];
  o[9 * Npad] = 0; o[10 * Npad] = 0; o[11 * Npad] = 0;
  const float f[12] = {x2 * x2, x2 * y2, y2 * y2, x2, y2, x1 * x1, x1 * y1, y1 * y1, x1, y1, one, 0.f};
  for (int k = 0; k < 12; ++k) o[(12 + k) * Npad] = f[k];
}

// coefficients [P][M][24] matching the feature order
extern "C" __global__ void coeffs_kernel(const float *__restrict__ models, size_t total, float *__restrict__ coef) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const float *m = models + i * 9;
  float *c = coef + i * 24;
  for (int k = 0; k < 9; ++k) c[k] = m[k];
  c[9] = c[10] = c[11] = 0.f;
  // a = M^T x2 (first two): a0 = m0 x2 + m3 y2 + m6, a1 = m1 x2 + m4 y2 + m7
  c[12] = m[0] * m[0] + m[1] * m[1];
  c[13] = 2.f * (m[0] * m[3] + m[1] * m[4]);
  c[14] = m[3] * m[3] + m[4] * m[4];
  c[15] = 2.f * (m[0] * m[6] + m[1] * m[7]);
  c[16] = 2.f * (m[3] * m[6] + m[4] * m[7]);
  // b = M x1 (first two): b0 = m0 x1 + m1 y1 + m2, b1 = m3 x1 + m4 y1 + m5
  c[17] = m[0] * m[0] + m[3] * m[3];
  c[18] = 2.f * (m[0] * m[1] + m[3] * m[4]);
  c[19] = m[1] * m[1] + m[4] * m[4];
  c[20] = 2.f * (m[0] * m[2] + m[3] * m[5]);
  c[21] = 2.f * (m[1] * m[2] + m[4] * m[5]);
  c[22] = m[6] * m[6] + m[7] * m[7] + m[2] * m[2] + m[5] * m[5];
  c[23] = 0.f;
}

// block = 256 threads = 4 waves, kMT tiles of 16 models of one pair (B operands are reused across them); wave w takes
// point tiles w, w+4, ...
constexpr int kMT = 4;
template <int mode>
__global__ __launch_bounds__(256) void score_mfma_kernel(const float *__restrict__ phi, const float *__restrict__ coef,
                                                                   const float *__restrict__ thr, int M, int Npad,
                                                                   float *__restrict__ scores, int) {
  __shared__ float part[4][16 * kMT];
  const int p = blockIdx.y, m0 = blockIdx.x * 16 * kMT;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const float t = 1.5f * thr[p];
  const float inv_thr2 = 1.0f / (t * t);
  float A[kMT][6];
#pragma unroll
  for (int mt = 0; mt < kMT; ++mt)
#pragma unroll
    for (int s = 0; s < 6; ++s) {
      const int m = m0 + 16 * mt + li;
      A[mt][s] = (m < M) ? coef[((size_t)p * M + m) * 24 + 4 * s + lg] : 0.f;
    }
  const float *ph = phi + (size_t)p * 24 * Npad;
  const int tiles = Npad / 16;
  float acc[kMT][4];
#pragma unroll
  for (int mt = 0; mt < kMT; ++mt)
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[mt][q] = 0.f;
  float Bn[6];
#pragma unroll
  for (int s = 0; s < 6; ++s) Bn[s] = wv < tiles ? ph[(size_t)(4 * s + lg) * Npad + 16 * wv + li] : 0.f;
  for (int tt = wv; tt < tiles; tt += 4) {
    float B[6];
#pragma unroll
    for (int s = 0; s < 6; ++s) B[s] = Bn[s];
    if (tt + 4 < tiles) {
#pragma unroll
      for (int s = 0; s < 6; ++s) Bn[s] = ph[(size_t)(4 * s + lg) * Npad + 16 * (tt + 4) + li];
    }
#pragma unroll
    for (int mt = 0; mt < kMT; ++mt) {
      f32x4 r = {0.f, 0.f, 0.f, 0.f}, jj = {0.f, 0.f, 0.f, 0.f};
      if (mode == 2) {
        r = (f32x4){B[0], B[1], B[2], B[3]}; jj = (f32x4){B[4], B[5], A[mt][0], A[mt][1]};
      } else {
        r = __builtin_amdgcn_mfma_f32_16x16x4f32(A[mt][0], B[0], r, 0, 0, 0);
        jj = __builtin_amdgcn_mfma_f32_16x16x4f32(A[mt][3], B[3], jj, 0, 0, 0);
        r = __builtin_amdgcn_mfma_f32_16x16x4f32(A[mt][1], B[1], r, 0, 0, 0);
        jj = __builtin_amdgcn_mfma_f32_16x16x4f32(A[mt][4], B[4], jj, 0, 0, 0);
        r = __builtin_amdgcn_mfma_f32_16x16x4f32(A[mt][2], B[2], r, 0, 0, 0);
        jj = __builtin_amdgcn_mfma_f32_16x16x4f32(A[mt][5], B[5], jj, 0, 0, 0);
      }
      if (mode == 1) {
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[mt][q] += r[q] + jj[q];
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float d2 = r[q] * r[q] * __builtin_amdgcn_rcpf(jj[q]);
          const float sv = fmaf(d2, inv_thr2, -1.0f);
          acc[mt][q] += __int_as_float(min(__float_as_int(sv), 0));
        }
      }
    }
  }
#pragma unroll
  for (int mt = 0; mt < kMT; ++mt)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float v = acc[mt][q];
      for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
      if (li == 0) part[wv][16 * mt + 4 * lg + q] = -v;
    }
  __syncthreads();
  if (threadIdx.x < 16 * kMT && m0 + threadIdx.x < M)
    scores[(size_t)p * M + m0 + threadIdx.x] = part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
}

extern "C" int run_features(const float *matches, int P, int N, int Npad, float *phi, void *st) {
  hipLaunchKernelGGL(features_kernel, dim3((Npad + 255) / 256, P), dim3(256), 0, (hipStream_t)st, matches, N, Npad, phi);
  return (int)hipGetLastError();
}
extern "C" int run_coeffs(const float *models, int P, int M, float *coef, void *st) {
  const size_t total = (size_t)P * M;
  hipLaunchKernelGGL(coeffs_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)st, models, total, coef);
  return (int)hipGetLastError();
}
extern "C" int run_score(const float *phi, const float *coef, const float *thr, int P, int M, int Npad, float *scores, void *st, int mode) {
  dim3 g((M + 16 * kMT - 1) / (16 * kMT), P);
  if (mode == 0) hipLaunchKernelGGL(score_mfma_kernel<0>, g, dim3(256), 0, (hipStream_t)st, phi, coef, thr, M, Npad, scores, mode);
  if (mode == 1) hipLaunchKernelGGL(score_mfma_kernel<1>, g, dim3(256), 0, (hipStream_t)st, phi, coef, thr, M, Npad, scores, mode);
  if (mode == 2) hipLaunchKernelGGL(score_mfma_kernel<2>, g, dim3(256), 0, (hipStream_t)st, phi, coef, thr, M, Npad, scores, mode);
  return (int)hipGetLastError();
}
