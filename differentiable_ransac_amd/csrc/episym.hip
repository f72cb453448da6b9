// SURVEY 8(f) rank 2 -- the training loss that sits right after the hot path: MatchLoss's clamped symmetric epipolar
// error of EVERY returned model on the ground-truth inlier points (loss.py:107-153, cv_utils.batch_episym :680-695):
//     ys[m,n] = (x2^T M x1)^2 * ( 1/((Mx1)_0^2 + (Mx1)_1^2 + 1e-15) + 1/((M^T x2)_0^2 + (M^T x2)_1^2 + 1e-15) )
//     sums[m] = sum_{n in mask} min(ys[m,n], 1)
// The reference materialises [M, n_in, 3, 3] repeats of the models; here it is the same (model x point) grid as K4:
// points in VGPRs, model coefficients wave-uniform, one partial per (wave, model).  Forward and backward.
#include "dr_common.hpp"

namespace dr {

constexpr int kET = 256, kEP = 8, kEChunk = kET * kEP, kEM = 16;

template <bool kBackward>
__global__ __launch_bounds__(kET) void episym_kernel(const float *__restrict__ matches, const uint8_t *__restrict__ mask,
                                                     const float *__restrict__ models, const uint8_t *__restrict__ valid,
                                                     const float *__restrict__ grad_sums, int M, int N,
                                                     float *__restrict__ out) {
  // forward: out = sums [P,M]; backward: out = grad_models [P,M,9]
  constexpr int kV = kBackward ? 9 : 1;
  __shared__ float part[kET / 64][kEM][kV];
  const int p = blockIdx.z, m0 = blockIdx.x * kEM;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int mcount = min(kEM, M - m0);
  for (int i = tid; i < (kET / 64) * kEM * kV; i += kET) (&part[0][0][0])[i] = 0.f;
  __syncthreads();
  const float *mt = matches + (size_t)p * N * 4;
  const uint8_t *mk = mask ? mask + (size_t)p * N : nullptr;
  for (int c0 = 0; c0 < N; c0 += kEChunk) {
    const int n0 = c0 + tid * kEP;
    float x1[kEP], y1[kEP], x2[kEP], y2[kEP], w[kEP];
#pragma unroll
    for (int j = 0; j < kEP; ++j) {
      const int n = n0 + j;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (n < N) v = reinterpret_cast<const float4 *>(mt)[n];
      x1[j] = v.x; y1[j] = v.y; x2[j] = v.z; y2[j] = v.w;
      w[j] = (n < N && (!mk || mk[n])) ? 1.f : 0.f;
    }
    for (int ml = 0; ml < mcount; ++ml) {
      if (valid && !valid[(size_t)p * M + m0 + ml]) continue;
      float m[9];
#pragma unroll
      for (int q = 0; q < 9; ++q) m[q] = models[((size_t)p * M + m0 + ml) * 9 + q];
      float acc[kV];
#pragma unroll
      for (int q = 0; q < kV; ++q) acc[q] = 0.f;
#pragma unroll
      for (int j = 0; j < kEP; ++j) {
        const float a0 = x2[j] * m[0] + y2[j] * m[3] + m[6], a1 = x2[j] * m[1] + y2[j] * m[4] + m[7];
        const float a2 = x2[j] * m[2] + y2[j] * m[5] + m[8];
        const float b0 = x1[j] * m[0] + y1[j] * m[1] + m[2], b1 = x1[j] * m[3] + y1[j] * m[4] + m[5];
        const float r = x1[j] * a0 + y1[j] * a1 + a2;
        const float ib = 1.0f / (b0 * b0 + b1 * b1 + 1e-15f), ia = 1.0f / (a0 * a0 + a1 * a1 + 1e-15f);
        const float ys = r * r * (ib + ia);
        if (!kBackward) {
          acc[0] += w[j] * fminf(ys, 1.0f);
        } else {
          const float live = (ys < 1.0f) ? w[j] : 0.f;   // the clamp passes no gradient at or above 1
          const float c1 = 2.f * r * (ib + ia) * live;   // d ys / d r
          const float cb = 2.f * r * r * ib * ib * live; // -(d ys / d B)/... folded: d ys = c1 dr - cb (b0 db0 + b1 db1) - ca (...)
          const float ca = 2.f * r * r * ia * ia * live;
          const float X2[3] = {x2[j], y2[j], 1.f}, X1[3] = {x1[j], y1[j], 1.f};
          const float bb[3] = {b0, b1, 0.f}, aa[3] = {a0, a1, 0.f};
#pragma unroll
          for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int jx = 0; jx < 3; ++jx)
              acc[3 * i + jx] += c1 * X2[i] * X1[jx] - cb * bb[i] * X1[jx] - ca * aa[jx] * X2[i];
        }
      }
#pragma unroll
      for (int q = 0; q < kV; ++q) {
        const float v = wave_sum(acc[q]);
        if (lane == 0) part[wv][ml][q] += v;
      }
    }
  }
  __syncthreads();
  for (int i = tid; i < mcount * kV; i += kET) {
    const int ml = i / kV, q = i % kV;
    float v = part[0][ml][q] + part[1][ml][q] + part[2][ml][q] + part[3][ml][q];
    if (kBackward) v *= grad_sums[(size_t)p * M + m0 + ml];
    out[((size_t)p * M + m0 + ml) * kV + q] = v;
  }
}

}  // namespace dr

extern "C" {

int dr_episym_fwd_f32(const float *matches, const uint8_t *mask, const float *models, const uint8_t *valid, int P, int M,
                      int N, float *sums, void *stream) {
  DR_REQUIRE(matches && models && sums, "null pointer");
  DR_REQUIRE(P > 0 && M > 0 && N > 0 && P <= 65535, "bad sizes");
  hipLaunchKernelGGL((dr::episym_kernel<false>), dim3((M + dr::kEM - 1) / dr::kEM, 1, P), dim3(dr::kET), 0,
                     (hipStream_t)stream, matches, mask, models, valid, (const float *)nullptr, M, N, sums);
  return dr::check_launch("episym_kernel");
}

int dr_episym_bwd_f32(const float *matches, const uint8_t *mask, const float *models, const uint8_t *valid,
                      const float *grad_sums, int P, int M, int N, float *grad_models, void *stream) {
  DR_REQUIRE(matches && models && grad_sums && grad_models, "null pointer");
  DR_REQUIRE(P > 0 && M > 0 && N > 0 && P <= 65535, "bad sizes");
  hipLaunchKernelGGL((dr::episym_kernel<true>), dim3((M + dr::kEM - 1) / dr::kEM, 1, P), dim3(dr::kET), 0,
                     (hipStream_t)stream, matches, mask, models, valid, grad_sums, M, N, grad_models);
  return dr::check_launch("episym_kernel");
}

}  // extern "C"
