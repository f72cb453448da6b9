// K3f8 / K3f7 -- fundamental-matrix solvers.
//   8-point / LSQ: FundamentalMatrixEstimatorNew.normalize + estimate_non_minimal_model,
//                  fundamental_matrix_estimator.py:177-260 (taken whenever n > 7)
//   7-point:       the mathematically correct algorithm; the reference's two
//                  implementations are degenerate (Q7/Q8), the output contract (4 slots per
//                  sample, invalid -> eye(3)) is FundamentalMatrixEstimatorNew's (:303-308).
// One lane = one sample, f64.  n == 8: the null vector of the 8x9 system by Householder QR in
// registers; n > 8 (inlier refit / LSQ): smallest eigenvector of A^T A by cyclic Jacobi in LDS.
#include "solver_common.hpp"

namespace dr {

constexpr int kF8Ws = 162;  // A^T A (81) + eigenvectors (81)

// kUniform (round 4): K1u + K2 + K3f8 in ONE launch for the minimal 8-point sample -- the lane draws its eight indices itself
// (the Philox counters of uniform_sample_kernel: element j, hypothesis b, pair p, stream 1), reads the eight correspondences
// straight from matches [P,N,4] and writes the index set next to the model.  BASELINE configs[0] is launch-bound (six launches
// of a few microseconds): two of them disappear.
template <typename T, bool kUniform>
__global__ __launch_bounds__(64) void f8_kernel(const T *__restrict__ samples, const T *__restrict__ weights, int Bt,
                                                int n_rt, T *__restrict__ models, uint8_t *__restrict__ valid, uint64_t seed = 0,
                                                const uint64_t *__restrict__ seed_ptr = nullptr, int B = 0, int N = 0,
                                                int32_t *__restrict__ idx_out = nullptr) {
  extern __shared__ __align__(16) double lds[];
  const int lane = threadIdx.x;
  const int s = blockIdx.x * 64 + lane;
  const bool active = s < Bt;
  const int sc = active ? s : Bt - 1;
  const int n = kUniform ? 8 : n_rt;
  const T *pts = kUniform ? nullptr : samples + (size_t)sc * n * 4;
  const T *wts = weights ? weights + (size_t)sc * n : nullptr;
  T loc[kUniform ? 8 : 1][4];
  if constexpr (kUniform) {
    if (seed_ptr) seed = *seed_ptr;
    const int p = sc / B, b = sc % B;
    const uint32_t span = (uint32_t)max(N - 1, 1);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      uint32_t r[4];
      Philox::gen(seed, (uint32_t)j, (uint32_t)b, (uint32_t)p, 1u, r);
      const int32_t i = (int32_t)(((uint64_t)r[0] * span) >> 32);
      if (active && idx_out) idx_out[(size_t)sc * 8 + j] = i;
      const T *row = samples + ((size_t)p * N + i) * 4;
#pragma unroll
      for (int d = 0; d < 4; ++d) loc[j][d] = row[d];
    }
  }
  constexpr int kUn = kUniform ? 8 : 1;   // the fused instantiation indexes registers: its row loops are unrolled
  auto P = [&](int r, int d) -> double { return kUniform ? (double)loc[kUniform ? r : 0][d] : (double)pts[4 * r + d]; };
  // Hartley normalisation (fundamental…:177-217): centroid, mean distance sqrt(2) per image
  double mu[4] = {0, 0, 0, 0};
#pragma unroll kUn
  for (int r = 0; r < n; ++r)
#pragma unroll
    for (int d = 0; d < 4; ++d) mu[d] += P(r, d);
#pragma unroll
  for (int d = 0; d < 4; ++d) mu[d] /= (double)n;
  double d1 = 0, d2 = 0;
#pragma unroll kUn
  for (int r = 0; r < n; ++r) {
    const double a = P(r, 0) - mu[0], b = P(r, 1) - mu[1];
    const double c = P(r, 2) - mu[2], d = P(r, 3) - mu[3];
    d1 += sqrt(a * a + b * b);
    d2 += sqrt(c * c + d * d);
  }
  const double r1 = M_SQRT2 / (d1 / (double)n), r2 = M_SQRT2 / (d2 / (double)n);
  double f[9];
  if (n == 8) {
    double A[8][9];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const double w = wts ? (double)wts[r] : 1.0;
      epipolar_row_f((P(r, 0) - mu[0]) * r1, (P(r, 1) - mu[1]) * r1, (P(r, 2) - mu[2]) * r2, (P(r, 3) - mu[3]) * r2, w, A[r]);
    }
    double nb[1][9];
    null_space_qr<8>(A, nb);
#pragma unroll
    for (int q = 0; q < 9; ++q) f[q] = nb[0][q];
  } else if constexpr (!kUniform) {
    LaneWs A{lds + lane}, V{lds + lane + 81 * 64};
    for (int e = 0; e < 81; ++e) A[e] = 0.0;
    for (int r = 0; r < n; ++r) {
      double row[9];
      const double w = wts ? (double)wts[r] : 1.0;
      epipolar_row_f(((double)pts[4 * r] - mu[0]) * r1, ((double)pts[4 * r + 1] - mu[1]) * r1,
                     ((double)pts[4 * r + 2] - mu[2]) * r2, ((double)pts[4 * r + 3] - mu[3]) * r2, w, row);
#pragma unroll
      for (int i = 0; i < 9; ++i)
#pragma unroll
        for (int j = 0; j < 9; ++j) A[i * 9 + j] += row[i] * row[j];
    }
    jacobi_eig_lds<9>(A, V);
    int best = 0;
    double bv = INFINITY;
    for (int i = 0; i < 9; ++i) {
      const double ev = A[i * 9 + i];
      if (ev < bv) { bv = ev; best = i; }
    }
#pragma unroll
    for (int q = 0; q < 9; ++q) f[q] = V[q * 9 + best];
  }
  // F = T2^T Fhat T1,  T1 = [[r1,0,-r1 mu0],[0,r1,-r1 mu1],[0,0,1]],  T2^T = [[r2,0,0],[0,r2,0],[-r2 mu2,-r2 mu3,1]]
  double G[3][3];  // Fhat T1
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    G[i][0] = f[3 * i] * r1;
    G[i][1] = f[3 * i + 1] * r1;
    G[i][2] = -r1 * (f[3 * i] * mu[0] + f[3 * i + 1] * mu[1]) + f[3 * i + 2];
  }
  double F[9];
  bool ok = true;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    F[j] = r2 * G[0][j];
    F[3 + j] = r2 * G[1][j];
    F[6 + j] = -r2 * (mu[2] * G[0][j] + mu[3] * G[1][j]) + G[2][j];
  }
#pragma unroll
  for (int q = 0; q < 9; ++q) ok = ok && is_finite(F[q]);
  if (active) {
#pragma unroll
    for (int q = 0; q < 9; ++q) models[(size_t)sc * 9 + q] = ok ? (T)F[q] : T(q % 4 == 0 ? 1 : 0);
    valid[sc] = ok;
  }
}

__device__ __forceinline__ double det3(const double (&m)[9]) {
  return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
}

template <typename T>
__global__ __launch_bounds__(64) void f7_kernel(const T *__restrict__ samples, int Bt, T *__restrict__ models,
                                                uint8_t *__restrict__ valid) {
  const int lane = threadIdx.x;
  const int s = blockIdx.x * 64 + lane;
  const bool active = s < Bt;
  const int sc = active ? s : Bt - 1;
  const T *pts = samples + (size_t)sc * 28;
  double A[7][9];
#pragma unroll
  for (int r = 0; r < 7; ++r)
    epipolar_row_f((double)pts[4 * r], (double)pts[4 * r + 1], (double)pts[4 * r + 2], (double)pts[4 * r + 3], 1.0, A[r]);
  double nb[2][9];
  null_space_qr<7>(A, nb);
  // p(l) = det(l F1 + (1-l) F2) = c0 + c1 l + c2 l^2 + c3 l^3, by interpolation at l = 0, +-1, +-2
  auto p = [&](double l) {
    double m[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) m[q] = l * nb[0][q] + (1.0 - l) * nb[1][q];
    return det3(m);
  };
  const double p0 = p(0.0), p1 = p(1.0), pm1 = p(-1.0), p2 = p(2.0), pm2 = p(-2.0);
  double c[4];
  c[0] = p0;
  c[2] = 0.5 * (p1 + pm1) - p0;
  c[1] = 2.0 * (p1 - pm1) / 3.0 - (p2 - pm2) / 12.0;
  c[3] = (p2 - pm2) / 12.0 - (p1 - pm1) / 6.0;
  double roots[3];
  int nroots;
  real_roots<3, 24, 8>(c, roots, nroots, 1e-14);   // cubic: cheap, and there is no polish afterwards -> run it to full precision
  T *mdl = models + (size_t)sc * 36;
  uint8_t *vld = valid + (size_t)sc * 4;
  int slot = 0;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    if (!(i < nroots)) continue;
    const double l = roots[i];
    double m[9], n2 = 0;
#pragma unroll
    for (int q = 0; q < 9; ++q) {
      m[q] = l * nb[0][q] + (1.0 - l) * nb[1][q];
      n2 += m[q] * m[q];
    }
    const double inv = 1.0 / sqrt(n2);
    const bool good = is_finite(inv) && n2 > 0;
    if (good && active) {
#pragma unroll
      for (int q = 0; q < 9; ++q) mdl[9 * slot + q] = (T)(m[q] * inv);
      vld[slot] = 1;
    }
    slot += good ? 1 : 0;
  }
  if (active)
    for (int q = slot; q < 4; ++q) {
#pragma unroll
      for (int e = 0; e < 9; ++e) mdl[9 * q + e] = T(e % 4 == 0 ? 1 : 0);
      vld[q] = 0;
    }
}

template <typename T>
int f8_launch(const T *samples, const T *weights, int Bt, int n, T *models, uint8_t *valid, hipStream_t st) {
  const size_t smem = (n == 8) ? 0 : sizeof(double) * kF8Ws * 64;
  static bool attr_set[64] = {false};   // per device: one process may drive several GPUs
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 64 || !attr_set[dev]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&f8_kernel<T, false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)(sizeof(double) * kF8Ws * 64));
    if (dev >= 0 && dev < 64) attr_set[dev] = true;
  }
  hipLaunchKernelGGL((f8_kernel<T, false>), dim3((Bt + 63) / 64), dim3(64), smem, st, samples, weights, Bt, n, models, valid);
  return check_launch("f8_kernel");
}

}  // namespace dr

extern "C" {

int dr_solve_f8_f32(const float *samples, const float *weights, int Bt, int n, float *models, uint8_t *valid,
                    void *stream) {
  DR_REQUIRE(samples && models && valid, "null pointer");
  DR_REQUIRE(Bt > 0 && n >= 8, "need Bt > 0 and n >= 8 points per sample");
  return dr::f8_launch<float>(samples, weights, Bt, n, models, valid, (hipStream_t)stream);
}
int dr_solve_f8_f64(const double *samples, const double *weights, int Bt, int n, double *models, uint8_t *valid,
                    void *stream) {
  DR_REQUIRE(samples && models && valid, "null pointer");
  DR_REQUIRE(Bt > 0 && n >= 8, "need Bt > 0 and n >= 8 points per sample");
  return dr::f8_launch<double>(samples, weights, Bt, n, models, valid, (hipStream_t)stream);
}
// K1u + K2 + K3f8 in one launch: P x B minimal samples of eight correspondences drawn by UniformSampler's rule (indices in
// [0, N - 2]: the same index sets dr_uniform_sample draws for this seed), solved by the 8-point kernel; idx [P,B,8] may be NULL
int dr_solve_f8_uniform_f32(const float *matches, uint64_t seed, const uint64_t *seed_dev, int P, int B, int N, int32_t *idx,
                            float *models, uint8_t *valid, void *stream) {
  DR_REQUIRE(matches && models && valid, "null pointer");
  DR_REQUIRE(P > 0 && B > 0 && N > 1 && (long)P * B < (1l << 31), "bad sizes");
  const int Bt = P * B;
  hipLaunchKernelGGL((dr::f8_kernel<float, true>), dim3((Bt + 63) / 64), dim3(64), 0, (hipStream_t)stream, matches,
                     (const float *)nullptr, Bt, 8, models, valid, seed, seed_dev, B, N, idx);
  return dr::check_launch("f8_kernel");
}

int dr_solve_f7_f32(const float *samples, int Bt, float *models, uint8_t *valid, void *stream) {
  DR_REQUIRE(samples && models && valid, "null pointer");
  DR_REQUIRE(Bt > 0, "need Bt > 0");
  hipLaunchKernelGGL((dr::f7_kernel<float>), dim3((Bt + 63) / 64), dim3(64), 0, (hipStream_t)stream, samples, Bt,
                     models, valid);
  return dr::check_launch("f7_kernel");
}
int dr_solve_f7_f64(const double *samples, int Bt, double *models, uint8_t *valid, void *stream) {
  DR_REQUIRE(samples && models && valid, "null pointer");
  DR_REQUIRE(Bt > 0, "need Bt > 0");
  hipLaunchKernelGGL((dr::f7_kernel<double>), dim3((Bt + 63) / 64), dim3(64), 0, (hipStream_t)stream, samples, Bt,
                     models, valid);
  return dr::check_launch("f7_kernel");
}

}  // extern "C"
