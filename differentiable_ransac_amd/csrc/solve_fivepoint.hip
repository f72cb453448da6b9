// K3n / K3s -- five-point essential-matrix solvers.
//   Nister:    EssentialMatrixEstimatorNister.estimate_minimal_model, nister.py:69-408
//   Stewenius: EssentialMatrixEstimator.estimate_minimal_model,       stewenius.py:20-80
//
// f64 arithmetic, one lane (or a pair of lanes: nister5_pair_kernel) per minimal sample.  The left half of
// the 10x20 constraint system is factored in VGPRs (Householder QR), the right half waits in LDS
// (element-major => bank-conflict free); null-space basis, B(z), the degree-10 polynomial and the roots
// stay in VGPRs.  Where the reference runs a per-sample Python loop with LAPACK eigvals / inverse / qr,
// this is one launch: Householder null space -> constraints -> QR solve -> det B(z) -> real roots by
// derivative-interlaced bisection/Newton on [-1,1] (and on the reversed polynomial) -> back-substitution
// -> Gauss-Newton polish on the defining constraints -> verification.
#include "fivepoint_device.hpp"

namespace dr {

template <typename T>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) void nister5_kernel(const T *__restrict__ samples, const T *__restrict__ weights,
                                                     int Bt, int n, T *__restrict__ models,
                                                     uint8_t *__restrict__ valid) {
  extern __shared__ __align__(16) double lds[];
  const int lane = threadIdx.x;
  const int s = blockIdx.x * 64 + lane;
  const bool active = s < Bt;
  const int sc = active ? s : Bt - 1;
  LaneWs w{lds + lane};
  const T *pts = samples + (size_t)sc * n * 4;
  const T *wts = weights ? weights + (size_t)sc * n : nullptr;
  double nb[4][9];
  DR_STAGE_BEGIN();
  if (n == 5) fivepoint_basis_minimal<T>(pts, wts, nb);
  else fivepoint_basis_nonminimal<T>(pts, wts, n, w, nb);
  DR_STAGE(0);
  double e[3][3][4];
  basis_to_entries(nb, e);
  double X[6][10];   // reduced rows e..j = rows 4..9 of A^-1 B
  const bool ok = constraints_reduce<NisterOrder, 4>(e, w, 1.0, X);
  DR_STAGE(2);
  nister_finish<T, false>(nb, X, ok, models + (size_t)sc * 90, valid + (size_t)sc * 10, active);
  DR_STAGE(5);
}

// Minimal samples, two lanes per sample (32 samples per 64-lane block).  At the benchmark size there are fewer samples
// than SIMD lanes on the chip (32 768 vs 65 536), so the redundancy is free: both lanes build the same system (they even
// share its LDS slot), then each runs ONE of the two root searches and finishes its own roots -- the longest serial
// stage is cut in half.
template <typename T>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) void nister5_pair_kernel(
    const T *__restrict__ samples, const T *__restrict__ weights, int Bt, T *__restrict__ models,
    uint8_t *__restrict__ valid) {
  extern __shared__ __align__(16) double lds[];
  const int lane = threadIdx.x;
  const int half = lane & 1;
  const int s = blockIdx.x * 32 + (lane >> 1);
  const bool active = s < Bt;
  const int sc = active ? s : Bt - 1;
  LaneWs w{lds + (lane >> 1), 32};
  const T *pts = samples + (size_t)sc * 20;
  const T *wts = weights ? weights + (size_t)sc * 5 : nullptr;
  double nb[4][9];
  fivepoint_basis_minimal<T>(pts, wts, nb);
  double e[3][3][4];
  basis_to_entries(nb, e);
  double X[6][10];
  const bool ok = constraints_reduce<NisterOrder, 4>(e, w, 1.0, X);
  nister_finish<T, true>(nb, X, ok, models + (size_t)sc * 90, valid + (size_t)sc * 10, active, half);
}

// ---- Stewenius ---------------------------------------------------------------------------------------------
// Action matrix M (10x10): rows 0-5 <- reduced rows 0,1,2,4,5,7; M[6][0] = M[7][1] = M[8][3] = M[9][6] = -1
// (stewenius.py:64-72).  M v = lambda v with v ~ (x^2, xy, y^2, xz, yz, z^2, x, y, z, 1), lambda = -x.
// Eigenvalues: Householder-Hessenberg + La Budde's recurrence give the characteristic polynomial, whose real
// roots come from the same root finder; the eigenvector follows from rows 0-5 of (M - lambda I) v = 0 with
// the structural rows substituted (unknowns y^2, yz, z^2, y, z).
template <typename T>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) void stewenius5_kernel(const T *__restrict__ samples, int Bt,
                                                        T *__restrict__ models, uint8_t *__restrict__ valid) {
  extern __shared__ __align__(16) double lds[];
  const int lane = threadIdx.x;
  const int s = blockIdx.x * 64 + lane;
  const bool active = s < Bt;
  const int sc = active ? s : Bt - 1;
  LaneWs w{lds + lane};
  double nb[4][9];
  fivepoint_basis_minimal<T>(samples + (size_t)sc * 20, nullptr, nb);
  double e[3][3][4];
  basis_to_entries(nb, e);
  double g[6][10];   // G rows (right block) needed by the action matrix: r in {0,1,2,4,5,7}
  bool ok;
  {
    double X[10][10];
    ok = constraints_reduce<GrevlexOrder, 0>(e, w, 2.0, X);
    const int src[6] = {0, 1, 2, 4, 5, 7};
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
      for (int c = 0; c < 10; ++c) g[r][c] = X[src[r]][c];
  }
  // H = action matrix in LDS (elements 0..99), reduce to upper Hessenberg by Householder similarity
  LaneWs H{w.base};
#pragma unroll
  for (int r = 0; r < 6; ++r)
#pragma unroll
    for (int c = 0; c < 10; ++c) H[r * 10 + c] = g[r][c];
  for (int r = 6; r < 10; ++r)
    for (int c = 0; c < 10; ++c) H[r * 10 + c] = 0.0;
  H[6 * 10 + 0] = -1.0; H[7 * 10 + 1] = -1.0; H[8 * 10 + 3] = -1.0; H[9 * 10 + 6] = -1.0;
  for (int k = 0; k < 8; ++k) {
    double v[10];
    double nrm2 = 0;
#pragma unroll
    for (int i = 0; i < 10; ++i) {
      v[i] = (i > k) ? H[i * 10 + k] : 0.0;
      nrm2 += v[i] * v[i];
    }
    double x0 = 0;
#pragma unroll
    for (int i = 0; i < 10; ++i) if (i == k + 1) x0 = v[i];
    const double alpha = -dsign(sqrt(nrm2), x0);
    const double v0 = x0 - alpha;
    const double vtv = v0 * v0 + (nrm2 - x0 * x0);
    const double beta = vtv > 0 ? 2.0 / vtv : 0.0;
#pragma unroll
    for (int i = 0; i < 10; ++i) if (i == k + 1) v[i] = v0;
    // H <- (I - beta v v^T) H (I - beta v v^T)
    for (int c = 0; c < 10; ++c) {
      double dot = 0;
#pragma unroll
      for (int i = 0; i < 10; ++i) dot += v[i] * H[i * 10 + c];
      dot *= beta;
#pragma unroll
      for (int i = 0; i < 10; ++i) if (i > k) H[i * 10 + c] -= dot * v[i];
    }
    for (int r = 0; r < 10; ++r) {
      double dot = 0;
#pragma unroll
      for (int i = 0; i < 10; ++i) dot += H[r * 10 + i] * v[i];
      dot *= beta;
#pragma unroll
      for (int i = 0; i < 10; ++i) if (i > k) H[r * 10 + i] -= dot * v[i];
    }
  }
  // La Budde: p_0 = 1, p_i(l) = (l - h_ii) p_{i-1} - sum_{m=1}^{i-1} h_{i-m,i} (prod_{j=i-m+1}^{i} h_{j,j-1}) p_{i-m-1}
  // (1-based).  Coefficients ascending, P[i][0..i] stored in LDS elements 100 + i*11 ...
  LaneWs P{w.base + 100 * 64};
  for (int e2 = 0; e2 < 110; ++e2) P[e2] = 0.0;
  P[0] = 1.0;  // p_0
  for (int i = 1; i <= 9; ++i) {
    // p_i for i = 1..9 kept in LDS; p_10 assembled in registers below
    const double hii = H[(i - 1) * 10 + (i - 1)];
    for (int t = 0; t <= i; ++t) {
      const double up = (t > 0) ? P[(i - 1) * 11 + t - 1] : 0.0;
      const double same = (t <= i - 1) ? P[(i - 1) * 11 + t] : 0.0;
      P[i * 11 + t] = up - hii * same;
    }
    double prod = 1.0;
    for (int m = 1; m <= i - 1; ++m) {
      prod *= H[(i - m) * 10 + (i - m - 1)];  // h_{i-m+1, i-m} (1-based) = H[i-m][i-m-1] (0-based)
      const double coef = H[(i - m - 1) * 10 + (i - 1)] * prod;  // h_{i-m, i}
      for (int t = 0; t <= i - m - 1; ++t) P[i * 11 + t] -= coef * P[(i - m - 1) * 11 + t];
    }
  }
  double cs[11];
  {
    const int i = 10;
    const double hii = H[9 * 10 + 9];
#pragma unroll
    for (int t = 0; t <= 10; ++t) {
      const double up = (t > 0) ? P[9 * 11 + t - 1] : 0.0;
      const double same = (t <= 9) ? P[9 * 11 + t] : 0.0;
      cs[t] = up - hii * same;
    }
    double prod = 1.0;
    for (int m = 1; m <= i - 1; ++m) {
      prod *= H[(i - m) * 10 + (i - m - 1)];
      const double coef = H[(i - m - 1) * 10 + (i - 1)] * prod;
#pragma unroll
      for (int t = 0; t <= 10; ++t)
        if (t <= i - m - 1) cs[t] -= coef * P[(i - m - 1) * 11 + t];
    }
  }
  double roots[10];
  int nroots;
  real_roots<10>(cs, roots, nroots);
  if (!ok) nroots = 0;

  T *mdl = models + (size_t)sc * 90;
  uint8_t *vld = valid + (size_t)sc * 10;
  int slot = 0;
  LaneWs K{w.base};  // 6 x 6 augmented system, reuses the Hessenberg area
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    if (!__any(i < nroots)) continue;
    const bool has = i < nroots;
    const double lam = roots[i];
    const double l2 = lam * lam;
    // unknown order u = (v2, v4, v5, v7, v8) ; column 5 = right-hand side (minus the constant term)
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      double c2 = g[r][2], c4 = g[r][4], c5 = g[r][5];
      double c7 = g[r][7] - lam * g[r][1];
      double c8 = g[r][8] - lam * g[r][3];
      double k0 = g[r][9] - lam * g[r][6] + l2 * g[r][0];
      if (r == 0) k0 -= lam * l2;   // -lam * v0,  v0 = lam^2
      if (r == 1) c7 += l2;          // -lam * v1,  v1 = -lam v7
      if (r == 2) c2 -= lam;
      if (r == 3) c8 += l2;          // -lam * v3,  v3 = -lam v8
      if (r == 4) c4 -= lam;
      if (r == 5) c5 -= lam;
      K[r * 6 + 0] = c2; K[r * 6 + 1] = c4; K[r * 6 + 2] = c5; K[r * 6 + 3] = c7; K[r * 6 + 4] = c8; K[r * 6 + 5] = -k0;
    }
    // Gaussian elimination with partial pivoting over the 6 rows, 5 unknowns
    bool solvable = true;
    for (int col = 0; col < 5; ++col) {
      int piv = col;
      double best = fabs(K[col * 6 + col]);
      for (int r = col + 1; r < 6; ++r) {
        const double vv = fabs(K[r * 6 + col]);
        if (vv > best) { best = vv; piv = r; }
      }
      if (!(best > 0)) solvable = false;
      for (int c = 0; c < 6; ++c) {
        const double a = K[piv * 6 + c], b = K[col * 6 + c];
        K[piv * 6 + c] = b;
        K[col * 6 + c] = a;
      }
      const double inv = best > 0 ? 1.0 / K[col * 6 + col] : 0.0;
      for (int r = col + 1; r < 6; ++r) {
        const double f = K[r * 6 + col] * inv;
        for (int c = col; c < 6; ++c) K[r * 6 + c] -= f * K[col * 6 + c];
      }
    }
    double u[5];
#pragma unroll
    for (int col = 4; col >= 0; --col) {
      double acc = K[col * 6 + 5];
#pragma unroll
      for (int c = 4; c > col; --c) acc -= K[col * 6 + c] * u[c];
      u[col] = acc / K[col * 6 + col];
    }
    const double x = -lam, y = u[3], z = u[4];
    const bool good = finish_solution<T>(nb, x, y, z, has && solvable && is_finite(y) && is_finite(z), mdl + 9 * slot,
                                         active);
    if (good && active) vld[slot] = 1;
    slot += good ? 1 : 0;
  }
  if (active) {
    for (int q = slot; q < 10; ++q) {
      write_identity<T>(mdl + 9 * q);
      vld[q] = 0;
    }
  }
}

template <typename T>
int nister_launch(const T *samples, const T *weights, int Bt, int n, T *models, uint8_t *valid, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&nister5_kernel<T>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(double) * kFiveWs * 64));
    attr_set = true;
  }
  if (n == 5) {
    // minimal samples: two lanes per sample, 100 doubles of LDS per SAMPLE (25 KiB per block => six blocks per CU)
    const size_t smem = sizeof(double) * 100 * 32;
    hipLaunchKernelGGL((nister5_pair_kernel<T>), dim3((Bt + 31) / 32), dim3(64), smem, st, samples, weights, Bt, models,
                       valid);
    return check_launch("nister5_pair_kernel");
  }
  // n > 5 fallback (refit): one lane per sample, A^T A + eigenvectors in LDS (162 doubles)
  const size_t smem = sizeof(double) * kFiveWs * 64;
  hipLaunchKernelGGL((nister5_kernel<T>), dim3((Bt + 63) / 64), dim3(64), smem, st, samples, weights, Bt, n, models,
                     valid);
  return check_launch("nister5_kernel");
}

template <typename T>
int stewenius_launch(const T *samples, int Bt, T *models, uint8_t *valid, hipStream_t st) {
  const size_t smem = sizeof(double) * kStewWs * 64;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&stewenius5_kernel<T>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr_set = true;
  }
  hipLaunchKernelGGL((stewenius5_kernel<T>), dim3((Bt + 63) / 64), dim3(64), smem, st, samples, Bt, models, valid);
  return check_launch("stewenius5_kernel");
}

}  // namespace dr

DR_DEFINE_STAGE_READER(dr_debug_stage_read_fivepoint)

extern "C" {

int dr_solve_nister5_f32(const float *samples, const float *weights, int Bt, int n, float *models, uint8_t *valid,
                         void *stream) {
  DR_REQUIRE(samples && models && valid, "null pointer");
  DR_REQUIRE(Bt > 0 && n >= 5, "need Bt > 0 and n >= 5 points per sample");
  return dr::nister_launch<float>(samples, weights, Bt, n, models, valid, (hipStream_t)stream);
}
int dr_solve_nister5_f64(const double *samples, const double *weights, int Bt, int n, double *models,
                         uint8_t *valid, void *stream) {
  DR_REQUIRE(samples && models && valid, "null pointer");
  DR_REQUIRE(Bt > 0 && n >= 5, "need Bt > 0 and n >= 5 points per sample");
  return dr::nister_launch<double>(samples, weights, Bt, n, models, valid, (hipStream_t)stream);
}
int dr_solve_stewenius5_f32(const float *samples, int Bt, float *models, uint8_t *valid, void *stream) {
  DR_REQUIRE(samples && models && valid, "null pointer");
  DR_REQUIRE(Bt > 0, "need Bt > 0");
  return dr::stewenius_launch<float>(samples, Bt, models, valid, (hipStream_t)stream);
}
int dr_solve_stewenius5_f64(const double *samples, int Bt, double *models, uint8_t *valid, void *stream) {
  DR_REQUIRE(samples && models && valid, "null pointer");
  DR_REQUIRE(Bt > 0, "need Bt > 0");
  return dr::stewenius_launch<double>(samples, Bt, models, valid, (hipStream_t)stream);
}

}  // extern "C"
