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

// waves per SIMD the minimal-sample kernels are compiled for (register budget = 512 / DR_K3_WAVES)
#ifndef DR_K3_WAVES
#define DR_K3_WAVES 1
#endif
// 1: the candidates of a wave are dealt out evenly over its lanes for the polish / verification stage (balanced_finish)
#ifndef DR_K3_BALANCED
#define DR_K3_BALANCED 1
#endif

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
// share its LDS slot), then each owns ONE of the two root searches (|z| <= 1 / |z| > 1).  From there on the work is dealt
// out over the wave instead of staying with its lane: brackets that hold a sign change (real_roots_half_wave: 1083 of the
// 3520 bracket refinements a wave used to run) and candidate solutions (balanced_finish: ~150 per wave, 2.3 per lane,
// where the busiest lane has 6-7).  Same arithmetic per bracket / candidate, so the output is bit-identical to the
// per-lane version (-DDR_K3_BALANCED=0 -DDR_K3_WAVE_ROOTS=0); 105 -> 72 us at 32 pairs x 1024, 386 -> 265 us at 128 pairs.
template <typename T>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(DR_K3_WAVES, DR_K3_WAVES))) void nister5_pair_kernel(
    const T *__restrict__ samples, const T *__restrict__ weights, int Bt, T *__restrict__ models,
    uint8_t *__restrict__ valid, double *__restrict__ models64, int spb) {
  // spb = samples per block: 32 when the grid fills the chip.  Calls with few samples (one pair = 1024 samples = 32 blocks on
  // 1024 SIMDs) run 16 / 8 / 4 samples per block instead: the lane pairs beyond spb hold no sample, queue no bracket and no
  // candidate, so the wave's task rounds (refine, polish, verification) shrink with spb while the per-lane stages cost what they
  // cost -- the block's latency is what a one-pair call waits for (round 4)
  extern __shared__ __align__(16) double lds[];
  const int lane = threadIdx.x;
  const int s = blockIdx.x * spb + (lane >> 1);
  const bool active = (lane >> 1) < spb && s < Bt;
  const int sc = active ? s : Bt - 1;
  LaneWs w{lds + (lane >> 1), 32};
  const T *pts = samples + (size_t)sc * 20;
  const T *wts = weights ? weights + (size_t)sc * 5 : nullptr;
  double nb[4][9];
  DR_STAGE_BEGIN();
  fivepoint_basis_minimal<T>(pts, wts, nb);
  DR_STAGE(0);
  double e[3][3][4];
  basis_to_entries(nb, e);
  double X[6][10];
  const bool ok = constraints_reduce<NisterOrder, 4, DR_K3_BALANCED != 0>(e, w, 1.0, X, lane & 1);
  DR_STAGE(2);
#if DR_K3_BALANCED
  nister_finish_pair<T>(nb, X, ok, lds, lane, (size_t)blockIdx.x * spb, active, models, valid, models64);
#else
  nister_finish<T, true>(nb, X, ok, models + (size_t)sc * 90, valid + (size_t)sc * 10, active, lane & 1,
                         models64 ? models64 + (size_t)sc * 90 : nullptr);
#endif
  DR_STAGE(5);
}

// ---- Stewenius ---------------------------------------------------------------------------------------------
// Action matrix M (10x10): rows 0-5 <- reduced rows 0,1,2,4,5,7; M[6][0] = M[7][1] = M[8][3] = M[9][6] = -1
// (stewenius.py:64-72).  M v = lambda v with v ~ (x^2, xy, y^2, xz, yz, z^2, x, y, z, 1), lambda = -x.
// Eigenvalues: Householder-Hessenberg + La Budde's recurrence give the characteristic polynomial, whose real
// roots come from the same root finder; the eigenvector follows from rows 0-5 of (M - lambda I) v = 0 with
// the structural rows substituted (unknowns y^2, yz, z^2, y, z), solved in the least-squares sense by Householder
// QR.  Everything after the constraint solve is statically indexed and lives in VGPRs; like the Nister kernel, two
// lanes share one sample and each takes one half of the root search.
template <typename T>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(DR_K3_WAVES, DR_K3_WAVES))) void stewenius5_pair_kernel(
    const T *__restrict__ samples, int Bt, T *__restrict__ models, uint8_t *__restrict__ valid, int spb) {
  // spb = samples per block (32, or 16 / 8 / 4 on small grids): see nister5_pair_kernel
  extern __shared__ __align__(16) double lds[];
  const int lane = threadIdx.x;
  const int half = lane & 1;
  const int s = blockIdx.x * spb + (lane >> 1);
  const bool active = (lane >> 1) < spb && s < Bt;
  const int sc = active ? s : Bt - 1;
  LaneWs w{lds + (lane >> 1), 32};
  double nb[4][9];
  fivepoint_basis_minimal<T>(samples + (size_t)sc * 20, nullptr, nb);
  double g[6][10];   // G rows (right block) needed by the action matrix: r in {0,1,2,4,5,7}
  bool ok;
  {
    double e[3][3][4];
    basis_to_entries(nb, e);
    double X[10][10];
    ok = constraints_reduce<GrevlexOrder, 0, DR_K3_BALANCED != 0>(e, w, 2.0, X, half);
    const int src[6] = {0, 1, 2, 4, 5, 7};
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
      for (int c = 0; c < 10; ++c) g[r][c] = X[src[r]][c];
  }
  // characteristic polynomial of the action matrix
  double cs[11];
  {
    double H[10][10];
#pragma unroll
    for (int r = 0; r < 10; ++r)
#pragma unroll
      for (int c = 0; c < 10; ++c) H[r][c] = (r < 6) ? g[r][c] : 0.0;
    H[6][0] = -1.0; H[7][1] = -1.0; H[8][3] = -1.0; H[9][6] = -1.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      double v[10];
      double nrm2 = 0;
#pragma unroll
      for (int i = 0; i < 10; ++i) {
        v[i] = (i > k) ? H[i][k] : 0.0;
        nrm2 += v[i] * v[i];
      }
      const double x0 = v[k + 1];
      const double alpha = -dsign(sqrt(nrm2), x0);
      const double v0 = x0 - alpha;
      const double vtv = v0 * v0 + (nrm2 - x0 * x0);
      const double beta = vtv > 0 ? 2.0 / vtv : 0.0;
      v[k + 1] = v0;
      // H <- (I - beta v v^T) H (I - beta v v^T)
#pragma unroll
      for (int c = 0; c < 10; ++c) {
        double dot = 0;
#pragma unroll
        for (int i = k + 1; i < 10; ++i) dot += v[i] * H[i][c];
        dot *= beta;
#pragma unroll
        for (int i = k + 1; i < 10; ++i) H[i][c] -= dot * v[i];
      }
#pragma unroll
      for (int r = 0; r < 10; ++r) {
        double dot = 0;
#pragma unroll
        for (int i = k + 1; i < 10; ++i) dot += H[r][i] * v[i];
        dot *= beta;
#pragma unroll
        for (int i = k + 1; i < 10; ++i) H[r][i] -= dot * v[i];
      }
    }
    // La Budde: p_0 = 1, p_i(l) = (l - h_ii) p_{i-1} - sum_{m=1}^{i-1} h_{i-m,i} (prod_{j=i-m+1}^{i} h_{j,j-1}) p_{i-m-1}
    // (1-based indices), coefficients ascending: P[i][0..i]
    double P[11][11];
#pragma unroll
    for (int i = 0; i < 11; ++i)
#pragma unroll
      for (int t = 0; t < 11; ++t) P[i][t] = 0.0;
    P[0][0] = 1.0;
#pragma unroll
    for (int i = 1; i <= 10; ++i) {
      const double hii = H[i - 1][i - 1];
#pragma unroll
      for (int t = 0; t <= i; ++t) {
        const double up = (t > 0) ? P[i - 1][t - 1] : 0.0;
        const double same = (t <= i - 1) ? P[i - 1][t] : 0.0;
        P[i][t] = up - hii * same;
      }
      double prod = 1.0;
#pragma unroll
      for (int m = 1; m <= i - 1; ++m) {
        prod *= H[i - m][i - m - 1];
        const double coef = H[i - m - 1][i - 1] * prod;
#pragma unroll
        for (int t = 0; t <= i - m - 1; ++t) P[i][t] -= coef * P[i - m - 1][t];
      }
    }
#pragma unroll
    for (int t = 0; t <= 10; ++t) cs[t] = P[10][t];
  }
  double roots[10];
  int nroots;
  if (!active) {   // no sample in this lane pair: 1 + z^10, no real root in either half of the search, no bracket in the queues
#pragma unroll
    for (int t = 0; t <= 10; ++t) cs[t] = (t == 0 || t == 10) ? 1.0 : 0.0;
  }
#if DR_K3_WAVE_ROOTS
#if DR_K3_STURM
  real_roots_half_sturm<10>(cs, half != 0, roots, nroots, lds, lane);   // the right block's LDS is free again
#else
  real_roots_half_wave<10>(cs, half != 0, roots, nroots, lds, lane);   // the right block's LDS is free again
#endif
#else
  real_roots_half<10>(cs, half != 0, roots, nroots);
#endif
  if (!ok) nroots = 0;

#if DR_K3_BALANCED
  // eigenvector of every root per lane (cheap), then polish / verification dealt out over the wave (balanced_finish)
  if (!active) nroots = 0;
  const FinishQueue fq(lds);   // the root-search workspace is dead: basis, candidate queue and vectors take its place
  park_basis(fq, nb, lane);
  double xs[10], ys[10], zs[10];
  unsigned cand = 0;
#else
  T *mdl = models + (size_t)sc * 90;
  uint8_t *vld = valid + (size_t)sc * 10;
  int slot = 0;
#endif
#pragma unroll
  for (int i = 0; i < 10; ++i) {
#if DR_K3_BALANCED
    xs[i] = 0; ys[i] = 0; zs[i] = 0;
#endif
    if (!__any(i < nroots)) continue;
    const bool has = i < nroots;
    const double lam = roots[i];
    const double l2 = lam * lam;
    // unknown order u = (v2, v4, v5, v7, v8); column 5 = right-hand side (minus the constant term)
    double K[6][6];
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
      K[r][0] = c2; K[r][1] = c4; K[r][2] = c5; K[r][3] = c7; K[r][4] = c8; K[r][5] = -k0;
    }
    // least squares of the consistent 6x5 system by Householder QR (no pivoting => static indexing)
    bool solvable = true;
#pragma unroll
    for (int c = 0; c < 5; ++c) {
      double nrm2 = 0;
#pragma unroll
      for (int r = c; r < 6; ++r) nrm2 += K[r][c] * K[r][c];
      const double nrm = sqrt(nrm2);
      const double alpha = -dsign(nrm, K[c][c]);
      const double v0 = K[c][c] - alpha;
      const double vtv = v0 * v0 + (nrm2 - K[c][c] * K[c][c]);
      const double beta = vtv > 0 ? 2.0 / vtv : 0.0;
      if (!(nrm > 0)) solvable = false;
      double v[6];
#pragma unroll
      for (int r = 0; r < 6; ++r) v[r] = (r > c) ? K[r][c] : 0.0;
      v[c] = v0;
#pragma unroll
      for (int cc = c + 1; cc < 6; ++cc) {
        double dot = 0;
#pragma unroll
        for (int r = c; r < 6; ++r) dot += v[r] * K[r][cc];
        dot *= beta;
#pragma unroll
        for (int r = c; r < 6; ++r) K[r][cc] -= dot * v[r];
      }
      K[c][c] = alpha;
    }
    double u[5];
#pragma unroll
    for (int c = 4; c >= 0; --c) {
      double acc = K[c][5];
#pragma unroll
      for (int cc = 4; cc > c; --cc) acc -= K[c][cc] * u[cc];
      u[c] = acc / K[c][c];
    }
    const double x = -lam, y = u[3], z = u[4];
#if DR_K3_BALANCED
    xs[i] = x; ys[i] = y; zs[i] = z;
    if (has && solvable && is_finite(y) && is_finite(z)) cand |= 1u << i;
  }
  balanced_finish<T>(fq, lane, nroots, xs, ys, zs, cand, (size_t)blockIdx.x * spb, active, models, valid, nullptr);
#else
    const int dst_slot = half ? 9 - slot : slot;
    const bool good = finish_solution<T>(nb, x, y, z, has && solvable && is_finite(y) && is_finite(z) && slot < 10,
                                         mdl + 9 * dst_slot, active);
    if (good && active) vld[dst_slot] = 1;
    slot += good ? 1 : 0;
  }
  const int other = __shfl_xor(slot, 1, 64);
  const int lo = half ? other : slot, hi = half ? slot : other;
  if (active && half == 0) {
    for (int q = min(lo, 10 - hi); q < 10 - hi; ++q) {
      write_identity<T>(mdl + 9 * q);
      vld[q] = 0;
    }
  }
#endif
}

static inline bool aligned_out(const void *models, const void *valid) {
  return !DR_K3_STAGE_OUT || ((reinterpret_cast<uintptr_t>(models) & 15u) == 0 && (reinterpret_cast<uintptr_t>(valid) & 3u) == 0);
}

// samples per 64-lane block of the two-lanes-per-sample kernels: 32, or fewer (down to 4) while the grid stays within one block
// per SIMD -- a one-pair call (1024 samples) is 256 blocks of 4 samples instead of 32 blocks of 32 on 1024 SIMDs
static inline int samples_per_block(int Bt) {
  int spb = 32;
#if DR_K3_BALANCED
  static int simds = 0;
  if (!simds) {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    simds = 4 * (cus > 0 ? cus : 256);
  }
  while (spb > 4 && (long)(Bt + spb / 2 - 1) / (spb / 2) <= simds) spb /= 2;
#endif
  return spb;
}

template <typename T>
int nister_launch(const T *samples, const T *weights, int Bt, int n, T *models, uint8_t *valid, hipStream_t st,
                  double *models64 = nullptr) {
  static bool attr_set[64] = {false};   // per device: one process may drive several GPUs
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 64 || !attr_set[dev]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&nister5_kernel<T>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(double) * kFiveWs * 64));
    if (dev >= 0 && dev < 64) attr_set[dev] = true;
  }
  if (n == 5) {
    // minimal samples: two lanes per sample; LDS per 32-sample block: 100 doubles per sample for the constraint solve,
    // later basis + B(z) / candidate queue + root-search workspace (36.5 KiB => four blocks per CU, one per SIMD)
    const size_t smem = sizeof(double) * (DR_K3_BALANCED ? kNisterPairDoubles : 100 * 32);
    const int spb = samples_per_block(Bt);
    hipLaunchKernelGGL((nister5_pair_kernel<T>), dim3((Bt + spb - 1) / spb), dim3(64), smem, st, samples, weights, Bt, models,
                       valid, models64, spb);
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
  // the right 10x10 block of 32 samples, later the root-search workspace, then the candidate queue
  constexpr int kDoubles = (DR_K3_BALANCED && FinishQueue::kDoubles > 100 * 32) ? FinishQueue::kDoubles : 100 * 32;
  static_assert(!DR_K3_WAVE_ROOTS || (RootWs<10>::kDoubles <= kDoubles && SturmWs<10>::kDoubles <= kDoubles), "root-search workspace");
  const size_t smem = sizeof(double) * kDoubles;
  const int spb = samples_per_block(Bt);
  hipLaunchKernelGGL((stewenius5_pair_kernel<T>), dim3((Bt + spb - 1) / spb), dim3(64), smem, st, samples, Bt, models, valid, spb);
  return check_launch("stewenius5_pair_kernel");
}

// ---- test hook: the real-root search of the two-lanes-per-sample kernels on given degree-10 polynomials -------------------
// One lane pair per polynomial (even lane: |z| <= 1, odd lane: |z| > 1 through the reversed polynomial), exactly as the solver
// kernels call it.  method 0 = derivative chain (real_roots_half_wave), 1 = Sturm isolation (real_roots_half_sturm).
template <int kMethod>
__global__ __launch_bounds__(64) void debug_roots10_kernel(const double *__restrict__ coef, int n, double *__restrict__ roots,
                                                           int32_t *__restrict__ counts) {
  extern __shared__ __align__(16) double lds[];
  const int lane = threadIdx.x, half = lane & 1;
  const int s = blockIdx.x * 32 + (lane >> 1);
  const int sc = s < n ? s : n - 1;
  double cs[11];
#pragma unroll
  for (int i = 0; i <= 10; ++i) cs[i] = coef[(size_t)sc * 11 + i];
  double r[10];
  int nr = 0;
  if (kMethod == 1) real_roots_half_sturm<10>(cs, half != 0, r, nr, lds, lane);
  else real_roots_half_wave<10>(cs, half != 0, r, nr, lds, lane);
  if (s < n) {
    counts[(size_t)s * 2 + half] = nr;
#pragma unroll
    for (int i = 0; i < 10; ++i) roots[((size_t)s * 2 + half) * 10 + i] = r[i];
  }
}

}  // namespace dr

DR_DEFINE_STAGE_READER(dr_debug_stage_read_fivepoint)

extern "C" {

int dr_solve_nister5_f32(const float *samples, const float *weights, int Bt, int n, float *models, uint8_t *valid,
                         void *stream) {
  DR_REQUIRE(samples && models && valid, "null pointer");
  DR_REQUIRE(dr::aligned_out(models, valid), "models must be 16-byte aligned and valid 4-byte aligned (whole-line output stores)");
  DR_REQUIRE(Bt > 0 && n >= 5, "need Bt > 0 and n >= 5 points per sample");
  return dr::nister_launch<float>(samples, weights, Bt, n, models, valid, (hipStream_t)stream);
}
int dr_solve_nister5_f64(const double *samples, const double *weights, int Bt, int n, double *models,
                         uint8_t *valid, void *stream) {
  DR_REQUIRE(samples && models && valid, "null pointer");
  DR_REQUIRE(Bt > 0 && n >= 5, "need Bt > 0 and n >= 5 points per sample");
  return dr::nister_launch<double>(samples, weights, Bt, n, models, valid, (hipStream_t)stream);
}
int dr_solve_nister5_f32_hp(const float *samples, const float *weights, int Bt, float *models, double *models_f64,
                            uint8_t *valid, void *stream) {
  DR_REQUIRE(samples && models && models_f64 && valid, "null pointer");
  DR_REQUIRE(dr::aligned_out(models, valid), "models must be 16-byte aligned and valid 4-byte aligned (whole-line output stores)");
  DR_REQUIRE(Bt > 0, "need Bt > 0");
  return dr::nister_launch<float>(samples, weights, Bt, 5, models, valid, (hipStream_t)stream, models_f64);
}
int dr_debug_real_roots10(const double *coef, int n, int method, double *roots, int32_t *counts, void *stream) {
  DR_REQUIRE(coef && roots && counts, "null pointer");
  DR_REQUIRE(n > 0 && (method == 0 || method == 1), "need n > 0 and method 0 (derivative chain) or 1 (Sturm)");
  constexpr int kD = dr::RootWs<10>::kDoubles > dr::SturmWs<10>::kDoubles ? dr::RootWs<10>::kDoubles : dr::SturmWs<10>::kDoubles;
  const size_t smem = sizeof(double) * kD;
  if (method == 1)
    hipLaunchKernelGGL((dr::debug_roots10_kernel<1>), dim3((n + 31) / 32), dim3(64), smem, (hipStream_t)stream, coef, n, roots, counts);
  else
    hipLaunchKernelGGL((dr::debug_roots10_kernel<0>), dim3((n + 31) / 32), dim3(64), smem, (hipStream_t)stream, coef, n, roots, counts);
  return dr::check_launch("debug_roots10_kernel");
}

int dr_solve_stewenius5_f32(const float *samples, int Bt, float *models, uint8_t *valid, void *stream) {
  DR_REQUIRE(samples && models && valid, "null pointer");
  DR_REQUIRE(dr::aligned_out(models, valid), "models must be 16-byte aligned and valid 4-byte aligned (whole-line output stores)");
  DR_REQUIRE(Bt > 0, "need Bt > 0");
  return dr::stewenius_launch<float>(samples, Bt, models, valid, (hipStream_t)stream);
}
int dr_solve_stewenius5_f64(const double *samples, int Bt, double *models, uint8_t *valid, void *stream) {
  DR_REQUIRE(samples && models && valid, "null pointer");
  DR_REQUIRE(Bt > 0, "need Bt > 0");
  return dr::stewenius_launch<double>(samples, Bt, models, valid, (hipStream_t)stream);
}

}  // extern "C"
