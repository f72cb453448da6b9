// K7 -- final refit of the best model (RANSAC.__call__, ransac.py:148-195), batched over pairs with no host round trip:
//   essential:    Nister on ALL N points of the pair as one "sample" (what nister.py:64-65 does when pymagsac is absent,
//                 ransac.py:157-165), in f64;
//   fundamental:  Hartley-normalised LSQ 8-point on the INLIERS of the best mask (ransac.py:150-155,
//                 fundamental_matrix_estimator.py:172-174,177-260).
// One 256-thread block per pair: the four waves accumulate the 9x9 Gram matrix of the (weighted / masked) epipolar rows
// cooperatively (three passes for F: centroid, mean distances, Gram), then wave 0 -- all 64 lanes redundantly, lane 0
// stores -- runs the same device code as the per-sample solvers (Jacobi eigen-decomposition, five-point pipeline).
// The ragged inlier sets never leave the device.
#include <algorithm>
#include <cstdlib>
#include "fivepoint_device.hpp"

namespace dr {

constexpr int kRefT = 256;
constexpr int kRefitPairFinishMinPairs = 16;   // launches of fewer pairs keep the light final stage (see refit_essential_kernel)

__device__ __forceinline__ double block_sum(double v, double *red /* [4] */) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// accumulates sum_n w_n * row_n row_n^T (45 unique entries) over this thread's points into gram[81] in LDS
template <bool kFundamental, typename T>
__device__ __forceinline__ void gram_accumulate(const T *__restrict__ mt, const uint8_t *__restrict__ mk,
                                                const T *__restrict__ wt, int N,
                                                const double (&mu)[4], double r1, double r2, double *gram, double *red) {
  double acc[45];
#pragma unroll
  for (int q = 0; q < 45; ++q) acc[q] = 0;
  for (int n = threadIdx.x; n < N; n += kRefT) {
    if (mk && !mk[n]) continue;
    double row[9];
    if (kFundamental)
      epipolar_row_f(((double)mt[4 * n] - mu[0]) * r1, ((double)mt[4 * n + 1] - mu[1]) * r1,
                     ((double)mt[4 * n + 2] - mu[2]) * r2, ((double)mt[4 * n + 3] - mu[3]) * r2,
                     wt ? (double)wt[n] : 1.0, row);   // weights scale the ROWS (fundamental_matrix_estimator.py:243-244)
    else
      epipolar_row_5pt((double)mt[4 * n], (double)mt[4 * n + 1], (double)mt[4 * n + 2], (double)mt[4 * n + 3], 1.0, row);
    int q = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i)
#pragma unroll
      for (int j = i; j < 9; ++j) acc[q++] += row[i] * row[j];
  }
  // 45 partial sums per thread: reduced inside each wave by DPP (no LDS traffic), one LDS slot per (wave, entry), ONE
  // barrier, then 45 threads add the four wave partials (was: 45 x (wave butterfly over ds_bpermute + two barriers))
  double *part = gram + 96;   // [4][45] behind gram[81] + red[4] (+ padding); see the LDS sizes at the launch sites
#pragma unroll
  for (int q = 0; q < 45; ++q) {
    const double s = wave_sum_lane63(acc[q]);
    if ((threadIdx.x & 63) == 63) part[(threadIdx.x >> 6) * 45 + q] = s;
  }
  __syncthreads();
  if (threadIdx.x < 45) {
    const double s = part[threadIdx.x] + part[45 + threadIdx.x] + part[90 + threadIdx.x] + part[135 + threadIdx.x];
    int i = 0, rem = threadIdx.x;     // entry q of the upper triangle -> (i, j)
    while (rem >= 9 - i) { rem -= 9 - i; ++i; }
    const int j = i + rem;
    gram[i * 9 + j] = s;
    gram[j * 9 + i] = s;
  }
  __syncthreads();
}

// Round 5: the stages after the elimination are the minimal solver's wave-cooperative ones (nister_finish_pair: Sturm isolation,
// refinement and the polish / verification of the candidates dealt out over the wave's lanes) with ONE occupied lane pair -- the
// one-sample form they replace (derivative-chain root search and up to ten polish + verification passes, one after the other,
// every lane redundantly) was 86 of the kernel's 120 us.  The price: the minimal solver's register and LDS footprint (one wave per
// SIMD, 38.9 KB) -- which costs nothing where it matters: a solver wave of the same call takes every register of its SIMD since
// round 3, so a refit wave never shared a SIMD with one; the sampler's and the scoring kernel's light waves still fit beside it.
// kPairFinish is chosen per launch: calls of a few pairs (the drop-in's one pair per call, where the refit hides behind the whole
// chain anyway) keep the light one-sample form -- 256 registers, 4.6 KB of LDS; same-box A/B of the per-pair loop: 0.208-0.211 ms
// per pair with it, 0.214-0.217 with the heavy form -- batched calls take the fast one.
// (register budget of the light form: 256 per lane, waves_per_eu(2, 2), although one wave per pair does the work)
template <typename T, bool kPairFinish>
__global__ __launch_bounds__(kRefT) __attribute__((amdgpu_waves_per_eu(kPairFinish ? 1 : 2, kPairFinish ? 1 : 2))) void refit_essential_kernel(const T *__restrict__ matches,
                                                                const uint8_t *__restrict__ mask, int N,
                                                                T *__restrict__ models, uint8_t *__restrict__ valid) {
  extern __shared__ __align__(16) double lds[];   // [192] five-point workspace, then gram[81] + red[4], wave partials, Jacobi
  double *gram = lds + 192;   // [0,162): the one five-point workspace slot all lanes share (identical values)
  double *red = gram + 81;
  const int p = blockIdx.x;
  const T *mt = matches + (size_t)p * N * 4;
  const uint8_t *mk = mask ? mask + (size_t)p * N : nullptr;
  const double mu[4] = {0, 0, 0, 0};
  gram_accumulate<false, T>(mt, mk, static_cast<const T *>(nullptr), N, mu, 1.0, 1.0, gram, red);
#if defined(DR_REFIT_STOP) && DR_REFIT_STOP == 1
  return;
#endif
  if (threadIdx.x >= 64) return;
  // the rest is one latency-bound wave per pair, usually running next to the scoring kernel of the same call (8 VALU-bound
  // waves per SIMD): raise its issue priority so that it proceeds at its own pace and the scoring waves fill the gaps
  __builtin_amdgcn_s_setprio(3);
  const int lane = threadIdx.x;
  double nb[4][9];
  {
    // 9x9 eigen-decomposition by the whole wave in LDS (was: every lane ran the full cyclic Jacobi redundantly in
    // registers: 324 live doubles, 512 registers + 1.5 KB of scratch per lane and ~120 us of the kernel)
    double *Vl = gram + 96 + 4 * 45, *cs = Vl + 81;
    jacobi_eig9_wave(gram, Vl, cs, lane);
    double ev[4][9];
    smallest_eigvecs9_lds<4>(gram, Vl, ev);
    // nb[3] <-> smallest eigenvalue ... nb[0] <-> fourth smallest (the order of torch.linalg.svd's Vh[-4:])
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 9; ++r) nb[3 - t][r] = ev[t][r];
  }
#if defined(DR_REFIT_STOP) && DR_REFIT_STOP == 2
  if (lane == 0) models[(size_t)p * 90] = (T)nb[0][0];
  return;
#endif
  double e[3][3][4];
  basis_to_entries(nb, e);
  LaneWs w{lds, 1};   // every lane solves the same system: one shared slot, same-address writes of equal values
  double X[6][10];
  const bool ok = constraints_reduce<NisterOrder, 4>(e, w, 1.0, X);
  if constexpr (kPairFinish) {
    // lane pair 0 holds the sample (lane 0 searches |z| <= 1, lane 1 |z| > 1); the other 31 pairs are empty slots of the wave's
    // task queues, which is where the pair's brackets and candidates are worked on side by side.  The block's LDS is reused whole.
    nister_finish_pair<T>(nb, X, ok, lds, lane, 0, lane < 2, models + (size_t)p * 90, valid + (size_t)p * 10, nullptr);
  } else {
    // like the minimal solver, two lanes share the sample: even lanes search |z| <= 1 and fill the slots from 0 upwards,
    // odd lanes |z| > 1 from 9 downwards (all 32 lane pairs do the same work; lanes 0 and 1 store)
    nister_finish<T, true>(nb, X, ok, models + (size_t)p * 90, valid + (size_t)p * 10, lane < 2, lane & 1);
  }
}

template <typename T>
__global__ __launch_bounds__(kRefT) __attribute__((amdgpu_waves_per_eu(1, 1))) void refit_fundamental_kernel(const T *__restrict__ matches,
                                                                  const uint8_t *__restrict__ mask,
                                                                  const T *__restrict__ weights, int N,
                                                                  T *__restrict__ models, uint8_t *__restrict__ valid) {
  extern __shared__ __align__(16) double lds[];
  double *gram = lds + 192;   // [0,162): the one five-point workspace slot all lanes share (identical values)
  double *red = gram + 81;
  const int p = blockIdx.x;
  const T *mt = matches + (size_t)p * N * 4;
  const uint8_t *mk = mask ? mask + (size_t)p * N : nullptr;
  // per-point row weights [P,N] (ransac.py:151-153 hands the estimator `soft_weights[0, inlier_indices]`); the Hartley
  // normalisation below stays unweighted, as in fundamental_matrix_estimator.py:177-228
  const T *wt = weights ? weights + (size_t)p * N : nullptr;
  // pass 1: centroid of the selected points
  double s[4] = {0, 0, 0, 0}, cnt = 0;
  for (int n = threadIdx.x; n < N; n += kRefT) {
    if (mk && !mk[n]) continue;
#pragma unroll
    for (int d = 0; d < 4; ++d) s[d] += (double)mt[4 * n + d];
    cnt += 1.0;
  }
  const double nsel = block_sum(cnt, red);
  double mu[4];
#pragma unroll
  for (int d = 0; d < 4; ++d) mu[d] = block_sum(s[d], red) / nsel;
  // pass 2: mean distances
  double d1 = 0, d2 = 0;
  for (int n = threadIdx.x; n < N; n += kRefT) {
    if (mk && !mk[n]) continue;
    const double a = (double)mt[4 * n] - mu[0], b = (double)mt[4 * n + 1] - mu[1];
    const double c = (double)mt[4 * n + 2] - mu[2], d = (double)mt[4 * n + 3] - mu[3];
    d1 += sqrt(a * a + b * b);
    d2 += sqrt(c * c + d * d);
  }
  const double r1 = M_SQRT2 * nsel / block_sum(d1, red), r2 = M_SQRT2 * nsel / block_sum(d2, red);
  // pass 3: Gram matrix of the normalised rows
  gram_accumulate<true, T>(mt, mk, wt, N, mu, r1, r2, gram, red);
  if (threadIdx.x >= 64) return;
  const int lane = threadIdx.x;
  double f[9];
  {
    double A[9][9];
#pragma unroll
    for (int i = 0; i < 9; ++i)
#pragma unroll
      for (int j = 0; j < 9; ++j) A[i][j] = gram[i * 9 + j];
    smallest_eigvec9_invit(A, f);   // only the last right singular vector is needed (fundamental_matrix_estimator.py:249-254)
  }
  double G[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    G[i][0] = f[3 * i] * r1;
    G[i][1] = f[3 * i + 1] * r1;
    G[i][2] = -r1 * (f[3 * i] * mu[0] + f[3 * i + 1] * mu[1]) + f[3 * i + 2];
  }
  double F[9];
  bool ok = nsel >= 8.0;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    F[j] = r2 * G[0][j];
    F[3 + j] = r2 * G[1][j];
    F[6 + j] = -r2 * (mu[2] * G[0][j] + mu[3] * G[1][j]) + G[2][j];
  }
#pragma unroll
  for (int q = 0; q < 9; ++q) ok = ok && is_finite(F[q]);
  if (lane == 0) {
#pragma unroll
    for (int q = 0; q < 9; ++q) models[(size_t)p * 9 + q] = ok ? (T)F[q] : T(q % 4 == 0 ? 1 : 0);
    valid[p] = ok;
  }
}

// A/B knob (DRANSAC_REFIT_PAIR_MIN): launches of at least this many pairs take the wave-cooperative final stage
static inline int refit_pair_finish_min_pairs() {
  static int v = -1;
  if (v < 0) {
    const char *e = getenv("DRANSAC_REFIT_PAIR_MIN");
    v = e ? atoi(e) : kRefitPairFinishMinPairs;
    if (v < 1) v = 1;
  }
  return v;
}

template <typename T>
int refit_launch(bool fundamental, const T *matches, const uint8_t *mask, const T *weights, int P, int N, T *models,
                 uint8_t *valid, hipStream_t st) {
  // five-point workspace (one slot), gram[81] + red[4] (padded to 96), wave partials [4][45], Jacobi V[81] + (C, S)[9]:
  // 4.6 KB -- with a 162-double slot PER LANE (83 KB) a block left room for only two of the four solver blocks a CU hosts
  const size_t smem = sizeof(double) * (192 + 96 + 4 * 45 + 81 + 18);   // (the Jacobi's per-index rotation table: C[9], S[9])
  static bool attr_e = false, attr_p = false, attr_f = false;
  if (fundamental) {
    if (!attr_f) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&refit_fundamental_kernel<T>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      attr_f = true;
    }
    hipLaunchKernelGGL((refit_fundamental_kernel<T>), dim3(P), dim3(kRefT), smem, st, matches, mask, weights, N, models, valid);
  } else if (P >= refit_pair_finish_min_pairs()) {
    const size_t smem_p = std::max(smem, sizeof(double) * (size_t)kNisterPairDoubles);   // the solver's workspace, overlaid
    if (!attr_p) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&refit_essential_kernel<T, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_p);
      attr_p = true;
    }
    hipLaunchKernelGGL((refit_essential_kernel<T, true>), dim3(P), dim3(kRefT), smem_p, st, matches, mask, N, models, valid);
  } else {
    if (!attr_e) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&refit_essential_kernel<T, false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      attr_e = true;
    }
    hipLaunchKernelGGL((refit_essential_kernel<T, false>), dim3(P), dim3(kRefT), smem, st, matches, mask, N, models, valid);
  }
  return check_launch("refit_kernel");
}

}  // namespace dr

extern "C" {

int dr_refit_essential_f32(const float *matches, const uint8_t *mask, int P, int N, float *models, uint8_t *valid,
                           void *stream) {
  DR_REQUIRE(matches && models && valid, "null pointer");
  DR_REQUIRE(P > 0 && N >= 5, "bad sizes");
  return dr::refit_launch<float>(false, matches, mask, nullptr, P, N, models, valid, (hipStream_t)stream);
}
int dr_refit_essential_f64(const double *matches, const uint8_t *mask, int P, int N, double *models, uint8_t *valid,
                           void *stream) {
  DR_REQUIRE(matches && models && valid, "null pointer");
  DR_REQUIRE(P > 0 && N >= 5, "bad sizes");
  return dr::refit_launch<double>(false, matches, mask, nullptr, P, N, models, valid, (hipStream_t)stream);
}
int dr_refit_fundamental_f32(const float *matches, const uint8_t *mask, const float *weights, int P, int N, float *models,
                             uint8_t *valid, void *stream) {
  DR_REQUIRE(matches && models && valid, "null pointer");
  DR_REQUIRE(P > 0 && N >= 8, "bad sizes");
  return dr::refit_launch<float>(true, matches, mask, weights, P, N, models, valid, (hipStream_t)stream);
}
int dr_refit_fundamental_f64(const double *matches, const uint8_t *mask, const double *weights, int P, int N, double *models,
                             uint8_t *valid, void *stream) {
  DR_REQUIRE(matches && models && valid, "null pointer");
  DR_REQUIRE(P > 0 && N >= 8, "bad sizes");
  return dr::refit_launch<double>(true, matches, mask, weights, P, N, models, valid, (hipStream_t)stream);
}

}  // extern "C"
