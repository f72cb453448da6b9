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

// 1: the two-phase Nister kernel hands over B(z) only and computes the null-space basis again in the back pass (A/B, round 5)
#ifndef DR_K3_FB_REBASIS
#define DR_K3_FB_REBASIS 0
#endif
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
    uint8_t *__restrict__ valid, double *__restrict__ models64, int spb, PairGate gate, int per_pair) {
  // gate (rounds > 1 of a multi-round call): a block all of whose samples belong to terminated pairs returns at once
  if (gate.closed_range((blockIdx.x * spb) / per_pair, (min(blockIdx.x * spb + spb, Bt) - 1) / per_pair)) return;
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
// Eigenvalues: Householder-Hessenberg + La Budde's recurrence give the characteristic polynomial (stewenius_charpoly), whose real
// roots come from the same root finder; the eigenvector follows from rows 0-5 of (M - lambda I) v = 0 with
// the structural rows substituted (stewenius_xyz_of_roots).  Everything after the constraint solve is statically indexed and
// lives in VGPRs; like the Nister kernel, two lanes share one sample and each takes one half of the root search.
// (The per-lane final stage of round 1, -DDR_K3_BALANCED=0, is gone from this kernel in round 5: the balanced one is the product.)
template <typename T>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(DR_K3_WAVES, DR_K3_WAVES))) void stewenius5_pair_kernel(
    const T *__restrict__ samples, int Bt, T *__restrict__ models, uint8_t *__restrict__ valid, int spb, PairGate gate, int per_pair) {
  if (gate.closed_range((blockIdx.x * spb) / per_pair, (min(blockIdx.x * spb + spb, Bt) - 1) / per_pair)) return;
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
    ok = constraints_reduce<GrevlexOrder, 0, true>(e, w, 2.0, X, half);
    const int src[6] = {0, 1, 2, 4, 5, 7};
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
      for (int c = 0; c < 10; ++c) g[r][c] = X[src[r]][c];
  }
  double cs[11];
  stewenius_charpoly(g, cs);
  double roots[10];
  int nroots;
  if (!active) {   // no sample in this lane pair: 1 + z^10, no real root in either half of the search, no bracket in the queues
#pragma unroll
    for (int t = 0; t <= 10; ++t) cs[t] = (t == 0 || t == 10) ? 1.0 : 0.0;
  }
#if DR_K3_STURM
  real_roots_half_sturm<10>(cs, half != 0, roots, nroots, lds, lane);   // the right block's LDS is free again
#else
  real_roots_half_wave<10>(cs, half != 0, roots, nroots, lds, lane);
#endif
  if (!ok || !active) nroots = 0;
  // eigenvector of every root per lane (cheap), then polish / verification dealt out over the wave (balanced_finish)
  const FinishQueue fq(lds);   // the root-search workspace is dead: basis, candidate queue and vectors take its place
  park_basis(fq, nb, lane);
  double xs[10], ys[10], zs[10];
  unsigned cand;
  stewenius_xyz_of_roots(g, roots, nroots, xs, ys, zs, cand);
  balanced_finish<T>(fq, lane, nroots, xs, ys, zs, cand, (size_t)blockIdx.x * spb, active, models, valid, nullptr);
}

// ---- two-phase kernels (round 5) -------------------------------------------------------------------------------------------
// 64 samples per 64-lane block.  Front: one lane per sample (what the lane pairs above compute twice: null space, constraints, QR,
// reduced rows, det B(z) / the characteristic polynomial); the right block's rows 0-6 wait in LDS (35 KiB), rows 7-9 in
// accumulation registers.  Hand-over: what a sample passes on (Nister: basis 36 + B(z) 39 doubles; Stewenius: basis 36 + reduced
// rows 60) does not fit in the 40 KiB of LDS a block may hold for both halves at once, and a round trip through global memory
// stalls a wave that is alone on its SIMD (first version: 214 instead of 171 us) -- every lane parks its own sample's values in its
// ACCUMULATION registers (AccDouble: one move per dword and direction) and the lanes of half p write theirs into the LDS image
// when pass p begins; the polynomial travels by ds_bpermute.  Back: the block's two 32-sample halves one after the other, two lanes
// per sample, exactly the stages of the pair kernels.  Used when the grid is at least two rounds of pair-kernel blocks
// (fivepoint_two_phase); small grids keep the pair kernels (latency).
__device__ __forceinline__ double lane_read(double v, int src) { return __shfl(v, src, 64); }

template <typename T>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) void nister5_fb_kernel(
    const T *__restrict__ samples, const T *__restrict__ weights, int Bt, T *__restrict__ models, uint8_t *__restrict__ valid,
    double *__restrict__ models64, PairGate gate, int per_pair) {
  if (gate.closed_range((blockIdx.x * 64) / per_pair, (min(blockIdx.x * 64 + 64, Bt) - 1) / per_pair)) return;
  extern __shared__ __align__(16) double lds[];
  const int lane = threadIdx.x;
#if DR_K3_FB_REBASIS
  AccDouble hand[39];        // this lane's sample: B(z) (the basis is computed again by the sample's lane pair in its pass)
#else
  AccDouble hand[36 + 39];   // this lane's sample: basis | B(z)
#endif
  double cs[11];
  DR_STAGE_BEGIN();
  {
    const int s = blockIdx.x * 64 + lane;
    const bool act = s < Bt;
    const int sc = act ? s : Bt - 1;
    double e[3][3][4];
    {
      double nb[4][9];
      fivepoint_basis_minimal<T>(samples + (size_t)sc * 20, weights ? weights + (size_t)sc * 5 : nullptr, nb);
#if !DR_K3_FB_REBASIS
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int q = 0; q < 9; ++q) hand[9 * t + q].put(nb[t][q]);
#endif
      basis_to_entries(nb, e);
    }
    double X[6][10];
    const bool ok = constraints_reduce_front<NisterOrder, 4>(e, LaneWs{lds + lane, 64}, 1.0, X);
    double bz[39];
    nister_bz_det(X, bz, cs);
#pragma unroll
    for (int k = 0; k < 39; ++k) hand[(DR_K3_FB_REBASIS ? 0 : 36) + k].put(bz[k]);
    if (!ok || !act) {   // 1 + z^10: no real root in either half of the search, no bracket in the wave's queues
#pragma unroll
      for (int i = 0; i <= 10; ++i) cs[i] = (i == 0 || i == 10) ? 1.0 : 0.0;
    }
  }
  DR_STAGE(2);
#pragma unroll 1
  for (int p = 0; p < 2; ++p) {
    const size_t s0 = (size_t)blockIdx.x * 64 + 32 * p;
    if (s0 >= (size_t)Bt) break;
    wave_lds_order();   // the front stage / the previous pass is done with the LDS
#if DR_K3_FB_REBASIS
    if ((lane >> 5) == p) {
      double *img = lds + 36 * 32 + (lane & 31);   // FinishQueue layout: B(z) behind the basis
#pragma unroll
      for (int k = 0; k < 39; ++k) img[k * 32] = hand[k].get();
    }
    {
      const size_t sj = s0 + (lane >> 1) < (size_t)Bt ? s0 + (lane >> 1) : (size_t)Bt - 1;
      double nb[4][9];
      fivepoint_basis_minimal<T>(samples + sj * 20, weights ? weights + sj * 5 : nullptr, nb);
      park_basis(FinishQueue(lds), nb, lane);
    }
#else
    if ((lane >> 5) == p) {
      double *img = lds + (lane & 31);   // FinishQueue layout: basis element e of sample j at [e * 32 + j], then B(z)
#pragma unroll
      for (int k = 0; k < 36 + 39; ++k) img[k * 32] = hand[k].get();
    }
#endif
    double csp[11];
#pragma unroll
    for (int i = 0; i <= 10; ++i) csp[i] = lane_read(cs[i], 32 * p + (lane >> 1));
    wave_lds_order();
    nister_back_pair<T>(csp, lds, lane, s0, s0 + (lane >> 1) < (size_t)Bt, models, valid, models64);
  }
}

// Stewenius: what a sample hands over is its reduced rows 0,1,2,4,5,7 (60 doubles, in accumulation registers) and the polynomial;
// the null-space basis (36 more doubles: with them the back stage spills) is computed again by the sample's lane pair in its
// pass -- 700 of the ~7 600 front instructions the lane pairs no longer execute twice.
template <typename T>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) void stewenius5_fb_kernel(
    const T *__restrict__ samples, int Bt, T *__restrict__ models, uint8_t *__restrict__ valid, PairGate gate, int per_pair) {
  if (gate.closed_range((blockIdx.x * 64) / per_pair, (min(blockIdx.x * 64 + 64, Bt) - 1) / per_pair)) return;
  extern __shared__ __align__(16) double lds[];
  const int lane = threadIdx.x;
  constexpr int kGOff = SturmWs<10>::kDoubles > FinishQueue::kDoubles ? SturmWs<10>::kDoubles : FinishQueue::kDoubles;   // reduced rows of a pass
  AccDouble hand[60];
  double cs[11];
  {
    const int s = blockIdx.x * 64 + lane;
    const bool act = s < Bt;
    const int sc = act ? s : Bt - 1;
    double e[3][3][4];
    {
      double nb[4][9];
      fivepoint_basis_minimal<T>(samples + (size_t)sc * 20, nullptr, nb);
      basis_to_entries(nb, e);
    }
    double g[6][10];
    bool ok;
    {
      double X[10][10];
      ok = constraints_reduce_front<GrevlexOrder, 0>(e, LaneWs{lds + lane, 64}, 2.0, X);
      const int src[6] = {0, 1, 2, 4, 5, 7};
#pragma unroll
      for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int c = 0; c < 10; ++c) g[r][c] = X[src[r]][c];
    }
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
      for (int c = 0; c < 10; ++c) hand[10 * r + c].put(g[r][c]);
    stewenius_charpoly(g, cs);
    if (!ok || !act) {
#pragma unroll
      for (int i = 0; i <= 10; ++i) cs[i] = (i == 0 || i == 10) ? 1.0 : 0.0;
    }
  }
#pragma unroll 1
  for (int p = 0; p < 2; ++p) {
    const size_t s0 = (size_t)blockIdx.x * 64 + 32 * p;
    if (s0 >= (size_t)Bt) break;
    const int half = lane & 1, j = lane >> 1;
    const bool active = s0 + j < (size_t)Bt;
    double csp[11];
#pragma unroll
    for (int i = 0; i <= 10; ++i) csp[i] = lane_read(cs[i], 32 * p + j);
    wave_lds_order();   // the front stage / the previous pass is done with the LDS
    if ((lane >> 5) == p) {   // the reduced rows of this pass' samples: behind the root-search workspace and the candidate queue
      double *img = lds + kGOff + (lane & 31);
#pragma unroll
      for (int k = 0; k < 60; ++k) img[k * 32] = hand[k].get();
    }
    double roots[10];
    int nroots;
#if DR_K3_STURM
    real_roots_half_sturm<10>(csp, half != 0, roots, nroots, lds, lane);
#else
    real_roots_half_wave<10>(csp, half != 0, roots, nroots, lds, lane);
#endif
    if (!active) nroots = 0;
    const FinishQueue fq(lds);
    {
      double nb[4][9];
      fivepoint_basis_minimal<T>(samples + (active ? s0 + j : (size_t)Bt - 1) * 20, nullptr, nb);
      wave_lds_order();   // the root-search workspace is dead: the basis takes its place
      park_basis(fq, nb, lane);
    }
    double xs[10], ys[10], zs[10];
    unsigned cand;
    {
      double g[6][10];
#pragma unroll
      for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int c = 0; c < 10; ++c) g[r][c] = lds[kGOff + (10 * r + c) * 32 + j];
      stewenius_xyz_of_roots(g, roots, nroots, xs, ys, zs, cand);
    }
    wave_lds_order();
    balanced_finish<T>(fq, lane, nroots, xs, ys, zs, cand, s0, active, models, valid, nullptr);
  }
}

static inline bool aligned_out(const void *models, const void *valid) {
  return !DR_K3_STAGE_OUT || ((reinterpret_cast<uintptr_t>(models) & 15u) == 0 && (reinterpret_cast<uintptr_t>(valid) & 3u) == 0);
}

// SIMDs of the current device (4 per CU), cached per device: one process may drive several GPUs
static inline int device_simds() {
  static int simds[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 1024;
  if (!simds[dev]) {
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    simds[dev] = 4 * (cus > 0 ? cus : 256);
  }
  return simds[dev];
}

// samples per 64-lane block of the two-lanes-per-sample kernels: 32, or fewer (down to 4) while the grid stays within one block
// per SIMD -- a one-pair call (1024 samples) is 256 blocks of 4 samples instead of 32 blocks of 32 on 1024 SIMDs.  Always even
// (>= 4): the staged-output alignment of balanced_finish relies on s0 * 90 floats being a multiple of 16 bytes.
static inline int samples_per_block(int Bt) {
  int spb = 32;
  const int simds = device_simds();
  while (spb > 4 && (long)(Bt + spb / 2 - 1) / (spb / 2) <= simds) spb /= 2;
  return spb;   // 32, 16, 8 or 4: even, so a block's first output row (s0 * 90 floats, s0 * 10 bytes) keeps the staged-output alignment
}

// Two-phase kernels or lane pairs?  A pair-kernel block (32 samples) costs kPairCost, a two-phase block (64 samples) kFbCost
// (relative units from the measured kernels, profiles/r5_k3_variants.md); blocks run one per SIMD, so a launch takes
// ceil(blocks / SIMDs) rounds.  131 072 samples: 4 x pair against 2 x two-phase; 32 768 samples (one round of pair blocks
// on 1024 SIMDs) stay with the lane pairs, which also serve the small grids (samples_per_block).
#ifndef DR_K3_FB_COST_NISTER
#define DR_K3_FB_COST_NISTER 174   // per cent of a pair-kernel block (measured: 132.9 us in two rounds against 153.4 in four)
#endif
#ifndef DR_K3_FB_COST_STEW
#define DR_K3_FB_COST_STEW 183     // (171.8 us in two rounds against 187.6 in four)
#endif
static inline bool fivepoint_two_phase(int Bt, int fb_cost_pct) {
  const long simds = device_simds();
  const long rounds_pair = ((Bt + 31) / 32 + simds - 1) / simds;
  const long rounds_fb = ((Bt + 63) / 64 + simds - 1) / simds;
  return rounds_fb * fb_cost_pct < rounds_pair * 100;
}

// path: 0 = automatic (fivepoint_two_phase), 1 = lane pairs, 2 = two-phase
template <typename T>
int nister_launch(const T *samples, const T *weights, int Bt, int n, T *models, uint8_t *valid, hipStream_t st,
                  double *models64 = nullptr, int path = 0, PairGate gate = PairGate(), int per_pair = 1) {
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
    if (DR_K3_BALANCED && (path == 2 || (path == 0 && fivepoint_two_phase(Bt, DR_K3_FB_COST_NISTER)))) {
      static_assert(!DR_K3_BALANCED || 70 * 64 <= kNisterPairDoubles, "front stage: rows 0-6 of 64 right blocks");
      hipLaunchKernelGGL((nister5_fb_kernel<T>), dim3((Bt + 63) / 64), dim3(64), smem, st, samples, weights, Bt, models, valid,
                         models64, gate, per_pair);
      return check_launch("nister5_fb_kernel");
    }
    const int spb = samples_per_block(Bt);
    hipLaunchKernelGGL((nister5_pair_kernel<T>), dim3((Bt + spb - 1) / spb), dim3(64), smem, st, samples, weights, Bt, models,
                       valid, models64, spb, gate, per_pair);
    return check_launch("nister5_pair_kernel");
  }
  // n > 5 fallback (refit): one lane per sample, A^T A + eigenvectors in LDS (162 doubles)
  const size_t smem = sizeof(double) * kFiveWs * 64;
  hipLaunchKernelGGL((nister5_kernel<T>), dim3((Bt + 63) / 64), dim3(64), smem, st, samples, weights, Bt, n, models,
                     valid);
  return check_launch("nister5_kernel");
}

template <typename T>
int stewenius_launch(const T *samples, int Bt, T *models, uint8_t *valid, hipStream_t st, int path = 0, PairGate gate = PairGate(),
                     int per_pair = 1) {
  // the right 10x10 block of 32 samples, later the root-search workspace, then the candidate queue
  constexpr int kDoubles = (FinishQueue::kDoubles > 100 * 32) ? FinishQueue::kDoubles : 100 * 32;
  static_assert(RootWs<10>::kDoubles <= kDoubles && SturmWs<10>::kDoubles <= kDoubles, "root-search workspace");
  if (path == 2 || (path == 0 && fivepoint_two_phase(Bt, DR_K3_FB_COST_STEW))) {
    // front stage: rows 0-6 of 64 right blocks (35 KiB); back: root search / candidate queue + the reduced rows of 32 samples
    constexpr int kBack = (SturmWs<10>::kDoubles > FinishQueue::kDoubles ? SturmWs<10>::kDoubles : FinishQueue::kDoubles) + 60 * 32;
    constexpr int kFb = 70 * 64 > kBack ? 70 * 64 : kBack;
    static_assert(kFb * sizeof(double) <= 40960, "four blocks per CU");
    hipLaunchKernelGGL((stewenius5_fb_kernel<T>), dim3((Bt + 63) / 64), dim3(64), sizeof(double) * kFb, st, samples, Bt, models,
                       valid, gate, per_pair);
    return check_launch("stewenius5_fb_kernel");
  }
  const size_t smem = sizeof(double) * kDoubles;
  const int spb = samples_per_block(Bt);
  hipLaunchKernelGGL((stewenius5_pair_kernel<T>), dim3((Bt + spb - 1) / spb), dim3(64), smem, st, samples, Bt, models, valid, spb,
                     gate, per_pair);
  return check_launch("stewenius5_pair_kernel");
}

// ---- test hook: the real-root search of the two-lanes-per-sample kernels on given degree-10 polynomials -------------------
// One lane pair per polynomial (even lane: |z| <= 1, odd lane: |z| > 1 through the reversed polynomial), exactly as the solver
// kernels call it.  method 0 = derivative chain (real_roots_half_wave), 1 = Sturm isolation (real_roots_half_sturm).
// DR_DBG_ROOTS_WAVES (round 6 experiment): waves per SIMD the standalone root-search kernel is compiled for -- what a separate
// root-search LAUNCH between a front and a finish kernel would run at (scratch/roots_occupancy.py)
#ifndef DR_DBG_ROOTS_WAVES
#define DR_DBG_ROOTS_WAVES 1
#endif
template <int kMethod>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(DR_DBG_ROOTS_WAVES, DR_DBG_ROOTS_WAVES))) void debug_roots10_kernel(const double *__restrict__ coef, int n, double *__restrict__ roots,
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

// THE f32 five-point entries (round 6: the `_hp`, `_path_` and `_gated_` twins of rounds 3-5 folded in; every argument after
// `valid` is optional):
//   models_f64 (Nister, minimal samples): the f64 models next to the f32 ones from the same launch (train mode);
//   path (minimal samples): 0 automatic, 1 lane-pair kernel, 2 two-phase kernel;
//   gate_iters / gate_max_iters + per_pair (rounds > 1 of a multi-round test-mode call; Bt = pairs x per_pair minimal samples):
//   blocks whose samples all belong to pairs with iters >= max_iters (the state dr_ransac_update keeps) return at once, their
//   models / valid keep their contents.
int dr_solve_nister5_f32(const float *samples, const float *weights, int Bt, int n, float *models, double *models_f64, uint8_t *valid,
                         int path, int per_pair, const int32_t *gate_iters, const double *gate_max_iters, void *stream) {
  DR_REQUIRE(samples && models && valid, "null pointer");
  DR_REQUIRE(dr::aligned_out(models, valid), "models must be 16-byte aligned and valid 4-byte aligned (whole-line output stores)");
  DR_REQUIRE(Bt > 0 && n >= 5, "need Bt > 0 and n >= 5 points per sample");
  DR_REQUIRE(path >= 0 && path <= 2, "path: 0 automatic, 1 lane pairs, 2 two-phase");
  DR_REQUIRE(n == 5 || (!models_f64 && path == 0 && !gate_iters), "f64 models, explicit paths and gates serve minimal samples (n = 5)");
  DR_REQUIRE((gate_iters == nullptr) == (gate_max_iters == nullptr), "gate: both pointers or neither");
  DR_REQUIRE(!gate_iters || (per_pair > 0 && Bt % per_pair == 0), "gate: need Bt = pairs x per_pair");
  dr::PairGate gate;
  gate.iters = gate_iters;
  gate.max_iters = gate_max_iters;
  return dr::nister_launch<float>(samples, weights, Bt, n, models, valid, (hipStream_t)stream, models_f64, path, gate,
                                  gate_iters ? per_pair : 1);
}
int dr_solve_nister5_f64(const double *samples, const double *weights, int Bt, int n, double *models,
                         uint8_t *valid, void *stream) {
  DR_REQUIRE(samples && models && valid, "null pointer");
  DR_REQUIRE(Bt > 0 && n >= 5, "need Bt > 0 and n >= 5 points per sample");
  return dr::nister_launch<double>(samples, weights, Bt, n, models, valid, (hipStream_t)stream);
}
int dr_debug_real_roots10(const double *coef, int n, int method, double *roots, int32_t *counts, void *stream) {
  DR_REQUIRE(coef && roots && counts, "null pointer");
  DR_REQUIRE(n > 0 && (method == 0 || method == 1), "need n > 0 and method 0 (derivative chain) or 1 (Sturm)");
  constexpr int kD = dr::RootWs<10>::kDoubles > dr::SturmWs<10>::kDoubles ? dr::RootWs<10>::kDoubles : dr::SturmWs<10>::kDoubles;
  const size_t smem = sizeof(double) * kD;
  if (method == 1)   // (the Sturm search's own workspace, 19.7 KB: eight blocks per CU fit when the kernel is built for two waves per SIMD)
    hipLaunchKernelGGL((dr::debug_roots10_kernel<1>), dim3((n + 31) / 32), dim3(64), sizeof(double) * dr::SturmWs<10>::kDoubles,
                       (hipStream_t)stream, coef, n, roots, counts);
  else
    hipLaunchKernelGGL((dr::debug_roots10_kernel<0>), dim3((n + 31) / 32), dim3(64), smem, (hipStream_t)stream, coef, n, roots, counts);
  return dr::check_launch("debug_roots10_kernel");
}

int dr_solve_stewenius5_f32(const float *samples, int Bt, float *models, uint8_t *valid, int path, int per_pair,
                            const int32_t *gate_iters, const double *gate_max_iters, void *stream) {
  DR_REQUIRE(samples && models && valid, "null pointer");
  DR_REQUIRE(dr::aligned_out(models, valid), "models must be 16-byte aligned and valid 4-byte aligned (whole-line output stores)");
  DR_REQUIRE(Bt > 0, "need Bt > 0");
  DR_REQUIRE(path >= 0 && path <= 2, "path: 0 automatic, 1 lane pairs, 2 two-phase");
  DR_REQUIRE((gate_iters == nullptr) == (gate_max_iters == nullptr), "gate: both pointers or neither");
  DR_REQUIRE(!gate_iters || (per_pair > 0 && Bt % per_pair == 0), "gate: need Bt = pairs x per_pair");
  dr::PairGate gate;
  gate.iters = gate_iters;
  gate.max_iters = gate_max_iters;
  return dr::stewenius_launch<float>(samples, Bt, models, valid, (hipStream_t)stream, path, gate, gate_iters ? per_pair : 1);
}
int dr_solve_stewenius5_f64(const double *samples, int Bt, double *models, uint8_t *valid, void *stream) {
  DR_REQUIRE(samples && models && valid, "null pointer");
  DR_REQUIRE(Bt > 0, "need Bt > 0");
  return dr::stewenius_launch<double>(samples, Bt, models, valid, (hipStream_t)stream);
}

}  // extern "C"
