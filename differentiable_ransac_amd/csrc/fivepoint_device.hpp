// Device-side building blocks of the five-point solvers, shared by solve_fivepoint.hip (minimal / per-sample kernels)
// and refit.hip (K7: one cooperative block per image pair).
#pragma once
#include "solver_common.hpp"

namespace dr {

// 1: the balanced final stage evaluates the ten constraints on every candidate BEFORE the first Gauss-Newton step and only
// steps the candidates above the stopping tolerance (0: one unconditional step for everybody, rounds 1-2), and the
// verification reads the residual norm the last step left in LDS instead of evaluating the constraints again.
// dr_solve_nister5_f32 at 32 x 1024 samples: 73.8 -> 68.7 us; same valid flags on 327 680 slots, 98.5 % of the models
// bit-identical, the rest within 1.7e-6 (the skipped step moved them below f32 rounding); error against the true E unchanged
// (median 9.37e-7, 99.2 % below 1e-4).  Shorter bisection / Newton schedules of the root search, tried in the same pass,
// lose 0.03-1.1 % of the solutions (profiles/r3_k3_precheck.log): the schedule stays.
#ifndef DR_K3_PRECHECK
#define DR_K3_PRECHECK 1
#endif
// 1: the Gauss-Newton step forms the Jacobian of the nine trace constraints as (2 dG - tr(dG) I) E + (2 G - tr(G) I) H
// (G = E E^T, dG = H E^T + E H^T: three 3x3 products per direction instead of six), takes the residual from the same
// pieces and evaluates |r|^2 at the new point with the symmetric G formed once.  68.7 -> 64 us per 32 x 1024 samples with
// the pre-check; all 158 800 f32 models of the A/B set bit-identical (the f64 differences round away), same valid flags.
// 1: the f32 models and validity bytes of a block's 32 samples (11.5 KB + 320 B, contiguous in the output) are assembled in
// LDS -- identity pattern first, verified solutions over it -- and written out as whole 16-byte pieces of consecutive lanes:
// 12 + 1 coalesced store instructions per wave instead of ~120 scattered ones (9 dwords per verified solution and per
// identity filler, 360 bytes apart from lane to lane: every dword its own 64-byte request).  Measured: 55.9 -> 57.7 us per
// 32 x 1024 samples, step 0.973 vs 0.974 ms -- the scattered stores are not what the final stage waits for (it is ~7 k
// instructions per wave: the per-lane start vectors, 2.3 Gauss-Newton rounds, three verification rounds).  Off; bit-identical.
#ifndef DR_K3_STAGE_OUT
#define DR_K3_STAGE_OUT 0
#endif
#ifndef DR_K3_TOL2_F32
#define DR_K3_TOL2_F32 1e-16   // squared residual norm at which a candidate of the f32 entry points stops iterating (balanced final stage).
                              // 1e-17 until round 4; 1e-16 measured then: 57.1 -> 54.6 us per 32 x 1024 samples, same valid flags and error
                              // statistics, 24 of 158 874 models move by up to 7e-6 (ill-conditioned ones: the skipped step mattered) --
                              // far inside the 1e-4 contract of an f32 model.  Adopted in round 5: a Gauss-Newton step is 1 400 instructions
                              // that 0.5 % of the candidates ask for, but 40 % of the waves then run (cold code: ~11 us per execution)
#endif
#ifndef DR_K3_PRECHECK_F64
#define DR_K3_PRECHECK_F64 1   // 1: the residual pre-check also with the f64 stopping tolerance (train mode): with converged roots
                               // most candidates are at rounding level already -- train step 0.2898 -> 0.2865 ms
#endif
#ifndef DR_K3_JAC2
#define DR_K3_JAC2 1
#endif
// 1: the root search of the two-lanes-per-sample kernels deals the brackets that hold a sign change out over the wave
// (real_roots_half_wave) instead of refining every bracket in every lane
#ifndef DR_K3_WAVE_ROOTS
#define DR_K3_WAVE_ROOTS 1
#endif

constexpr int kFiveWs = 162;   // doubles of LDS per lane: B block (100) for the minimal path, A^T A + V (162) for n > 5

// ---- null-space basis ---------------------------------------------------------------------------------
// minimal: Householder QR of the 5x9 system (registers).  nb[t][0..8], t = 0..3
template <typename T>
__device__ __forceinline__ void fivepoint_basis_minimal(const T *__restrict__ pts, const T *__restrict__ wts,
                                                        double (&nb)[4][9]) {
  double A[5][9];
#pragma unroll
  for (int r = 0; r < 5; ++r) {
    const double w = wts ? (double)wts[r] : 1.0;
    epipolar_row_5pt((double)pts[4 * r], (double)pts[4 * r + 1], (double)pts[4 * r + 2], (double)pts[4 * r + 3], w, A[r]);
  }
  null_space_qr<5>(A, nb);
}

// non-minimal (n > 5; nister.py:64-65 runs the minimal code on all points): the four eigenvectors of A^T A
// with the smallest eigenvalues, by cyclic Jacobi in LDS.  Order: nb[0] <-> 4th smallest ... nb[3] <-> smallest,
// which is the order torch.linalg.svd's Vh[-4:] has.
template <typename T>
__device__ __forceinline__ void fivepoint_basis_nonminimal(const T *__restrict__ pts, const T *__restrict__ wts, int n,
                                           const LaneWs &ws, double (&nb)[4][9]) {
  LaneWs A{ws.base}, V{ws.base + 81 * 64};
  for (int e = 0; e < 81; ++e) A[e] = 0.0;
  for (int r = 0; r < n; ++r) {
    double row[9];
    const double w = wts ? (double)wts[r] : 1.0;
    epipolar_row_5pt((double)pts[4 * r], (double)pts[4 * r + 1], (double)pts[4 * r + 2], (double)pts[4 * r + 3], w, row);
#pragma unroll
    for (int i = 0; i < 9; ++i)
#pragma unroll
      for (int j = 0; j < 9; ++j) A[i * 9 + j] += row[i] * row[j];
  }
  jacobi_eig_lds<9>(A, V);
  unsigned used = 0;
  for (int t = 3; t >= 0; --t) {  // t = 3 takes the smallest
    int best = 0;
    double bv = INFINITY;
    for (int i = 0; i < 9; ++i) {
      const double ev = A[i * 9 + i];
      if (!((used >> i) & 1u) && ev < bv) { bv = ev; best = i; }
    }
    used |= 1u << best;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      const double v = V[i * 9 + best];
#pragma unroll
      for (int tt = 0; tt < 4; ++tt)
        if (tt == t) nb[tt][i] = v;
    }
  }
}

// entry polynomial (i,j) of E~(x,y,z) = x N0 + y N1 + z N2 + N3 is (N0..N3)[3j+i]  (nister.py:123, stewenius.py:53)
__device__ __forceinline__ void basis_to_entries(const double (&nb)[4][9], double (&e)[3][3][4]) {
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int t = 0; t < 4; ++t) e[i][j][t] = nb[t][3 * j + i];
}

template <typename T>
__device__ __forceinline__ void write_identity(T *__restrict__ dst) {
#pragma unroll
  for (int q = 0; q < 9; ++q) dst[q] = T(q % 4 == 0 ? 1 : 0);
}

// Gauss-Newton refinement of (x, y, z) on the ten defining constraints 2EE^TE - tr(EE^T)E = 0, det E = 0 with
// E = x N0 + y N1 + z N2 + N3.  The hidden-variable resultant (Nister) and the action-matrix eigen-problem
// (Stewenius) can lose digits when roots cluster; the constraint system itself is well conditioned wherever
// the essential matrix is, so two iterations restore full f64 accuracy (and are basis independent).
__device__ __forceinline__ void essential_residual(const double (&E)[9], double (&r)[10]) {
  double G[9];  // E E^T
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) G[3 * i + j] = E[3 * i] * E[3 * j] + E[3 * i + 1] * E[3 * j + 1] + E[3 * i + 2] * E[3 * j + 2];
  const double tr = G[0] + G[4] + G[8];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
      r[3 * i + j] = 2.0 * (G[3 * i] * E[j] + G[3 * i + 1] * E[3 + j] + G[3 * i + 2] * E[6 + j]) - tr * E[3 * i + j];
  r[9] = E[0] * (E[4] * E[8] - E[5] * E[7]) - E[1] * (E[3] * E[8] - E[5] * E[6]) + E[2] * (E[3] * E[7] - E[4] * E[6]);
}

// |r(E)|^2 alone, with the symmetric G = E E^T formed once (6 entries) and r = (2 G - tr(G) I) E: ~80 instead of ~120 FMAs
__device__ __forceinline__ double essential_residual_norm2(const double (&E)[9]) {
  const double g00 = E[0] * E[0] + E[1] * E[1] + E[2] * E[2], g01 = E[0] * E[3] + E[1] * E[4] + E[2] * E[5];
  const double g02 = E[0] * E[6] + E[1] * E[7] + E[2] * E[8], g11 = E[3] * E[3] + E[4] * E[4] + E[5] * E[5];
  const double g12 = E[3] * E[6] + E[4] * E[7] + E[5] * E[8], g22 = E[6] * E[6] + E[7] * E[7] + E[8] * E[8];
  const double tr = g00 + g11 + g22;
  const double a00 = 2.0 * g00 - tr, a11 = 2.0 * g11 - tr, a22 = 2.0 * g22 - tr, a01 = 2.0 * g01, a02 = 2.0 * g02, a12 = 2.0 * g12;
  double n = 0;
#pragma unroll
  for (int jx = 0; jx < 3; ++jx) {
    const double r0 = a00 * E[jx] + a01 * E[3 + jx] + a02 * E[6 + jx];
    const double r1 = a01 * E[jx] + a11 * E[3 + jx] + a12 * E[6 + jx];
    const double r2 = a02 * E[jx] + a12 * E[3 + jx] + a22 * E[6 + jx];
    n += r0 * r0 + r1 * r1 + r2 * r2;
  }
  const double det = E[0] * (E[4] * E[8] - E[5] * E[7]) + E[1] * (E[5] * E[6] - E[3] * E[8]) + E[2] * (E[3] * E[7] - E[4] * E[6]);
  return n + det * det;
}

// Gauss-Newton in HOMOGENEOUS coordinates: E = sum_k u_k N_k with |u| = 1 (so |E|_F = 1: the basis is orthonormal).
// r(E) is homogeneous of degree 3, hence J u = 3 r ~ 0 and the normal matrix is singular along u; adding u u^T picks the
// step orthogonal to u.  No chart, no scaling problem when a solution has a vanishing N3 component (|z| -> infinity), and
// no per-root permutation of the basis (everything is statically indexed).
// `tol2`: squared residual norm at which a sample stops iterating -- 1e-28 for f64 output (rounding level of the
// unit-norm E; 1e-24 is not measurably faster), 1e-17 for f32 output (E to 3e-9, below the f32 rounding that follows;
// usually ONE Gauss-Newton step: K3 119.6 -> 106.3 us at C2 x 32 pairs).
// one Gauss-Newton step from u: un = the normalised new vector, n0 / n1 = squared residual norms before / after
__device__ __forceinline__ void polish_step(const double (&nb)[4][9], const double (&u)[4], double (&un)[4], double &n0_out, double &n1_out) {
  double E[9], r[10];
#pragma unroll
  for (int q = 0; q < 9; ++q) E[q] = u[0] * nb[0][q] + u[1] * nb[1][q] + u[2] * nb[2][q] + u[3] * nb[3][q];
#if !DR_K3_JAC2
  essential_residual(E, r);
  double n0 = 0;
#pragma unroll
  for (int q = 0; q < 10; ++q) n0 += r[q] * r[q];
#endif
  double J[4][10];
#if DR_K3_JAC2
  // d r[H] = (2 dG - tr(dG) I) E + (2 G - tr(G) I) H  with G = E E^T, dG = H E^T + E H^T : three 3x3 products per direction
  // instead of six
  double A2[9];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int jx = 0; jx < 3; ++jx)
      A2[3 * i + jx] = 2.0 * (E[3 * i] * E[3 * jx] + E[3 * i + 1] * E[3 * jx + 1] + E[3 * i + 2] * E[3 * jx + 2]);
  {
    const double trG = 0.5 * (A2[0] + A2[4] + A2[8]);
    A2[0] -= trG; A2[4] -= trG; A2[8] -= trG;
  }
  const double cof[9] = {E[4] * E[8] - E[5] * E[7], E[5] * E[6] - E[3] * E[8], E[3] * E[7] - E[4] * E[6],
                         E[2] * E[7] - E[1] * E[8], E[0] * E[8] - E[2] * E[6], E[1] * E[6] - E[0] * E[7],
                         E[1] * E[5] - E[2] * E[4], E[2] * E[3] - E[0] * E[5], E[0] * E[4] - E[1] * E[3]};
  // the residual itself from the same pieces: r = (2 G - tr(G) I) E, det E by the first row of cofactors
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int jx = 0; jx < 3; ++jx) r[3 * i + jx] = A2[3 * i] * E[jx] + A2[3 * i + 1] * E[3 + jx] + A2[3 * i + 2] * E[6 + jx];
  r[9] = E[0] * cof[0] + E[1] * cof[1] + E[2] * cof[2];
  double n0 = 0;
#pragma unroll
  for (int q = 0; q < 10; ++q) n0 += r[q] * r[q];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const double(&H)[9] = nb[k];
    double HEt[9], A1[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int jx = 0; jx < 3; ++jx)
        HEt[3 * i + jx] = H[3 * i] * E[3 * jx] + H[3 * i + 1] * E[3 * jx + 1] + H[3 * i + 2] * E[3 * jx + 2];
    const double trd = 2.0 * (HEt[0] + HEt[4] + HEt[8]);   // tr(dG)
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int jx = 0; jx < 3; ++jx) A1[3 * i + jx] = 2.0 * (HEt[3 * i + jx] + HEt[3 * jx + i]) - (i == jx ? trd : 0.0);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int jx = 0; jx < 3; ++jx)
        J[k][3 * i + jx] = A1[3 * i] * E[jx] + A1[3 * i + 1] * E[3 + jx] + A1[3 * i + 2] * E[6 + jx] +
                           A2[3 * i] * H[jx] + A2[3 * i + 1] * H[3 + jx] + A2[3 * i + 2] * H[6 + jx];
    double dd = 0;
#pragma unroll
    for (int q = 0; q < 9; ++q) dd += cof[q] * H[q];
    J[k][9] = dd;
  }
#else
  double EEt[9], EtE[9];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int jx = 0; jx < 3; ++jx) {
      EEt[3 * i + jx] = E[3 * i] * E[3 * jx] + E[3 * i + 1] * E[3 * jx + 1] + E[3 * i + 2] * E[3 * jx + 2];
      EtE[3 * i + jx] = E[i] * E[jx] + E[3 + i] * E[3 + jx] + E[6 + i] * E[6 + jx];
    }
  const double tr = EEt[0] + EEt[4] + EEt[8];
  const double cof[9] = {E[4] * E[8] - E[5] * E[7], E[5] * E[6] - E[3] * E[8], E[3] * E[7] - E[4] * E[6],
                         E[2] * E[7] - E[1] * E[8], E[0] * E[8] - E[2] * E[6], E[1] * E[6] - E[0] * E[7],
                         E[1] * E[5] - E[2] * E[4], E[2] * E[3] - E[0] * E[5], E[0] * E[4] - E[1] * E[3]};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const double(&H)[9] = nb[k];
    double HEt[9];
    double trEHt = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int jx = 0; jx < 3; ++jx)
        HEt[3 * i + jx] = H[3 * i] * E[3 * jx] + H[3 * i + 1] * E[3 * jx + 1] + H[3 * i + 2] * E[3 * jx + 2];
#pragma unroll
    for (int q = 0; q < 9; ++q) trEHt += E[q] * H[q];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int jx = 0; jx < 3; ++jx) {
        const double t1 = H[3 * i] * EtE[jx] + H[3 * i + 1] * EtE[3 + jx] + H[3 * i + 2] * EtE[6 + jx];     // H E^T E
        const double t2 = HEt[i] * E[jx] + HEt[3 + i] * E[3 + jx] + HEt[6 + i] * E[6 + jx];                  // E H^T E
        const double t3 = EEt[3 * i] * H[jx] + EEt[3 * i + 1] * H[3 + jx] + EEt[3 * i + 2] * H[6 + jx];     // E E^T H
        J[k][3 * i + jx] = 2.0 * (t1 + t2 + t3) - 2.0 * trEHt * E[3 * i + jx] - tr * H[3 * i + jx];
      }
    double dd = 0;
#pragma unroll
    for (int q = 0; q < 9; ++q) dd += cof[q] * H[q];
    J[k][9] = dd;
  }
#endif
  // (J^T J + u u^T) d = J^T r : 4x4 SPD, LDL^T without pivoting
  double a[4][4], g[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    g[i] = 0;
#pragma unroll
    for (int q = 0; q < 10; ++q) g[i] += J[i][q] * r[q];
#pragma unroll
    for (int jx = i; jx < 4; ++jx) {   // symmetric, also in floating point (same products, same order)
      double acc = u[i] * u[jx];
#pragma unroll
      for (int q = 0; q < 10; ++q) acc += J[i][q] * J[jx][q];
      a[i][jx] = acc;
      a[jx][i] = acc;
    }
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const double inv = frcp(a[c][c]);
#pragma unroll
    for (int rr = c + 1; rr < 4; ++rr) {
      const double f = a[rr][c] * inv;
#pragma unroll
      for (int k = c; k < 4; ++k) a[rr][k] -= f * a[c][k];
      g[rr] -= f * g[c];
    }
  }
  double d[4];
#pragma unroll
  for (int c = 3; c >= 0; --c) {
    double acc = g[c];
#pragma unroll
    for (int k = c + 1; k < 4; ++k) acc -= a[c][k] * d[k];
    d[c] = fdiv(acc, a[c][c]);
  }
  double nn = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) { un[k] = u[k] - d[k]; nn += un[k] * un[k]; }
  const double sc = frsqrt(nn);
#pragma unroll
  for (int k = 0; k < 4; ++k) un[k] *= sc;
  double E2[9], n1 = 0;
#pragma unroll
  for (int q = 0; q < 9; ++q) E2[q] = un[0] * nb[0][q] + un[1] * nb[1][q] + un[2] * nb[2][q] + un[3] * nb[3][q];
#if DR_K3_JAC2
  n1 = essential_residual_norm2(E2);
#else
  double r2[10];
  essential_residual(E2, r2);
#pragma unroll
  for (int q = 0; q < 10; ++q) n1 += r2[q] * r2[q];
#endif
  n0_out = n0;
  n1_out = n1;
}

__device__ __forceinline__ void polish_homog(const double (&nb)[4][9], double (&u)[4], bool live, double tol2) {
  // one iteration for everybody, then only waves that still hold an unconverged sample go on (max 8)
#pragma unroll 1
  for (int it = 0; it < 8; ++it) {
    if (it >= 1 && !__any(live)) break;
    double un[4], n0, n1;
    polish_step(nb, u, un, n0, n1);
    const bool better = n1 <= n0 && is_finite(n1);
    if (better && (live || it < 1)) {
#pragma unroll
      for (int k = 0; k < 4; ++k) u[k] = un[k];
    }
    // |E| = 1: stop at rounding level, or when the residual no longer shrinks geometrically
    live = live && better && (n1 > tol2) && (n1 < 0.25 * n0);
  }
}

// Final stage shared by both solvers: homogeneous polish of the coefficient vector (x, y, z, 1)/|.|, then a
// VERIFICATION of the ten constraints on the (unit-norm) matrix.  `valid` therefore means "checked essential matrix
// through the five points", not "the root finder said so".
// `dst64` (optional): the same model in f64 as a second output (train mode keeps it for the backward); it also selects the
// f64 stopping tolerance.
// core: polish + verification; E = the unit-norm matrix sum_k u_k N_k (row-major in the basis' own entry order)
__device__ __forceinline__ bool finish_core(const double (&nb)[4][9], double x, double y, double z, bool candidate, double tol2,
                                            double (&E)[9]) {
  const double inv = frsqrt(x * x + y * y + z * z + 1.0);
  double u[4] = {x * inv, y * inv, z * inv, inv};
  bool good = candidate && is_finite(u[0]) && is_finite(u[1]) && is_finite(u[2]) && is_finite(u[3]);
  if (!good) { u[0] = 0.5; u[1] = 0.5; u[2] = 0.5; u[3] = 0.5; }
  polish_homog(nb, u, good, tol2);
  double r[10];
#pragma unroll
  for (int q = 0; q < 9; ++q) E[q] = u[0] * nb[0][q] + u[1] * nb[1][q] + u[2] * nb[2][q] + u[3] * nb[3][q];
  essential_residual(E, r);
  double rn = 0;
#pragma unroll
  for (int q = 0; q < 10; ++q) rn += r[q] * r[q];
  return good && is_finite(rn) && rn <= 1e-14;
}

template <typename T>
__device__ __forceinline__ void store_model(const double (&E)[9], T *__restrict__ dst, double *__restrict__ dst64) {
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int jx = 0; jx < 3; ++jx) {
      dst[3 * i + jx] = (T)E[3 * jx + i];   // stored transposed (nister.py:407)
      if (dst64) dst64[3 * i + jx] = E[3 * jx + i];
    }
}

template <typename T>
__device__ __forceinline__ bool finish_solution(const double (&nb)[4][9], double x, double y, double z, bool candidate,
                                                T *__restrict__ dst, bool store, double *__restrict__ dst64 = nullptr) {
  double E[9];
  const bool good = finish_core(nb, x, y, z, candidate, (sizeof(T) == 4 && !dst64) ? 1e-17 : 1e-28, E);
  if (good && store) store_model<T>(E, dst, dst64);
  return good;
}

// ---- balanced final stage of the two-lanes-per-sample kernels -------------------------------------------------
// After the root search a lane holds 0..10 candidate solutions (x, y, z) of its half of its sample; a wave's 64 lanes hold
// ~130 of them in total, but the lane with the most has 6-7, and a loop "every lane finishes its own" runs that long
// with two thirds of the lanes idle.  Here the candidates of the whole wave go into one LDS queue (in lane order, then
// root order) and are dealt out 64 at a time: lane l of round r polishes and verifies candidate 64 r + l, whichever
// sample it came from (the sample's null-space basis waits in LDS: `nb_lds[(9 t + q) * 32 + j]`, sample j of the block).
// Output slots are what the per-lane loop produced: the verified solutions of half 0 fill a sample's slots upwards from 0
// in root order, those of half 1 downwards from 9, eye(3) in between (rank of a candidate among the verified ones of its
// (sample, half) = verified ones before it in this round, by ballot, + the running count kept in LDS).
// Block = ONE wave, so LDS traffic is ordered by program order and no barrier is needed.
struct FinishQueue {
  const double *nb_src;   // basis of sample j, element e = 9 t + q:  nb_src[e * 32 + j]  (LDS, element-major: conflict-free)
  double *nb_lds;
  double *u;         // 4 x 320: the candidates' coefficient vectors (unit 4-vectors; (0.5, 0.5, 0.5, 0.5) for non-finite ones)
  uint16_t *meta;    // 320: source lane | finite-candidate flag << 6
  int *cnt;          // 64: verified solutions so far of (sample, half) = source lane
  uint16_t *live;    // 2 x 320: candidates that need another Gauss-Newton step (this round | next round)
  double *rn;        // 320: squared norm of the ten constraints at the candidate's current vector (DR_K3_PRECHECK)
  float *stage;      // 32 x 90: the block's f32 output, assembled here and written out in whole lines (DR_K3_STAGE_OUT)
  uint8_t *vstage;   // 32 x 10 validity bytes of the same
  static constexpr int kQueueDoubles = 4 * 320 + 320 / 4 + 64 / 2 + 2 * 320 / 4 + 320 + (DR_K3_STAGE_OUT ? 32 * 90 / 2 + 320 / 8 : 0);
  static constexpr int kDoubles = 36 * 32 + kQueueDoubles;
  __device__ __forceinline__ explicit FinishQueue(double *lds)
      : nb_src(lds), nb_lds(lds), u(lds + 36 * 32), meta(reinterpret_cast<uint16_t *>(lds + 36 * 32 + 4 * 320)),
        cnt(reinterpret_cast<int *>(lds + 36 * 32 + 4 * 320 + 320 / 4)),
        live(reinterpret_cast<uint16_t *>(lds + 36 * 32 + 4 * 320 + 320 / 4 + 64 / 2)),
        rn(lds + 36 * 32 + 4 * 320 + 320 / 4 + 64 / 2 + 2 * 320 / 4),
        stage(reinterpret_cast<float *>(lds + 36 * 32 + 4 * 320 + 320 / 4 + 64 / 2 + 2 * 320 / 4 + 320)),
        vstage(reinterpret_cast<uint8_t *>(lds + 36 * 32 + 4 * 320 + 320 / 4 + 64 / 2 + 2 * 320 / 4 + 320 + 32 * 90 / 2)) {}
  __device__ __forceinline__ void load_basis(int j, double (&nb)[4][9]) const {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int qq = 0; qq < 9; ++qq) nb[t][qq] = nb_src[(9 * t + qq) * 32 + j];
  }
};

__device__ __forceinline__ void park_basis(const FinishQueue &fq, const double (&nb)[4][9], int lane) {
  if ((lane & 1) == 0) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int qq = 0; qq < 9; ++qq) fq.nb_lds[(9 * t + qq) * 32 + (lane >> 1)] = nb[t][qq];
  }
}

// xs/ys/zs[0..n-1]: this lane's candidates (dense), cand: bit i = candidate i is finite / solvable.  s0 = first sample of
// the block.  Lanes of samples >= Bt pass n = 0.
// Three phases, each 64 candidates at a time: (A) the first Gauss-Newton step of every candidate; (B) further steps for
// the candidates that are not yet at the stopping tolerance -- they go through a second, much shorter queue, so a slow
// candidate does not hold 63 finished ones (per candidate the sequence of steps is exactly polish_homog's); (C)
// verification, rank, store.
template <typename T>
__device__ __forceinline__ void balanced_finish(const FinishQueue &fq, int lane, int n, const double (&xs)[10], const double (&ys)[10],
                                                const double (&zs)[10], unsigned cand, size_t s0, bool active,
                                                T *__restrict__ models, uint8_t *__restrict__ valid, double *__restrict__ models64) {
  DR_STAGE_BEGIN();
  // exclusive prefix sum of n over the wave
  int incl = n;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int o = __shfl_up(incl, d, 64);
    incl += (lane >= d) ? o : 0;
  }
  // the queues hold 320 entries = 10 per sample: a degree-10 polynomial has at most 10 real roots, but each half of a sample's
  // root search (|z| <= 1, |z| > 1) may report up to 10 on its own when rounding splits a multiple root into spurious sign
  // changes.  Entries past the end are dropped (never written, never counted) instead of overrunning u / meta / cnt.
  const int total = min(__shfl(incl, 63, 64), 320);
  const int off = incl - n;
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    if (!__any(i < n)) continue;
    // start vector (x, y, z, 1) / |.| ; a non-finite candidate keeps a harmless placeholder and never counts
    const double x = xs[i], y = ys[i], z = zs[i];
    const double inv = frsqrt(x * x + y * y + z * z + 1.0);
    double u0[4] = {x * inv, y * inv, z * inv, inv};
    const bool good = ((cand >> i) & 1u) && is_finite(u0[0]) && is_finite(u0[1]) && is_finite(u0[2]) && is_finite(u0[3]);
    if (i < n && off + i < 320) {
#pragma unroll
      for (int k = 0; k < 4; ++k) fq.u[k * 320 + off + i] = good ? u0[k] : 0.5;
      fq.meta[off + i] = (uint16_t)(lane | (good ? 1u << 6 : 0u));
    }
  }
  fq.cnt[lane] = 0;
  wave_lds_order();
  DR_STAGE(16);   // start vectors + queue
  const double tol2 = (sizeof(T) == 4 && !models64) ? DR_K3_TOL2_F32 : 1e-28;
  auto mbcnt = [](unsigned long long bm) {
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bm, 0u));
  };
#if DR_K3_PRECHECK
  // ---- (A') residual of every candidate as the root search left it; only those above the stopping tolerance queue for
  // Gauss-Newton steps (the others are final: the same rule that ends the iteration after a step).  (DR_K3_PRECHECK_F64 = 0:
  // with the f64 tolerance of train mode everybody queues unseen.)  Every queued candidate's residual norm is
  // (re)written by its steps, and (C) reads it back instead of evaluating the ten constraints a last time.
  int nlive = 0;
  const bool precheck = DR_K3_PRECHECK_F64 || tol2 > 1e-20;
#pragma unroll 1
  for (int base = 0; base < total; base += 64) {
    const int e = base + lane;
    const bool has = e < total;
    const int ec = has ? e : total - 1;
    const unsigned m = fq.meta[ec];
    bool lv = has && ((m >> 6) & 1u);
    if (precheck) {
      double nb[4][9], u[4], E[9];
      fq.load_basis((m & 63) >> 1, nb);
#pragma unroll
      for (int k = 0; k < 4; ++k) u[k] = fq.u[k * 320 + ec];
#pragma unroll
      for (int q = 0; q < 9; ++q) E[q] = u[0] * nb[0][q] + u[1] * nb[1][q] + u[2] * nb[2][q] + u[3] * nb[3][q];
#if DR_K3_JAC2
      const double n0 = essential_residual_norm2(E);
#else
      double r[10];
      essential_residual(E, r);
      double n0 = 0;
#pragma unroll
      for (int q = 0; q < 10; ++q) n0 += r[q] * r[q];
#endif
      lv = lv && !(n0 <= tol2);
      if (has) fq.rn[e] = n0;
    }
    const unsigned long long bm = __ballot(lv);
    if (lv) fq.live[nlive + mbcnt(bm)] = (uint16_t)e;
    nlive += __popcll(bm);
  }
  wave_lds_order();
  DR_STAGE(17);   // residual pre-check of every candidate
#ifdef DR_PROFILE_STAGES
  if (lane == 0) { atomicAdd(&::dr::g_stage_cycles[13], (unsigned long long)nlive); atomicAdd(&::dr::g_stage_cycles[14], (unsigned long long)total); }
  int _steps = 0;
#endif
  // ---- (B') steps 1..8 of the candidates that asked for them
#pragma unroll 1
  for (int it = 0; it < 8 && nlive > 0; ++it) {
    const uint16_t *cur = fq.live + 320 * (it & 1);
    uint16_t *nxt = fq.live + 320 * ((it + 1) & 1);
    int nnext = 0;
#pragma unroll 1
    for (int base = 0; base < nlive; base += 64) {
#ifdef DR_PROFILE_STAGES
      ++_steps;
#endif
      const bool has = base + lane < nlive;
      const int e = cur[has ? base + lane : base];
      double nb[4][9], u[4], un[4], n0, n1;
      fq.load_basis((fq.meta[e] & 63) >> 1, nb);
#pragma unroll
      for (int k = 0; k < 4; ++k) u[k] = fq.u[k * 320 + e];
      polish_step(nb, u, un, n0, n1);
      const bool better = n1 <= n0 && is_finite(n1);
      if (has && better) {
#pragma unroll
        for (int k = 0; k < 4; ++k) fq.u[k * 320 + e] = un[k];
      }
      if (has) fq.rn[e] = better ? n1 : n0;   // norm at the vector the candidate keeps
      const bool lv = has && better && (n1 > tol2) && (n1 < 0.25 * n0);
      const unsigned long long bm = __ballot(lv);
      if (lv) nxt[nnext + mbcnt(bm)] = (uint16_t)e;
      nnext += __popcll(bm);
    }
    nlive = nnext;
    wave_lds_order();
  }
#ifdef DR_PROFILE_STAGES
  if (lane == 0) atomicAdd(&::dr::g_stage_cycles[15], (unsigned long long)_steps);
#endif
  DR_STAGE(18);   // Gauss-Newton steps of the candidates that asked for them
#else
  // ---- (A) first step
  int nlive = 0;
#pragma unroll 1
  for (int base = 0; base < total; base += 64) {
    const int e = base + lane;
    const bool has = e < total;
    const int ec = has ? e : total - 1;
    const unsigned m = fq.meta[ec];
    double nb[4][9], u[4];
    fq.load_basis((m & 63) >> 1, nb);
#pragma unroll
    for (int k = 0; k < 4; ++k) u[k] = fq.u[k * 320 + ec];
    const bool good = has && ((m >> 6) & 1u);
    double un[4], n0, n1;
    polish_step(nb, u, un, n0, n1);
    const bool better = n1 <= n0 && is_finite(n1);
    const bool lv = good && better && (n1 > tol2) && (n1 < 0.25 * n0);
    if (has && better) {
#pragma unroll
      for (int k = 0; k < 4; ++k) fq.u[k * 320 + e] = un[k];
    }
    const unsigned long long bm = __ballot(lv);
    if (lv) fq.live[nlive + mbcnt(bm)] = (uint16_t)e;
    nlive += __popcll(bm);
  }
  wave_lds_order();
  // ---- (B) steps 2..8 of the candidates that asked for them
#pragma unroll 1
  for (int it = 1; it < 8 && nlive > 0; ++it) {
    const uint16_t *cur = fq.live + 320 * ((it - 1) & 1);
    uint16_t *nxt = fq.live + 320 * (it & 1);
    int nnext = 0;
#pragma unroll 1
    for (int base = 0; base < nlive; base += 64) {
      const bool has = base + lane < nlive;
      const int e = cur[has ? base + lane : base];
      double nb[4][9], u[4], un[4], n0, n1;
      fq.load_basis((fq.meta[e] & 63) >> 1, nb);
#pragma unroll
      for (int k = 0; k < 4; ++k) u[k] = fq.u[k * 320 + e];
      polish_step(nb, u, un, n0, n1);
      const bool better = n1 <= n0 && is_finite(n1);
      if (has && better) {
#pragma unroll
        for (int k = 0; k < 4; ++k) fq.u[k * 320 + e] = un[k];
      }
      const bool lv = has && better && (n1 > tol2) && (n1 < 0.25 * n0);
      const unsigned long long bm = __ballot(lv);
      if (lv) nxt[nnext + mbcnt(bm)] = (uint16_t)e;
      nnext += __popcll(bm);
    }
    nlive = nnext;
    wave_lds_order();
  }
#endif
  constexpr bool kStage = DR_K3_STAGE_OUT && sizeof(T) == 4;
  if (kStage) {
    // every slot starts as the eye(3) filler with validity 0
#pragma unroll 1
    for (int i = lane; i < 32 * 90 / 4; i += 64) {
      const int e0 = (4 * i) % 9;
      float4 v;
      v.x = (e0 % 4 == 0) ? 1.f : 0.f;
      v.y = (((e0 + 1) % 9) % 4 == 0) ? 1.f : 0.f;
      v.z = (((e0 + 2) % 9) % 4 == 0) ? 1.f : 0.f;
      v.w = (((e0 + 3) % 9) % 4 == 0) ? 1.f : 0.f;
      *reinterpret_cast<float4 *>(fq.stage + 4 * i) = v;
    }
    for (int i = lane; i < 320 / 4; i += 64) *reinterpret_cast<uint32_t *>(fq.vstage + 4 * i) = 0u;
    wave_lds_order();
  }
  // ---- (C) verification of the ten constraints, rank among the verified candidates of the same source lane, store
#pragma unroll 1
  for (int base = 0; base < total; base += 64) {
    const int e = base + lane;
    const bool has = e < total;
    const int ec = has ? e : total - 1;
    const unsigned m = fq.meta[ec];
    const int src = m & 63;
    const int j = src >> 1;
    double nb[4][9], u[4], E[9], r[10];
    fq.load_basis(j, nb);
#pragma unroll
    for (int k = 0; k < 4; ++k) u[k] = fq.u[k * 320 + ec];
#pragma unroll
    for (int q = 0; q < 9; ++q) E[q] = u[0] * nb[0][q] + u[1] * nb[1][q] + u[2] * nb[2][q] + u[3] * nb[3][q];
#if DR_K3_PRECHECK
    const double rn = fq.rn[ec];   // the residual of this very vector, from the pre-check or from the step that produced it
    (void)r;
#else
    essential_residual(E, r);
    double rn = 0;
#pragma unroll
    for (int q = 0; q < 10; ++q) rn += r[q] * r[q];
#endif
    const bool good = has && ((m >> 6) & 1u) && is_finite(rn) && rn <= 1e-14;
    const int gid = has ? src : 64 + lane;
    const int prev = __shfl_up(gid, 1, 64);
    const bool is_start = lane == 0 || prev != gid;
    const unsigned long long sm = __ballot(is_start), gm = __ballot(good);
    const unsigned long long upto = lane == 63 ? ~0ull : ((1ull << (lane + 1)) - 1ull);
    const int st = 63 - __clzll((long long)(sm & upto));
    const unsigned long long below = ((1ull << lane) - 1ull) & ~((1ull << st) - 1ull);
    const int rank = fq.cnt[src] + __popcll(gm & below);
    const bool is_end = lane == 63 || ((sm >> (lane + 1)) & 1ull);
    if (has && is_end) fq.cnt[src] = rank + (good ? 1 : 0);
    if (good && rank < 10) {
      const int slot = (src & 1) ? 9 - rank : rank;
      const size_t sm_ = s0 + (size_t)j;
      if (kStage) {
        float *dst = fq.stage + (j * 10 + slot) * 9;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int jx = 0; jx < 3; ++jx) dst[3 * i + jx] = (float)E[3 * jx + i];   // stored transposed (nister.py:407)
        fq.vstage[j * 10 + slot] = 1;
        if (models64) {
#pragma unroll
          for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int jx = 0; jx < 3; ++jx) models64[sm_ * 90 + 9 * slot + 3 * i + jx] = E[3 * jx + i];
        }
      } else {
#ifdef DR_K3_NOSTORE   // timing experiment: the final stage without its global stores (one data-dependent store keeps E alive)
        if (E[0] == 123.456) valid[sm_ * 10 + slot] = 1;
#else
        store_model<T>(E, models + sm_ * 90 + 9 * slot, models64 ? models64 + sm_ * 90 + 9 * slot : nullptr);
        valid[sm_ * 10 + slot] = 1;
#endif
      }
    }
  }
  DR_STAGE(19);   // verification, rank, store
  // eye(3) between the two halves' solutions
#ifndef DR_K3_NOSTORE
  if ((!kStage || models64) && active && (lane & 1) == 0) {
    const int lo = min(fq.cnt[lane], 10), hi = min(fq.cnt[lane + 1], 10);
    const size_t sm_ = s0 + (size_t)(lane >> 1);
    for (int s = min(lo, 10 - hi); s < 10 - hi; ++s) {
      if (!kStage) {
        write_identity<T>(models + sm_ * 90 + 9 * s);
        valid[sm_ * 10 + s] = 0;
      }
      if (models64) write_identity<double>(models64 + sm_ * 90 + 9 * s);
    }
  }
#endif
  DR_STAGE(20);   // identity fillers
  if (kStage) {
    // the block's output rows [s0, s0 + ns) x 90 floats and x 10 bytes are contiguous: whole 16-byte pieces, lane after lane
    wave_lds_order();
    const int ns = __popcll(__ballot(active)) >> 1;
    const int nf = ns * 90, nv = ns * 10;
    float *out = reinterpret_cast<float *>(models) + s0 * 90;    // 16-byte aligned: s0 is a multiple of 32
#pragma unroll 1
    for (int i = lane; i < 32 * 90 / 4; i += 64) {
      const float4 val = *reinterpret_cast<const float4 *>(fq.stage + 4 * i);
      if (4 * i + 3 < nf) *reinterpret_cast<float4 *>(out + 4 * i) = val;
      else {
        if (4 * i < nf) out[4 * i] = val.x;
        if (4 * i + 1 < nf) out[4 * i + 1] = val.y;
        if (4 * i + 2 < nf) out[4 * i + 2] = val.z;
      }
    }
    uint8_t *vout = valid + s0 * 10;                                // 4-byte aligned: s0 * 10 is a multiple of 320
    for (int i = lane; i < 320 / 4; i += 64) {
      const uint32_t w = *reinterpret_cast<const uint32_t *>(fq.vstage + 4 * i);
      if (4 * i + 3 < nv) *reinterpret_cast<uint32_t *>(vout + 4 * i) = w;
      else
        for (int b = 0; b < 4; ++b)
          if (4 * i + b < nv) vout[4 * i + b] = (uint8_t)(w >> (8 * b));
    }
  }
}

// ---- Nister: B(z) from the reduced rows, det B(z), roots, back-substitution -----------------------------
// kPair: two lanes share one sample -- `half` 0 searches |z| <= 1 and fills the slots upwards from 0, `half` 1 searches
// |z| > 1 and fills downwards from 9; the slots in between become eye(3).
template <typename T, bool kPair>
__device__ __forceinline__ void nister_finish(const double (&nb)[4][9], const double (&X)[6][10], bool ok, T *__restrict__ models,
                              uint8_t *__restrict__ valid, bool active, int half = 0,
                              double *__restrict__ models64 = nullptr) {
  // reduced rows e..j = rows 4..9, right block columns 10..19 hold (x z^2, x z, x | y z^2, y z, y | z^3, z^2, z, 1)
  // k = e - z f, l = g - z h, m = i - z j  ->  B(z) columns (x: deg 3, y: deg 3, 1: deg 4), ascending coefficients
  double bx[3][4], by[3][4], b1[3][5];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    double hi[10], lo[10];
#pragma unroll
    for (int c = 0; c < 10; ++c) {
      hi[c] = X[2 * r][c];
      lo[c] = X[2 * r + 1][c];
    }
    // hi = (a2 z^2 + a1 z + a0) with hi[0]=a2,hi[1]=a1,hi[2]=a0 ; minus z*(lo)
    bx[r][0] = hi[2];          bx[r][1] = hi[1] - lo[2]; bx[r][2] = hi[0] - lo[1]; bx[r][3] = -lo[0];
    by[r][0] = hi[5];          by[r][1] = hi[4] - lo[5]; by[r][2] = hi[3] - lo[4]; by[r][3] = -lo[3];
    b1[r][0] = hi[9];          b1[r][1] = hi[8] - lo[9]; b1[r][2] = hi[7] - lo[8]; b1[r][3] = hi[6] - lo[7];
    b1[r][4] = -lo[6];
  }
  // det = sum_r c2[r] * cofactor_r ; minors of columns (x,y): degree 6
  double cs[11];
#pragma unroll
  for (int i = 0; i < 11; ++i) cs[i] = 0;
  auto minor_acc = [&](int a, int b, int r, double sgn) {
    double mn[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) mn[i] = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) mn[i + j] += bx[a][i] * by[b][j] - bx[b][i] * by[a][j];
#pragma unroll
    for (int i = 0; i < 7; ++i)
#pragma unroll
      for (int j = 0; j < 5; ++j) cs[i + j] += sgn * mn[i] * b1[r][j];
  };
  minor_acc(1, 2, 0, 1.0);
  minor_acc(0, 2, 1, -1.0);
  minor_acc(0, 1, 2, 1.0);

  double roots[10];
  int nroots;
  DR_STAGE_BEGIN();
  if (kPair) real_roots_half<10>(cs, half != 0, roots, nroots);
  else real_roots<10>(cs, roots, nroots);
  DR_STAGE(3);
  if (!ok) nroots = 0;

  int slot = 0;
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    if (!__any(i < nroots)) continue;   // wave-uniform skip; lanes without this root compute and discard
    const bool has_root = i < nroots;
    const double z = roots[i];
    // rows of B(z): (bx(z), by(z), b1(z)) . (x, y, 1) = 0 ; null vector = best-conditioned cross product
    double rx[3], ry[3], r1[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      rx[r] = ((bx[r][3] * z + bx[r][2]) * z + bx[r][1]) * z + bx[r][0];
      ry[r] = ((by[r][3] * z + by[r][2]) * z + by[r][1]) * z + by[r][0];
      r1[r] = (((b1[r][4] * z + b1[r][3]) * z + b1[r][2]) * z + b1[r][1]) * z + b1[r][0];
    }
    double bestn = -1, vx = 0, vy = 0, vw = 1;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = a + 1; b < 3; ++b) {
        const double cx = ry[a] * r1[b] - r1[a] * ry[b];
        const double cy = r1[a] * rx[b] - rx[a] * r1[b];
        const double cw = rx[a] * ry[b] - ry[a] * rx[b];
        const double nn = cw * cw;  // we divide by the w component: pick the largest
        if (nn > bestn) { bestn = nn; vx = cx; vy = cy; vw = cw; }
      }
    const double rw = frcp(vw), x = vx * rw, y = vy * rw;
    const int dst_slot = (kPair && half) ? 9 - slot : slot;
    const bool good = finish_solution<T>(nb, x, y, z, has_root && is_finite(x) && is_finite(y) && slot < 10,
                                         models + 9 * dst_slot, active, models64 ? models64 + 9 * dst_slot : nullptr);
    if (good && active) valid[dst_slot] = 1;
    slot += good ? 1 : 0;
  }
  if (kPair) {
    // the partner lane's count; if numerical duplicates ever made the two halves overlap, the upper half wins
    const int other = __shfl_xor(slot, 1, 64);
    const int lo = half ? other : slot, hi = half ? slot : other;   // slots [0,lo) and (9-hi, 9] are taken
    if (active && half == 0) {
      for (int s = min(lo, 10 - hi); s < 10 - hi; ++s) {
        write_identity<T>(models + 9 * s);
        if (models64) write_identity<double>(models64 + 9 * s);
        valid[s] = 0;
      }
    }
  } else if (active) {
    for (int s = slot; s < 10; ++s) {
      write_identity<T>(models + 9 * s);
      if (models64) write_identity<double>(models64 + 9 * s);
      valid[s] = 0;
    }
  }
}

// LDS doubles per 32-sample block of the two-lanes-per-sample Nister kernel: basis (36 x 32) + B(z) (39 x 32; the
// candidate queue overlays it later), then the root-search workspace
constexpr int kPairFinishDoubles = (36 + 39) * 32;
constexpr int kmax(int a, int b) { return a > b ? a : b; }
constexpr int kNisterPairDoubles =
    kmax(100 * 32, kmax(FinishQueue::kDoubles, kPairFinishDoubles + (DR_K3_WAVE_ROOTS ? kmax(RootWs<10>::kDoubles, DR_K3_STURM ? SturmWs<10>::kDoubles : 0) : 0)));

// B(z) (rows k = e - z f, l = g - z h, m = i - z j of the reduced system; columns x: degree 3, y: degree 3, 1: degree 4) as
// 39 doubles bz[13 r + (0..3 | 4..7 | 8..12)] and its determinant's coefficients cs[0..10] (ascending)
__device__ __forceinline__ void nister_bz_det(const double (&X)[6][10], double (&bz)[39], double (&cs)[11]) {
  double bx[3][4], by[3][4], b1[3][5];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const double(&hi)[10] = X[2 * r];
    const double(&lo)[10] = X[2 * r + 1];
    bx[r][0] = hi[2]; bx[r][1] = hi[1] - lo[2]; bx[r][2] = hi[0] - lo[1]; bx[r][3] = -lo[0];
    by[r][0] = hi[5]; by[r][1] = hi[4] - lo[5]; by[r][2] = hi[3] - lo[4]; by[r][3] = -lo[3];
    b1[r][0] = hi[9]; b1[r][1] = hi[8] - lo[9]; b1[r][2] = hi[7] - lo[8]; b1[r][3] = hi[6] - lo[7];
    b1[r][4] = -lo[6];
  }
#pragma unroll
  for (int i = 0; i < 11; ++i) cs[i] = 0;
  auto minor_acc = [&](int a, int b, int r, double sgn) {
    double mn[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) mn[i] = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) mn[i + j] += bx[a][i] * by[b][j] - bx[b][i] * by[a][j];
#pragma unroll
    for (int i = 0; i < 7; ++i)
#pragma unroll
      for (int j = 0; j < 5; ++j) cs[i + j] += sgn * mn[i] * b1[r][j];
  };
  minor_acc(1, 2, 0, 1.0);
  minor_acc(0, 2, 1, -1.0);
  minor_acc(0, 1, 2, 1.0);
#pragma unroll
  for (int r = 0; r < 3; ++r) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      bz[13 * r + i] = bx[r][i];
      bz[13 * r + 4 + i] = by[r][i];
    }
#pragma unroll
    for (int i = 0; i < 5; ++i) bz[13 * r + 8 + i] = b1[r][i];
  }
}

// (x, y) of every root z of this lane: null vector of B(z) by the best-conditioned cross product of two rows
__device__ __forceinline__ void nister_xy_of_roots(const double (&bz)[39], const double (&roots)[10], int nroots, double (&xs)[10],
                                                   double (&ys)[10], unsigned &cand) {
  cand = 0;
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    xs[i] = 0;
    ys[i] = 0;
    if (!__any(i < nroots)) continue;
    const double z = roots[i];
    double rx[3], ry[3], r1[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const double *bx = bz + 13 * r, *by = bz + 13 * r + 4, *b1 = bz + 13 * r + 8;
      rx[r] = ((bx[3] * z + bx[2]) * z + bx[1]) * z + bx[0];
      ry[r] = ((by[3] * z + by[2]) * z + by[1]) * z + by[0];
      r1[r] = (((b1[4] * z + b1[3]) * z + b1[2]) * z + b1[1]) * z + b1[0];
    }
    double bestn = -1, vx = 0, vy = 0, vw = 1;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = a + 1; b < 3; ++b) {
        const double cx = ry[a] * r1[b] - r1[a] * ry[b];
        const double cy = r1[a] * rx[b] - rx[a] * r1[b];
        const double cw = rx[a] * ry[b] - ry[a] * rx[b];
        const double nn = cw * cw;  // we divide by the w component: pick the largest
        if (nn > bestn) { bestn = nn; vx = cx; vy = cy; vw = cw; }
      }
    const double rw = frcp(vw), x = vx * rw, y = vy * rw;
    xs[i] = x;
    ys[i] = y;
    if (is_finite(x) && is_finite(y)) cand |= 1u << i;
  }
}

// Two lanes per sample, balanced final stage (see balanced_finish).  LDS use after the constraint solve: the basis (36
// doubles per sample) for the whole stage; B(z) (39 doubles per sample) only across the root search, where it would
// otherwise occupy 78 registers -- the candidate queue reuses that space afterwards.
// nister_back_pair: everything after det B(z) -- the basis and B(z) of the block's 32 samples wait in LDS (FinishQueue layout:
// basis at lds[(9 t + q) * 32 + j], B(z) element k at lds[36 * 32 + k * 32 + j]), cs = the lane's degree-10 polynomial
// (1 + z^10 for a lane pair without a usable sample: no real root in either half of the search, no bracket in the wave's queues)
template <typename T>
__device__ __forceinline__ void nister_back_pair(const double (&cs)[11], double *lds, int lane, size_t s0, bool active,
                                                 T *__restrict__ models, uint8_t *__restrict__ valid, double *__restrict__ models64) {
  const FinishQueue fq(lds);
  const int half = lane & 1;
  const double *bzl = fq.u + (lane >> 1);   // B(z): element k of sample j at bzl[k * 32] (the queue's space, later)
  double roots[10];
  int nroots;
  DR_STAGE_BEGIN();
#if DR_K3_WAVE_ROOTS
#if DR_K3_STURM
  real_roots_half_sturm<10>(cs, half != 0, roots, nroots, lds + kPairFinishDoubles, lane);
#else
  real_roots_half_wave<10>(cs, half != 0, roots, nroots, lds + kPairFinishDoubles, lane);
#endif
#else
  real_roots_half<10>(cs, half != 0, roots, nroots);
#endif
  DR_STAGE(3);
  if (!active) nroots = 0;
  double xs[10], ys[10];
  unsigned cand;
  {
    double bz[39];
#pragma unroll
    for (int k = 0; k < 39; ++k) bz[k] = bzl[k * 32];
    nister_xy_of_roots(bz, roots, nroots, xs, ys, cand);
  }
  DR_STAGE(4);
#ifdef DR_K3_SKIP_FINAL   // timing experiment: everything but the final stage (the results are kept alive by one store)
  if (active && lane == 0) models[s0 * 90] = (T)(xs[0] + ys[1] + (double)cand);
  return;
#endif
  balanced_finish<T>(fq, lane, nroots, xs, ys, roots, cand, s0, active, models, valid, models64);
  DR_STAGE(11);   // the balanced final stage alone
}

template <typename T>
__device__ __forceinline__ void nister_finish_pair(const double (&nb)[4][9], const double (&X)[6][10], bool ok, double *lds, int lane,
                                                   size_t s0, bool active, T *__restrict__ models, uint8_t *__restrict__ valid,
                                                   double *__restrict__ models64) {
  const FinishQueue fq(lds);
  park_basis(fq, nb, lane);
  double cs[11];
  double *bzl = fq.u + (lane >> 1);
  {
    double bz[39];
    nister_bz_det(X, bz, cs);
    if ((lane & 1) == 0) {
#pragma unroll
      for (int k = 0; k < 39; ++k) bzl[k * 32] = bz[k];
    }
  }
  if (!ok || !active) {   // a lane pair without a sample (partial block, or fewer samples per block) or with a rank-deficient
                          // system: 1 + z^10 -- no real root in either half of the search, hence no bracket in the wave's task queues
#pragma unroll
    for (int i = 0; i <= 10; ++i) cs[i] = (i == 0 || i == 10) ? 1.0 : 0.0;
  }
  nister_back_pair<T>(cs, lds, lane, s0, active, models, valid, models64);
}

// ---- Stewenius: characteristic polynomial of the action matrix, eigenvectors of its real eigenvalues ----------------------
// Action matrix M (10x10): rows 0-5 <- reduced rows 0,1,2,4,5,7 (g); M[6][0] = M[7][1] = M[8][3] = M[9][6] = -1
// (stewenius.py:64-72).  Householder-Hessenberg + La Budde's recurrence, fully unrolled in registers; cs ascending.
__device__ __forceinline__ void stewenius_charpoly(const double (&g)[6][10], double (&cs)[11]) {
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
    const double alpha = -dsign(fsqrt(nrm2), x0);
    const double v0 = x0 - alpha;
    const double vtv = v0 * v0 + (nrm2 - x0 * x0);
    const double beta = vtv > 0 ? 2.0 * frcp(vtv) : 0.0;
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

// (x, y, z) of every eigenvalue of this lane: the eigenvector follows from rows 0-5 of (M - lambda I) v = 0 with the structural
// rows substituted (unknowns y^2, yz, z^2, y, z), solved in the least-squares sense by Householder QR (no pivoting => static
// indexing); v ~ (x^2, xy, y^2, xz, yz, z^2, x, y, z, 1), lambda = -x
__device__ __forceinline__ void stewenius_xyz_of_roots(const double (&g)[6][10], const double (&roots)[10], int nroots,
                                                       double (&xs)[10], double (&ys)[10], double (&zs)[10], unsigned &cand) {
  cand = 0;
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    xs[i] = 0; ys[i] = 0; zs[i] = 0;
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
    bool solvable = true;
#pragma unroll
    for (int c = 0; c < 5; ++c) {
      double nrm2 = 0;
#pragma unroll
      for (int r = c; r < 6; ++r) nrm2 += K[r][c] * K[r][c];
      const double nrm = fsqrt(nrm2);
      const double alpha = -dsign(nrm, K[c][c]);
      const double v0 = K[c][c] - alpha;
      const double vtv = v0 * v0 + (nrm2 - K[c][c] * K[c][c]);
      const double beta = vtv > 0 ? 2.0 * frcp(vtv) : 0.0;
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
      u[c] = fdiv(acc, K[c][c]);
    }
    const double x = -lam, y = u[3], z = u[4];
    xs[i] = x; ys[i] = y; zs[i] = z;
    if (has && solvable && is_finite(y) && is_finite(z)) cand |= 1u << i;
  }
}

}  // namespace dr
