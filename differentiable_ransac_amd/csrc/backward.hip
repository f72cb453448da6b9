// Backward passes of the solvers and of the scoring kernels (train mode: ransac.py:78-108 back-propagates a
// loss on the returned models through solver -> gather -> sampler).
//
// The reference differentiates through torch.linalg.svd of a matrix with a 4-fold zero singular value; its
// five-point gradients are numerically meaningless (f32 vs f64 differ by >100 %, Q12).  Here every solver
// backward is the implicit-function derivative of the constraints that DEFINE the returned model, which is
// basis independent and finite:
//   five-point : x2_k^T E x1_k = 0 (k = 1..5) on the tangent space of the unit-norm essential manifold;
//   8-point    : eigenvector perturbation of A^T A (all eigenpairs by Jacobi) + the Hartley normalisation chain;
//   rigid      : derivative of the orthogonal polar factor of the covariance (Kabsch), plus the t formula.
#include "solver_common.hpp"

namespace dr {

// ---------------------------------------------------------------------------------------------- five-point
__device__ __forceinline__ bool solve_spd5(double (&G)[5][5], double (&b)[5]) {
  // Gaussian elimination with partial pivoting (G is SPD up to rounding; pivoting keeps it safe near rank loss)
#pragma unroll
  for (int c = 0; c < 5; ++c) {
    int piv = c;
    double best = fabs(G[c][c]);
#pragma unroll
    for (int r = c + 1; r < 5; ++r)
      if (r > c && fabs(G[r][c]) > best) { best = fabs(G[r][c]); piv = r; }
    if (!(best > 0)) return false;
#pragma unroll
    for (int r = c + 1; r < 5; ++r) {
      if (r == piv) {
#pragma unroll
        for (int k = 0; k < 5; ++k) { const double t = G[c][k]; G[c][k] = G[r][k]; G[r][k] = t; }
        const double t = b[c]; b[c] = b[r]; b[r] = t;
      }
    }
    const double inv = 1.0 / G[c][c];
#pragma unroll
    for (int r = c + 1; r < 5; ++r) {
      const double f = G[r][c] * inv;
#pragma unroll
      for (int k = c; k < 5; ++k) G[r][k] -= f * G[c][k];
      b[r] -= f * b[c];
    }
  }
#pragma unroll
  for (int c = 4; c >= 0; --c) {
    double acc = b[c];
#pragma unroll
    for (int k = c + 1; k < 5; ++k) acc -= G[c][k] * b[k];
    b[c] = acc / G[c][c];
  }
  return true;
}

// T = type of samples / gradients in memory (f32: the training path; f64: `-pr 2 -tr 1`, round 5 -- the arithmetic is f64 either way)
template <typename MT, typename T = float>
__global__ __launch_bounds__(64) void fivepoint_bwd_kernel(const T *__restrict__ samples,
                                                           const MT *__restrict__ models,
                                                           const uint8_t *__restrict__ valid,
                                                           const T *__restrict__ grad_models, int Bt,
                                                           T *__restrict__ grad_samples,
                                                           const int32_t *__restrict__ which) {
  // which != NULL: the gradient arrives SPARSE -- grad_models is grad_chosen [Bt,9], the gradient of the one slot
  // which[s] K5 picked for sample s (which[s] < 0: none) -- instead of a dense [Bt,10,9] tensor that is zero in nine slots of
  // ten (what dr_select_closest_bwd used to write and this kernel used to scan: 11.8 MB each way per training step)
  const int s = blockIdx.x * 64 + threadIdx.x;
  if (s >= Bt) return;
  double x1[5][3], x2[5][3], gacc[5][4];
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    x1[k][0] = samples[(size_t)s * 20 + 4 * k];
    x1[k][1] = samples[(size_t)s * 20 + 4 * k + 1];
    x1[k][2] = 1.0;
    x2[k][0] = samples[(size_t)s * 20 + 4 * k + 2];
    x2[k][1] = samples[(size_t)s * 20 + 4 * k + 3];
    x2[k][2] = 1.0;
#pragma unroll
    for (int d = 0; d < 4; ++d) gacc[k][d] = 0;
  }
  // Each lane walks ITS OWN list of slots that carry a gradient (valid and non-zero): round r of the loop serves the r-th
  // such slot of every lane at once.  In the training path exactly one slot per sample has a gradient (the model picked
  // by K5), so the body runs once per wave with all lanes busy instead of once per slot index with 1 lane in 4.
  int next = 0;
  while (true) {
    int slot = -1;
    if (which) {
      if (next == 0) {
        const int w = which[s];
        if (w >= 0 && w < 10 && valid[(size_t)s * 10 + w]) {
          double gn = 0.0;
#pragma unroll
          for (int q = 0; q < 9; ++q) gn += fabs((double)grad_models[(size_t)s * 9 + q]);
          if (gn > 0.0) slot = w;
        }
      }
      next = 10;
    } else {
      for (; next < 10 && slot < 0; ++next) {
        if (!valid[(size_t)s * 10 + next]) continue;
        double gn = 0.0;
#pragma unroll
        for (int q = 0; q < 9; ++q) gn += fabs((double)grad_models[((size_t)s * 10 + next) * 9 + q]);
        if (gn > 0.0) slot = next;
      }
    }
    if (!__any(slot >= 0)) break;
    if (slot < 0) continue;
    double E[3][3], g[3][3];
#pragma unroll
    for (int q = 0; q < 9; ++q) {
      E[q / 3][q % 3] = models[((size_t)s * 10 + slot) * 9 + q];
      g[q / 3][q % 3] = grad_models[which ? (size_t)s * 9 + q : ((size_t)s * 10 + slot) * 9 + q];
    }
    // tangent directions J_c (3x3 each): c<3: [e_c]x E ; c>=3: E [e_{c-3}]x
    double J[6][3][3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      J[0][0][j] = 0;        J[0][1][j] = -E[2][j]; J[0][2][j] = E[1][j];
      J[1][0][j] = E[2][j];  J[1][1][j] = 0;        J[1][2][j] = -E[0][j];
      J[2][0][j] = -E[1][j]; J[2][1][j] = E[0][j];  J[2][2][j] = 0;
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      J[3][i][0] = 0;        J[3][i][1] = E[i][2];  J[3][i][2] = -E[i][1];
      J[4][i][0] = -E[i][2]; J[4][i][1] = 0;        J[4][i][2] = E[i][0];
      J[5][i][0] = E[i][1];  J[5][i][1] = -E[i][0]; J[5][i][2] = 0;
    }
    double AJ[5][6], Jg[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      double acc = 0;
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc += J[c][i][j] * g[i][j];
      Jg[c] = acc;
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        double a = 0;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int j = 0; j < 3; ++j) a += x2[k][i] * J[c][i][j] * x1[k][j];
        AJ[k][c] = a;
      }
    }
    double G[5][5], lam[5];
#pragma unroll
    for (int a = 0; a < 5; ++a) {
      double r = 0;
#pragma unroll
      for (int c = 0; c < 6; ++c) r += AJ[a][c] * Jg[c];
      lam[a] = r;
#pragma unroll
      for (int b = 0; b < 5; ++b) {
        double v = 0;
#pragma unroll
        for (int c = 0; c < 6; ++c) v += AJ[a][c] * AJ[b][c];
        G[a][b] = v;
      }
    }
    if (!solve_spd5(G, lam)) continue;
    bool fin = true;
#pragma unroll
    for (int k = 0; k < 5; ++k) fin = fin && is_finite(lam[k]);
    if (!fin) continue;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      // d/dx1 of x2^T E x1 = (E^T x2)[0:2] ; d/dx2 = (E x1)[0:2]
#pragma unroll
      for (int d = 0; d < 2; ++d) {
        const double etx2 = E[0][d] * x2[k][0] + E[1][d] * x2[k][1] + E[2][d] * x2[k][2];
        const double ex1 = E[d][0] * x1[k][0] + E[d][1] * x1[k][1] + E[d][2] * x1[k][2];
        gacc[k][d] -= lam[k] * etx2;
        gacc[k][2 + d] -= lam[k] * ex1;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 5; ++k)
#pragma unroll
    for (int d = 0; d < 4; ++d) grad_samples[(size_t)s * 20 + 4 * k + d] = (T)gacc[k][d];
}

// ---------------------------------------------------------------------------------------------- five-point, n > 5
// Non-minimal samples (ransac.py:82-83: the 8-point Gumbel sampler feeding the five-point estimator; nister.py:64-65 runs the
// minimal code on all rows): the forward takes S = span of the four eigenvectors of M = sum_r w_r^2 rho_r rho_r^T with the
// smallest eigenvalues and returns the unit-norm essential matrices inside S.  A returned model e is therefore DEFINED by
//   q_j(M)^T e = 0  for the five eigenvectors q_j of the complement,  e on the essential manifold, |e| = 1,
// the same shape as the minimal case with the rows x2_k (x) x1_k replaced by the q_j.  Implicit differentiation on the tangent
// space (directions J_c as above) gives lambda = (AJ AJ^T)^-1 AJ Jg with AJ[j][c] = q_j . vec(J_c), and the part of dq_j that
// matters is its component inside S:  dq_j^T e = sum_{i in S} (v_i . e) (v_i^T dM q_j) / (mu_j - mu_i).  Hence
//   dL = <Mbar, dM>,  Mbar = -sum_j p_j q_j^T,  p_j = sum_{i in S} lambda_j (v_i . e) / (mu_j - mu_i) v_i   (summed over slots),
// and rho_bar_r = -w_r^2 sum_j [p_j (q_j . rho_r) + q_j (p_j . rho_r)],  w_bar_r = -2 w_r sum_j (p_j . rho_r)(q_j . rho_r).
// One lane per sample; M and its eigenvectors (cyclic Jacobi) live in LDS (162 doubles per lane), in the forward's entry
// order rho[3a + b] = x1_a x2_b, so a stored model E[i][j] (x2^T E x1) is the vector e[3j + i].
// T = type of samples / weights / gradients in memory (f32: the training path; f64: `-pr 2 -tr 1 -sam 3`, round 6 -- the arithmetic
// is f64 either way)
template <typename MT, typename T = float>
__global__ __launch_bounds__(64) void fivepoint_nm_bwd_kernel(const T *__restrict__ samples, const T *__restrict__ weights,
                                                              const MT *__restrict__ models, const uint8_t *__restrict__ valid,
                                                              const T *__restrict__ grad_models, int Bt, int n,
                                                              T *__restrict__ grad_samples, T *__restrict__ grad_weights) {
  extern __shared__ __align__(16) double lds[];
  const int lane = threadIdx.x;
  const int s = blockIdx.x * 64 + lane;
  const bool active = s < Bt;
  const int sc = active ? s : Bt - 1;
  const T *pts = samples + (size_t)sc * n * 4;
  const T *wts = weights ? weights + (size_t)sc * n : nullptr;
  LaneWs A{lds + lane}, V{lds + lane + 81 * 64};
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
  // the forward's choice of S: four times the smallest remaining eigenvalue (fivepoint_basis_nonminimal)
  unsigned inS = 0;
  int si[4], cj[5];
  for (int t = 0; t < 4; ++t) {
    int best = 0;
    double bv = INFINITY;
    for (int i = 0; i < 9; ++i) {
      const double ev = A[i * 9 + i];
      if (!((inS >> i) & 1u) && ev < bv) { bv = ev; best = i; }
    }
    inS |= 1u << best;
    si[t] = best;
  }
  {
    int k = 0;
    for (int i = 0; i < 9; ++i)
      if (!((inS >> i) & 1u)) {
#pragma unroll
        for (int t = 0; t < 5; ++t)
          if (t == k) cj[t] = i;
        ++k;
      }
  }
  double Pm[5][9];
#pragma unroll
  for (int j = 0; j < 5; ++j)
#pragma unroll
    for (int q = 0; q < 9; ++q) Pm[j][q] = 0;
#pragma unroll 1
  for (int slot = 0; slot < 10; ++slot) {
    bool has = active && valid[(size_t)sc * 10 + slot];
    double E[3][3], g[3][3];
    double gn = 0.0;
#pragma unroll
    for (int q = 0; q < 9; ++q) {
      E[q / 3][q % 3] = (double)models[((size_t)sc * 10 + slot) * 9 + q];
      const double gv = (double)grad_models[((size_t)sc * 10 + slot) * 9 + q];
      g[q / 3][q % 3] = gv;
      gn += fabs(gv);
    }
    has = has && gn > 0.0;
    if (!__any(has)) continue;
    double J[6][3][3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      J[0][0][j] = 0;        J[0][1][j] = -E[2][j]; J[0][2][j] = E[1][j];
      J[1][0][j] = E[2][j];  J[1][1][j] = 0;        J[1][2][j] = -E[0][j];
      J[2][0][j] = -E[1][j]; J[2][1][j] = E[0][j];  J[2][2][j] = 0;
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      J[3][i][0] = 0;        J[3][i][1] = E[i][2];  J[3][i][2] = -E[i][1];
      J[4][i][0] = -E[i][2]; J[4][i][1] = 0;        J[4][i][2] = E[i][0];
      J[5][i][0] = E[i][1];  J[5][i][1] = -E[i][0]; J[5][i][2] = 0;
    }
    double AJ[5][6], Jg[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      double acc = 0;
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc += J[c][i][j] * g[i][j];
      Jg[c] = acc;
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      double qv[9];
#pragma unroll
      for (int q = 0; q < 9; ++q) qv[q] = V[q * 9 + cj[k]];
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        double a = 0;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int j = 0; j < 3; ++j) a += qv[3 * j + i] * J[c][i][j];
        AJ[k][c] = a;
      }
    }
    double G[5][5], lam[5];
#pragma unroll
    for (int a = 0; a < 5; ++a) {
      double r = 0;
#pragma unroll
      for (int c = 0; c < 6; ++c) r += AJ[a][c] * Jg[c];
      lam[a] = r;
#pragma unroll
      for (int b = 0; b < 5; ++b) {
        double v = 0;
#pragma unroll
        for (int c = 0; c < 6; ++c) v += AJ[a][c] * AJ[b][c];
        G[a][b] = v;
      }
    }
    bool fin = solve_spd5(G, lam);
#pragma unroll
    for (int k = 0; k < 5; ++k) fin = fin && is_finite(lam[k]);
    if (!(has && fin)) {
#pragma unroll
      for (int k = 0; k < 5; ++k) lam[k] = 0;
    }
    for (int t = 0; t < 4; ++t) {
      double vi[9], ui = 0;
#pragma unroll
      for (int q = 0; q < 9; ++q) vi[q] = V[q * 9 + si[t]];
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) ui += vi[3 * j + i] * E[i][j];
      const double mui = A[si[t] * 9 + si[t]];
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        const double gap = A[cj[k] * 9 + cj[k]] - mui;
        const double kap = gap > 0 ? lam[k] * ui / gap : 0.0;
#pragma unroll
        for (int q = 0; q < 9; ++q) Pm[k][q] += kap * vi[q];
      }
    }
  }
  if (!active) return;
  for (int r = 0; r < n; ++r) {
    const double x1 = pts[4 * r], y1 = pts[4 * r + 1], x2 = pts[4 * r + 2], y2 = pts[4 * r + 3];
    const double w = wts ? (double)wts[r] : 1.0;
    double rho[9], gr[9];
    epipolar_row_5pt(x1, y1, x2, y2, 1.0, rho);
#pragma unroll
    for (int q = 0; q < 9; ++q) gr[q] = 0;
    double gw = 0;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      double qv[9], a = 0, b = 0;
#pragma unroll
      for (int q = 0; q < 9; ++q) {
        qv[q] = V[q * 9 + cj[k]];
        a += qv[q] * rho[q];
        b += Pm[k][q] * rho[q];
      }
      gw += a * b;
#pragma unroll
      for (int q = 0; q < 9; ++q) gr[q] -= Pm[k][q] * a + qv[q] * b;
    }
    const double w2 = w * w;
    // rho = (x1 x2, x1 y2, x1, y1 x2, y1 y2, y1, x2, y2, 1)
    grad_samples[((size_t)s * n + r) * 4 + 0] = (T)(w2 * (gr[0] * x2 + gr[1] * y2 + gr[2]));
    grad_samples[((size_t)s * n + r) * 4 + 1] = (T)(w2 * (gr[3] * x2 + gr[4] * y2 + gr[5]));
    grad_samples[((size_t)s * n + r) * 4 + 2] = (T)(w2 * (gr[0] * x1 + gr[3] * y1 + gr[6]));
    grad_samples[((size_t)s * n + r) * 4 + 3] = (T)(w2 * (gr[1] * x1 + gr[4] * y1 + gr[7]));
    if (grad_weights) grad_weights[(size_t)s * n + r] = (T)(-2.0 * w * gw);
  }
}

// ---------------------------------------------------------------------------------------------- 8-point / LSQ
template <typename T>   // T = type of samples / models / gradients in memory (f64: `-pr 2 -tr 1`, round 5); f64 arithmetic either way
__global__ __launch_bounds__(64) void f8_bwd_kernel(const T *__restrict__ samples, const T *__restrict__ weights,
                                                    const T *__restrict__ models,
                                                    const T *__restrict__ grad_models, int Bt, int n,
                                                    T *__restrict__ grad_samples,
                                                    T *__restrict__ grad_weights) {
  extern __shared__ __align__(16) double lds[];
  const int lane = threadIdx.x;
  const int s = blockIdx.x * 64 + lane;
  const bool active = s < Bt;
  const int sc = active ? s : Bt - 1;
  const T *pts = samples + (size_t)sc * n * 4;
  const T *wts = weights ? weights + (size_t)sc * n : nullptr;
  LaneWs A{lds + lane}, V{lds + lane + 81 * 64};
  double mu[4] = {0, 0, 0, 0};
  for (int r = 0; r < n; ++r)
#pragma unroll
    for (int d = 0; d < 4; ++d) mu[d] += (double)pts[4 * r + d];
#pragma unroll
  for (int d = 0; d < 4; ++d) mu[d] /= (double)n;
  double d1 = 0, d2 = 0;
  for (int r = 0; r < n; ++r) {
    const double a = (double)pts[4 * r] - mu[0], b = (double)pts[4 * r + 1] - mu[1];
    const double c = (double)pts[4 * r + 2] - mu[2], d = (double)pts[4 * r + 3] - mu[3];
    d1 += sqrt(a * a + b * b);
    d2 += sqrt(c * c + d * d);
  }
  const double r1 = M_SQRT2 * (double)n / d1, r2 = M_SQRT2 * (double)n / d2;
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
  int imin = 0;
  double lmin = INFINITY;
  for (int i = 0; i < 9; ++i) {
    const double ev = A[i * 9 + i];
    if (ev < lmin) { lmin = ev; imin = i; }
  }
  double f[9];
#pragma unroll
  for (int q = 0; q < 9; ++q) f[q] = V[q * 9 + imin];
  // the forward's sign: align f with the stored model (F = T2^T Fhat T1 is linear in Fhat)
  double gF[9], Fm[9];
#pragma unroll
  for (int q = 0; q < 9; ++q) {
    gF[q] = grad_models[(size_t)sc * 9 + q];
    Fm[q] = models[(size_t)sc * 9 + q];
  }
  // T1 = [[r1,0,-r1 mu0],[0,r1,-r1 mu1],[0,0,1]] ; T2^T = [[r2,0,0],[0,r2,0],[-r2 mu2,-r2 mu3,1]]
  const double T1[9] = {r1, 0, -r1 * mu[0], 0, r1, -r1 * mu[1], 0, 0, 1};
  const double T2t[9] = {r2, 0, 0, 0, r2, 0, -r2 * mu[2], -r2 * mu[3], 1};
  auto mm = [](const double (&a)[9], const double (&b)[9], double (&o)[9], bool ta, bool tb) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        double acc = 0;
#pragma unroll
        for (int k = 0; k < 3; ++k) acc += (ta ? a[3 * k + i] : a[3 * i + k]) * (tb ? b[3 * j + k] : b[3 * k + j]);
        o[3 * i + j] = acc;
      }
  };
  double tmp[9], Fchk[9];
  mm(T2t, f, tmp, false, false);
  mm(tmp, T1, Fchk, false, false);
  double dotsign = 0;
#pragma unroll
  for (int q = 0; q < 9; ++q) dotsign += Fchk[q] * Fm[q];
  if (dotsign < 0) {
#pragma unroll
    for (int q = 0; q < 9; ++q) f[q] = -f[q];
  }
  // grad Fhat = T2 gF T1^T ; grad T1 = (T2^T Fhat)^T gF ; grad T2^T = gF (Fhat T1)^T
  double gf[9], gT1[9], gT2t[9];
  mm(T2t, gF, tmp, true, false);   // T2 gF  (T2 = (T2^T)^T)
  mm(tmp, T1, gf, false, true);    // ... T1^T
  mm(T2t, f, tmp, false, false);   // T2^T Fhat
  mm(tmp, gF, gT1, true, false);
  mm(f, T1, tmp, false, false);    // Fhat T1
  mm(gF, tmp, gT2t, false, true);
  double g_r1 = gT1[0] + gT1[4] - mu[0] * gT1[2] - mu[1] * gT1[5];
  double g_r2 = gT2t[0] + gT2t[4] - mu[2] * gT2t[6] - mu[3] * gT2t[7];
  double g_mu[4] = {-r1 * gT1[2], -r1 * gT1[5], -r2 * gT2t[6], -r2 * gT2t[7]};
  // eigenvector perturbation: u = sum_{i != min} v_i (v_i . g_f) / (lambda_i - lambda_min)
  double u[9];
#pragma unroll
  for (int q = 0; q < 9; ++q) u[q] = 0;
  for (int i = 0; i < 9; ++i) {
    if (i == imin) continue;
    double dot = 0;
#pragma unroll
    for (int q = 0; q < 9; ++q) dot += V[q * 9 + i] * gf[q];
    const double gap = A[i * 9 + i] - lmin;
    const double c = gap > 0 ? dot / gap : 0.0;
#pragma unroll
    for (int q = 0; q < 9; ++q) u[q] += c * V[q * 9 + i];
  }
  // pass A over the rows: grad row_r = -(row_r.u) f - (row_r.f) u
  double S_mu[4] = {0, 0, 0, 0}, S_r1 = 0, S_r2 = 0;
  T *gs = grad_samples + (size_t)sc * n * 4;
  for (int r = 0; r < n; ++r) {
    const double a = (double)pts[4 * r] - mu[0], b = (double)pts[4 * r + 1] - mu[1];
    const double c = (double)pts[4 * r + 2] - mu[2], d = (double)pts[4 * r + 3] - mu[3];
    const double X1 = a * r1, Y1 = b * r1, X2 = c * r2, Y2 = d * r2;
    const double w = wts ? (double)wts[r] : 1.0;
    double rho[9];
    epipolar_row_f(X1, Y1, X2, Y2, 1.0, rho);
    double ru = 0, rf = 0;
#pragma unroll
    for (int q = 0; q < 9; ++q) { ru += rho[q] * u[q]; rf += rho[q] * f[q]; }
    ru *= w; rf *= w;
    double grow[9], gw = 0;
#pragma unroll
    for (int q = 0; q < 9; ++q) {
      grow[q] = -(ru * f[q] + rf * u[q]);
      gw += grow[q] * rho[q];
      grow[q] *= w;  // now d L / d rho
    }
    const double gX1 = grow[0] * X2 + grow[3] * Y2 + grow[6];
    const double gY1 = grow[1] * X2 + grow[4] * Y2 + grow[7];
    const double gX2 = grow[0] * X1 + grow[1] * Y1 + grow[2];
    const double gY2 = grow[3] * X1 + grow[4] * Y1 + grow[5];
    if (active) {
      gs[4 * r] = (T)(gX1 * r1); gs[4 * r + 1] = (T)(gY1 * r1);
      gs[4 * r + 2] = (T)(gX2 * r2); gs[4 * r + 3] = (T)(gY2 * r2);
      if (grad_weights) grad_weights[(size_t)sc * n + r] = (T)gw;
    }
    S_mu[0] += gX1 * r1; S_mu[1] += gY1 * r1; S_mu[2] += gX2 * r2; S_mu[3] += gY2 * r2;
    S_r1 += gX1 * a + gY1 * b;
    S_r2 += gX2 * c + gY2 * d;
  }
  g_r1 += S_r1;
  g_r2 += S_r2;
  const double g_d1 = -g_r1 * r1 / d1, g_d2 = -g_r2 * r2 / d2;
  // centroid: direct (-sum of per-point grads) + through T + through the mean distances
  double dsum[4] = {0, 0, 0, 0};
  for (int r = 0; r < n; ++r) {
    const double a = (double)pts[4 * r] - mu[0], b = (double)pts[4 * r + 1] - mu[1];
    const double c = (double)pts[4 * r + 2] - mu[2], d = (double)pts[4 * r + 3] - mu[3];
    const double n1 = sqrt(a * a + b * b), n2 = sqrt(c * c + d * d);
    dsum[0] += a / n1; dsum[1] += b / n1; dsum[2] += c / n2; dsum[3] += d / n2;
  }
  double gm[4];
  gm[0] = g_mu[0] - S_mu[0] - g_d1 * dsum[0];
  gm[1] = g_mu[1] - S_mu[1] - g_d1 * dsum[1];
  gm[2] = g_mu[2] - S_mu[2] - g_d2 * dsum[2];
  gm[3] = g_mu[3] - S_mu[3] - g_d2 * dsum[3];
  if (active) {
    for (int r = 0; r < n; ++r) {
      const double a = (double)pts[4 * r] - mu[0], b = (double)pts[4 * r + 1] - mu[1];
      const double c = (double)pts[4 * r + 2] - mu[2], d = (double)pts[4 * r + 3] - mu[3];
      const double n1 = sqrt(a * a + b * b), n2 = sqrt(c * c + d * d);
      gs[4 * r] += (T)(g_d1 * a / n1 + gm[0] / n);
      gs[4 * r + 1] += (T)(g_d1 * b / n1 + gm[1] / n);
      gs[4 * r + 2] += (T)(g_d2 * c / n2 + gm[2] / n);
      gs[4 * r + 3] += (T)(g_d2 * d / n2 + gm[3] / n);
    }
  }
}

// ---------------------------------------------------------------------------------------------- rigid
__global__ __launch_bounds__(64) void rigid_bwd_kernel(const float *__restrict__ samples,
                                                       const float *__restrict__ models,
                                                       const float *__restrict__ grad_models, int Bt, int n, int flag,
                                                       float *__restrict__ grad_samples) {
  const int s = blockIdx.x * 64 + threadIdx.x;
  if (s >= Bt) return;
  const float *pts = samples + (size_t)s * n * 6;
  float *gs = grad_samples + (size_t)s * n * 6;
  double R[3][3], gR[3][3], gt[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      R[i][j] = models[(size_t)s * 16 + 4 * i + j];
      gR[i][j] = grad_models[(size_t)s * 16 + 4 * i + j];
    }
    gt[i] = grad_models[(size_t)s * 16 + 4 * i + 3];
  }
  double c[6] = {0, 0, 0, 0, 0, 0};
  for (int r = 0; r < n; ++r)
#pragma unroll
    for (int d = 0; d < 6; ++d) c[d] += (double)pts[6 * r + d];
#pragma unroll
  for (int d = 0; d < 6; ++d) c[d] /= (double)n;
  // t_j = c1_j - c0_j * colsum_j(R)
  double g_c0[3], g_c1[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const double cs = R[0][j] + R[1][j] + R[2][j];
    g_c1[j] = gt[j];
    g_c0[j] = -gt[j] * cs;
#pragma unroll
    for (int i = 0; i < 3; ++i) gR[i][j] -= gt[j] * c[j];
  }
  double Gc[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};  // d L / d (sum_r dp dq^T), scale folded in
  if (!flag) {
    // cov = sc * sum dp dq^T = R Y (Y symmetric); G_cov = 2 R Z, Z~_ij = skew(R^T G_R)~_ij / (s_i + s_j) in Y's eigenbasis
    double cov[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    double a0 = 0, a1 = 0;
    for (int r = 0; r < n; ++r) {
      double dp[3], dq[3];
#pragma unroll
      for (int d = 0; d < 3; ++d) { dp[d] = (double)pts[6 * r + d] - c[d]; dq[d] = (double)pts[6 * r + 3 + d] - c[3 + d]; }
      a0 += sqrt(dp[0] * dp[0] + dp[1] * dp[1] + dp[2] * dp[2]);
      a1 += sqrt(dq[0] * dq[0] + dq[1] * dq[1] + dq[2] * dq[2]);
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) cov[i][j] += dp[i] * dq[j];
    }
    const double sc = 3.0 * (double)n * (double)n / (a0 * a1);
    double Y[3][3], Ue[3][3], sv[3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        double acc = 0;
#pragma unroll
        for (int k = 0; k < 3; ++k) acc += R[k][i] * cov[k][j];
        Y[i][j] = acc * sc;
      }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = i + 1; j < 3; ++j) { const double m = 0.5 * (Y[i][j] + Y[j][i]); Y[i][j] = m; Y[j][i] = m; }
    jacobi_eig3(Y, Ue, sv);
    double W[3][3], Wt[3][3], Zt[3][3], Z[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        double acc = 0;
#pragma unroll
        for (int k = 0; k < 3; ++k) acc += R[k][i] * gR[k][j];
        W[i][j] = acc;
      }
    // W~ = U^T skew(W) U
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        double acc = 0;
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
          for (int b = 0; b < 3; ++b) acc += Ue[a][i] * 0.5 * (W[a][b] - W[b][a]) * Ue[b][j];
        Wt[i][j] = acc;
      }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const double den = sv[i] + sv[j];
        Zt[i][j] = (i != j && fabs(den) > 1e-12 * (fabs(sv[0]) + fabs(sv[1]) + fabs(sv[2]))) ? Wt[i][j] / den : 0.0;
      }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        double acc = 0;
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
          for (int b = 0; b < 3; ++b) acc += Ue[i][a] * Zt[a][b] * Ue[j][b];
        Z[i][j] = acc;
      }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        double acc = 0;
#pragma unroll
        for (int k = 0; k < 3; ++k) acc += R[i][k] * Z[k][j];
        Gc[i][j] = 2.0 * acc * sc;
      }
  }
  for (int r = 0; r < n; ++r) {
    double dp[3], dq[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) { dp[d] = (double)pts[6 * r + d] - c[d]; dq[d] = (double)pts[6 * r + 3 + d] - c[3 + d]; }
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      // sum_r dp = sum_r dq = 0, so the centroid receives nothing from the covariance term
      const double gp = Gc[d][0] * dq[0] + Gc[d][1] * dq[1] + Gc[d][2] * dq[2];
      const double gq = Gc[0][d] * dp[0] + Gc[1][d] * dp[1] + Gc[2][d] * dp[2];
      gs[6 * r + d] = (float)(gp + g_c0[d] / n);
      gs[6 * r + 3 + d] = (float)(gq + g_c1[d] / n);
    }
  }
}

// ---------------------------------------------------------------------------------------------- MSAC backward
constexpr int kBT = 256, kBP = 8, kBChunk = kBT * kBP, kBM = 16;

__global__ __launch_bounds__(kBT) void msac_bwd_kernel(const float *__restrict__ matches,
                                                       const float *__restrict__ models, const float *__restrict__ thr,
                                                       const float *__restrict__ grad_scores, int M, int N,
                                                       float *__restrict__ grad_models) {
  __shared__ float part[kBT / 64][kBM][9];
  const int p = blockIdx.z, m0 = blockIdx.x * kBM;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int mcount = min(kBM, M - m0);
  const float t = 1.5f * thr[p];
  const float inv_thr2 = 1.0f / (t * t);
  for (int i = tid; i < (kBT / 64) * kBM * 9; i += kBT) (&part[0][0][0])[i] = 0.f;
  __syncthreads();
  const float *mt = matches + (size_t)p * N * 4;
  for (int c0 = 0; c0 < N; c0 += kBChunk) {
    const int n0 = c0 + tid * kBP;
    for (int ml = 0; ml < mcount; ++ml) {
      float m[9];
#pragma unroll
      for (int q = 0; q < 9; ++q) m[q] = models[((size_t)p * M + m0 + ml) * 9 + q];
      float acc[9];
#pragma unroll
      for (int q = 0; q < 9; ++q) acc[q] = 0.f;
      for (int j = 0; j < kBP; ++j) {
        const int nn = n0 + j;
        if (nn >= N) break;
        const float x1 = mt[4 * nn], y1 = mt[4 * nn + 1], x2 = mt[4 * nn + 2], y2 = mt[4 * nn + 3];
        const float a0 = x2 * m[0] + y2 * m[3] + m[6], a1 = x2 * m[1] + y2 * m[4] + m[7], a2 = x2 * m[2] + y2 * m[5] + m[8];
        const float b0 = x1 * m[0] + y1 * m[1] + m[2], b1 = x1 * m[3] + y1 * m[4] + m[5];
        const float r = x1 * a0 + y1 * a1 + a2;
        const float jj = a0 * a0 + a1 * a1 + b0 * b0 + b1 * b1;
        const float d2 = r * r / jj;
        if (!(d2 * inv_thr2 < 1.0f)) continue;
        const float c1 = 2.f * r / jj, c2 = 2.f * r * r / (jj * jj);
        const float X2[3] = {x2, y2, 1.f}, X1[3] = {x1, y1, 1.f}, a[3] = {a0, a1, 0.f}, b[3] = {b0, b1, 0.f};
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int jx = 0; jx < 3; ++jx)
            acc[3 * i + jx] += c1 * X2[i] * X1[jx] - c2 * (a[jx] * X2[i] + b[i] * X1[jx]);
      }
#pragma unroll
      for (int q = 0; q < 9; ++q) {
        const float v = wave_sum_lane63(acc[q]);
        if (lane == 63) part[wv][ml][q] += v;
      }
    }
  }
  __syncthreads();
  for (int i = tid; i < mcount * 9; i += kBT) {
    const int ml = i / 9, q = i % 9;
    const float v = part[0][ml][q] + part[1][ml][q] + part[2][ml][q] + part[3][ml][q];
    grad_models[((size_t)p * M + m0 + ml) * 9 + q] = -inv_thr2 * v * grad_scores[(size_t)p * M + m0 + ml];
  }
}

__global__ __launch_bounds__(kBT) void rigid_residual_bwd_kernel(const float *__restrict__ pts,
                                                                 const float *__restrict__ models,
                                                                 const float *__restrict__ grad_res, int M, int N,
                                                                 float *__restrict__ grad_models) {
  __shared__ float part[kBT / 64][kBM][12];
  const int p = blockIdx.z, m0 = blockIdx.x * kBM;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int mcount = min(kBM, M - m0);
  for (int i = tid; i < (kBT / 64) * kBM * 12; i += kBT) (&part[0][0][0])[i] = 0.f;
  __syncthreads();
  const float *pt = pts + (size_t)p * N * 6;
  for (int c0 = 0; c0 < N; c0 += kBChunk) {
    const int n0 = c0 + tid * kBP;
    for (int ml = 0; ml < mcount; ++ml) {
      float m[12];
#pragma unroll
      for (int q = 0; q < 12; ++q) m[q] = models[((size_t)p * M + m0 + ml) * 16 + q];
      float acc[12];
#pragma unroll
      for (int q = 0; q < 12; ++q) acc[q] = 0.f;
      for (int j = 0; j < kBP; ++j) {
        const int nn = n0 + j;
        if (nn >= N) break;
        const float *x = pt + (size_t)nn * 6;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const float e = x[3 + i] - (m[4 * i] * x[0] + m[4 * i + 1] * x[1] + m[4 * i + 2] * x[2] + m[4 * i + 3]);
          acc[4 * i] -= 2.f * e * x[0];
          acc[4 * i + 1] -= 2.f * e * x[1];
          acc[4 * i + 2] -= 2.f * e * x[2];
          acc[4 * i + 3] -= 2.f * e;
        }
      }
#pragma unroll
      for (int q = 0; q < 12; ++q) {
        const float v = wave_sum_lane63(acc[q]);
        if (lane == 63) part[wv][ml][q] += v;
      }
    }
  }
  __syncthreads();
  for (int i = tid; i < mcount * 16; i += kBT) {
    const int ml = i / 16, q = i % 16;
    float v = 0.f;
    if (q < 12) v = (part[0][ml][q] + part[1][ml][q] + part[2][ml][q] + part[3][ml][q]) * grad_res[(size_t)p * M + m0 + ml];
    grad_models[((size_t)p * M + m0 + ml) * 16 + q] = v;
  }
}

}  // namespace dr

template <typename T>
static int f8_bwd_launch(const T *samples, const T *weights, const T *models, const T *grad_models, int Bt, int n, T *grad_samples,
                         T *grad_weights, void *stream) {
  const size_t smem = sizeof(double) * 162 * 64;
  // the attribute is per device: remember which devices of this process have it (one process may drive several GPUs)
  static bool attr_set[64] = {false};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 64 || !attr_set[dev]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&dr::f8_bwd_kernel<T>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (dev >= 0 && dev < 64) attr_set[dev] = true;
  }
  hipLaunchKernelGGL((dr::f8_bwd_kernel<T>), dim3((Bt + 63) / 64), dim3(64), smem, (hipStream_t)stream, samples, weights,
                     models, grad_models, Bt, n, grad_samples, grad_weights);
  return dr::check_launch("f8_bwd_kernel");
}

extern "C" {

int dr_solve_nister5_bwd_f32(const float *samples, const float *models, const double *models_f64,
                             const uint8_t *valid, const float *grad_models, int Bt, float *grad_samples,
                             void *stream) {
  DR_REQUIRE(samples && (models || models_f64) && valid && grad_models && grad_samples, "null pointer");
  DR_REQUIRE(Bt > 0, "need Bt > 0");
  if (models_f64)
    hipLaunchKernelGGL((dr::fivepoint_bwd_kernel<double>), dim3((Bt + 63) / 64), dim3(64), 0, (hipStream_t)stream,
                       samples, models_f64, valid, grad_models, Bt, grad_samples, (const int32_t *)nullptr);
  else
    hipLaunchKernelGGL((dr::fivepoint_bwd_kernel<float>), dim3((Bt + 63) / 64), dim3(64), 0, (hipStream_t)stream,
                       samples, models, valid, grad_models, Bt, grad_samples, (const int32_t *)nullptr);
  return dr::check_launch("fivepoint_bwd_kernel");
}

int dr_solve_nister5_bwd_sel_f32(const float *samples, const float *models, const double *models_f64,
                                 const uint8_t *valid, const float *grad_chosen, const int32_t *which, int Bt,
                                 float *grad_samples, void *stream) {
  DR_REQUIRE(samples && (models || models_f64) && valid && grad_chosen && which && grad_samples, "null pointer");
  DR_REQUIRE(Bt > 0, "need Bt > 0");
  if (models_f64)
    hipLaunchKernelGGL((dr::fivepoint_bwd_kernel<double>), dim3((Bt + 63) / 64), dim3(64), 0, (hipStream_t)stream,
                       samples, models_f64, valid, grad_chosen, Bt, grad_samples, which);
  else
    hipLaunchKernelGGL((dr::fivepoint_bwd_kernel<float>), dim3((Bt + 63) / 64), dim3(64), 0, (hipStream_t)stream,
                       samples, models, valid, grad_chosen, Bt, grad_samples, which);
  return dr::check_launch("fivepoint_bwd_kernel");
}

int dr_solve_nister5_nm_bwd_f32(const float *samples, const float *weights, const float *models, const double *models_f64,
                                const uint8_t *valid, const float *grad_models, int Bt, int n, float *grad_samples,
                                float *grad_weights, void *stream) {
  DR_REQUIRE(samples && (models || models_f64) && valid && grad_models && grad_samples, "null pointer");
  DR_REQUIRE(Bt > 0 && n > 5, "need Bt > 0 and n > 5 points per sample (minimal samples: dr_solve_nister5_bwd_f32)");
  const size_t smem = sizeof(double) * 162 * 64;
  static bool attr_set[64] = {false};   // per device
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 64 || !attr_set[dev]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&dr::fivepoint_nm_bwd_kernel<double>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&dr::fivepoint_nm_bwd_kernel<float>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (dev >= 0 && dev < 64) attr_set[dev] = true;
  }
  if (models_f64)
    hipLaunchKernelGGL((dr::fivepoint_nm_bwd_kernel<double>), dim3((Bt + 63) / 64), dim3(64), smem, (hipStream_t)stream,
                       samples, weights, models_f64, valid, grad_models, Bt, n, grad_samples, grad_weights);
  else
    hipLaunchKernelGGL((dr::fivepoint_nm_bwd_kernel<float>), dim3((Bt + 63) / 64), dim3(64), smem, (hipStream_t)stream,
                       samples, weights, models, valid, grad_models, Bt, n, grad_samples, grad_weights);
  return dr::check_launch("fivepoint_nm_bwd_kernel");
}

/* the same with everything f64 in memory (`-sam 3 -fmat 0 -tr 1 -pr 2`; round 6: the f32-I/O kernel rounded samples and gradients) */
int dr_solve_nister5_nm_bwd_f64(const double *samples, const double *weights, const double *models, const uint8_t *valid,
                                const double *grad_models, int Bt, int n, double *grad_samples, double *grad_weights, void *stream) {
  DR_REQUIRE(samples && models && valid && grad_models && grad_samples, "null pointer");
  DR_REQUIRE(Bt > 0 && n > 5, "need Bt > 0 and n > 5 points per sample (minimal samples: dr_solve_nister5_bwd_f64)");
  const size_t smem = sizeof(double) * 162 * 64;
  static bool attr_set[64] = {false};   // per device
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 64 || !attr_set[dev]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&dr::fivepoint_nm_bwd_kernel<double, double>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (dev >= 0 && dev < 64) attr_set[dev] = true;
  }
  hipLaunchKernelGGL((dr::fivepoint_nm_bwd_kernel<double, double>), dim3((Bt + 63) / 64), dim3(64), smem, (hipStream_t)stream, samples,
                     weights, models, valid, grad_models, Bt, n, grad_samples, grad_weights);
  return dr::check_launch("fivepoint_nm_bwd_kernel");
}

int dr_solve_f8_bwd_f32(const float *samples, const float *weights, const float *models, const float *grad_models,
                        int Bt, int n, float *grad_samples, float *grad_weights, void *stream) {
  DR_REQUIRE(samples && models && grad_models && grad_samples, "null pointer");
  DR_REQUIRE(Bt > 0 && n >= 8, "need Bt > 0 and n >= 8");
  return f8_bwd_launch<float>(samples, weights, models, grad_models, Bt, n, grad_samples, grad_weights, stream);
}
/* f64 in memory as well (`-pr 2 -tr 1`, model_cl.py:164-169): the same kernel, nothing rounded to f32 on the way */
int dr_solve_f8_bwd_f64(const double *samples, const double *weights, const double *models, const double *grad_models,
                        int Bt, int n, double *grad_samples, double *grad_weights, void *stream) {
  DR_REQUIRE(samples && models && grad_models && grad_samples, "null pointer");
  DR_REQUIRE(Bt > 0 && n >= 8, "need Bt > 0 and n >= 8");
  return f8_bwd_launch<double>(samples, weights, models, grad_models, Bt, n, grad_samples, grad_weights, stream);
}
/* minimal five-point backward, everything f64 in memory (dense gradient [Bt,10,9]) */
int dr_solve_nister5_bwd_f64(const double *samples, const double *models, const uint8_t *valid, const double *grad_models, int Bt,
                             double *grad_samples, void *stream) {
  DR_REQUIRE(samples && models && valid && grad_models && grad_samples, "null pointer");
  DR_REQUIRE(Bt > 0, "need Bt > 0");
  hipLaunchKernelGGL((dr::fivepoint_bwd_kernel<double, double>), dim3((Bt + 63) / 64), dim3(64), 0, (hipStream_t)stream, samples,
                     models, valid, grad_models, Bt, grad_samples, (const int32_t *)nullptr);
  return dr::check_launch("fivepoint_bwd_kernel");
}

int dr_solve_rigid_bwd_f32(const float *samples, const float *models, const float *grad_models, int Bt, int n,
                           int flag, float *grad_samples, void *stream) {
  DR_REQUIRE(samples && models && grad_models && grad_samples, "null pointer");
  DR_REQUIRE(Bt > 0 && n >= 3, "need Bt > 0 and n >= 3");
  hipLaunchKernelGGL(dr::rigid_bwd_kernel, dim3((Bt + 63) / 64), dim3(64), 0, (hipStream_t)stream, samples, models,
                     grad_models, Bt, n, flag, grad_samples);
  return dr::check_launch("rigid_bwd_kernel");
}

int dr_msac_score_bwd_f32(const float *matches, const float *models, const float *thr, const float *grad_scores,
                          int P, int M, int N, float *grad_models, void *stream) {
  DR_REQUIRE(matches && models && thr && grad_scores && grad_models, "null pointer");
  DR_REQUIRE(P > 0 && M > 0 && N > 0 && P <= 65535, "bad sizes");
  hipLaunchKernelGGL(dr::msac_bwd_kernel, dim3((M + dr::kBM - 1) / dr::kBM, 1, P), dim3(dr::kBT), 0,
                     (hipStream_t)stream, matches, models, thr, grad_scores, M, N, grad_models);
  return dr::check_launch("msac_bwd_kernel");
}

int dr_rigid_residual_bwd_f32(const float *pts, const float *models, const float *grad_res, int P, int M, int N,
                              float *grad_models, void *stream) {
  DR_REQUIRE(pts && models && grad_res && grad_models, "null pointer");
  DR_REQUIRE(P > 0 && M > 0 && N > 0 && P <= 65535, "bad sizes");
  hipLaunchKernelGGL(dr::rigid_residual_bwd_kernel, dim3((M + dr::kBM - 1) / dr::kBM, 1, P), dim3(dr::kBT), 0,
                     (hipStream_t)stream, pts, models, grad_res, M, N, grad_models);
  return dr::check_launch("rigid_residual_bwd_kernel");
}

}  // extern "C"
