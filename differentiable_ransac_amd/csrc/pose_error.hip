// Lp -- pose error of essential matrices: the body of PoseLoss.forward_average (loss.py:11-68),
// i.e. eval_essential_matrix(svd=False) (cv_utils.py:503-525) for every model of every pair in one launch:
//   Horn decomposition           new_decompose_E      cv_utils.py:118-161
//   cheirality vote              recoverPose          cv_utils.py:48-80, cheirality_check :177-189
//   rotation / translation error evaluate_R_t_tensor  cv_utils.py:361-380
// The reference does this per model in Python with one cv2.triangulatePoints call per candidate pose (4 per model) on
// the CPU.  Here: the same (model x point) grid as K4 -- a block owns a tile of models of one pair, a lane keeps its
// points in VGPRs -- and per (model, point) only TWO triangulations: the candidates (R, t) and (R, -t) have systems
// that differ by the sign of the last column, so their DLT solutions differ by the sign of the last coordinate.
// Triangulation = OpenCV's DLT (4x4 system per point, right singular vector of the smallest singular value), computed
// in closed form as the smallest eigenvector of A^T A (characteristic quartic + adjugate), in f64 (f64 FMA issues at
// the f32 rate on CDNA4).  OpenCV is absent from the build container: parity for this primitive is pinned against the
// oracle's SVD restatement only (votes identical on the fixtures); everything around it against the reference.
// Backward: forward-mode differentiation (dual numbers, one pass per entry of E) of Horn + error at the selected
// candidate -- 9 passes of ~200 flops per model; the skew matrix [b]x is a constant, as in the reference (:144-148).
#include "dr_common.hpp"
#include "solver_common.hpp"   // jacobi_eig3

namespace dr {

struct Dual {
  double v, d;
};
__device__ __forceinline__ Dual operator+(Dual a, Dual b) { return {a.v + b.v, a.d + b.d}; }
__device__ __forceinline__ Dual operator-(Dual a, Dual b) { return {a.v - b.v, a.d - b.d}; }
__device__ __forceinline__ Dual operator*(Dual a, Dual b) { return {a.v * b.v, a.d * b.v + a.v * b.d}; }
__device__ __forceinline__ Dual operator/(Dual a, Dual b) {
  const double q = a.v / b.v;
  return {q, (a.d - q * b.d) / b.v};
}
__device__ __forceinline__ Dual operator*(double a, Dual b) { return {a * b.v, a * b.d}; }
__device__ __forceinline__ Dual operator-(Dual a) { return {-a.v, -a.d}; }
__device__ __forceinline__ Dual dsqrt(Dual a) {
  const double s = sqrt(a.v);
  return {s, a.d / (2.0 * s)};
}
__device__ __forceinline__ double dsqrt(double a) { return sqrt(a); }
__device__ __forceinline__ Dual dacos(Dual a) {   // a constant argument (clamped branch) has derivative 0, not 0 * inf
  return {acos(a.v), a.d == 0.0 ? 0.0 : -a.d / sqrt(1.0 - a.v * a.v)};
}
__device__ __forceinline__ double dacos(double a) { return acos(a); }
__device__ __forceinline__ double val(double a) { return a; }
__device__ __forceinline__ double val(Dual a) { return a.v; }
__device__ __forceinline__ double lift(double, double c) { return c; }   // constant of the scalar type of the 1st argument
__device__ __forceinline__ Dual lift(Dual, double c) { return {c, 0.0}; }

template <typename S>
__device__ __forceinline__ void cross3(const S *a, const S *b, S *o) {
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}

// new_decompose_E (cv_utils.py:118-161).  E row-major.  R1, R2 row-major, t unit.
template <typename S>
__device__ __forceinline__ void horn_decompose(const S (&E)[9], S (&R1)[9], S (&R2)[9], S (&t)[3]) {
  S c0[3] = {E[0], E[3], E[6]}, c1[3] = {E[1], E[4], E[7]}, c2[3] = {E[2], E[5], E[8]};   // columns e1, e2, e3
  S x01[3], x12[3], x20[3];
  cross3(c0, c1, x01);
  cross3(c1, c2, x12);
  cross3(c2, c0, x20);
  const double n01 = val(x01[0]) * val(x01[0]) + val(x01[1]) * val(x01[1]) + val(x01[2]) * val(x01[2]);
  const double n12 = val(x12[0]) * val(x12[0]) + val(x12[1]) * val(x12[1]) + val(x12[2]) * val(x12[2]);
  const double n20 = val(x20[0]) * val(x20[0]) + val(x20[1]) * val(x20[1]) + val(x20[2]) * val(x20[2]);
  // torch.argmax: first maximum
  S pick[3];
  const int largest = (n01 >= n12 && n01 >= n20) ? 0 : (n12 >= n20 ? 1 : 2);
#pragma unroll
  for (int i = 0; i < 3; ++i) pick[i] = largest == 0 ? x01[i] : (largest == 1 ? x12[i] : x20[i]);
  S tr = E[0] * E[0];
#pragma unroll
  for (int i = 1; i < 9; ++i) tr = tr + E[i] * E[i];
  const S scale = dsqrt(0.5 * tr);
  const S pn = dsqrt(pick[0] * pick[0] + pick[1] * pick[1] + pick[2] * pick[2]);
  S b[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) b[i] = scale * pick[i] / pn;
  const S bb = b[0] * b[0] + b[1] * b[1] + b[2] * b[2];
  const S bn = dsqrt(bb);
#pragma unroll
  for (int i = 0; i < 3; ++i) t[i] = b[i] / bn;
  // cofactor matrix: rows = cross products of the other two rows (inv(E).T * det in the reference, :163-175)
  S r0[3] = {E[0], E[1], E[2]}, r1[3] = {E[3], E[4], E[5]}, r2[3] = {E[6], E[7], E[8]};
  S cof[9];
  cross3(r1, r2, cof + 0);
  cross3(r2, r0, cof + 3);
  cross3(r0, r1, cof + 6);
  // B = [b]x with DETACHED entries (torch.tensor(...) at :144-148 cuts the graph)
  const double b0 = val(b[0]), b1 = val(b[1]), b2 = val(b[2]);
  S BE[9];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    BE[0 + j] = -b2 * E[3 + j] + b1 * E[6 + j];
    BE[3 + j] = b2 * E[0 + j] - b0 * E[6 + j];
    BE[6 + j] = -b1 * E[0 + j] + b0 * E[3 + j];
  }
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    R1[i] = (cof[i] - BE[i]) / bb;
    R2[i] = (cof[i] + BE[i]) / bb;
  }
}

// decompose_E (cv_utils.py:83-116): E = U S V^T, R1 = U_ W V_^T, R2 = U_ W^T V_^T (U_, V_: negated when their determinant is
// negative), t = last column of U.  Whatever signs an SVD routine returns, {R1, R2} is the set {A + C, -A + C} with
// A = u1 v0^T - u0 v1^T, C = u2 v2^T, u2 = u0 x u1, v2 = v0 x v1 (a joint sign flip of a pair (u_k, v_k), a lone flip of
// u2 or v2 when sigma_3 = 0, and the determinant fix-ups all just swap the two), and t = +-u2: the four candidate poses
// are fixed as a set; their ORDER (hence `which`, and the winner of an exact tie of votes) is a LAPACK artefact.
// Here: V from the symmetric Jacobi eigen-decomposition of E^T E, u_k = E v_k / sigma_k.  Forward only.
__device__ __forceinline__ void svd_decompose(const double (&E)[9], double (&R1)[9], double (&R2)[9], double (&t)[3]) {
  double A[3][3], V[3][3], d[3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) A[i][j] = E[0 + i] * E[0 + j] + E[3 + i] * E[3 + j] + E[6 + i] * E[6 + j];
  jacobi_eig3(A, V, d);
  // indices of the two largest eigenvalues (static select network)
  const int i0 = (d[0] >= d[1] && d[0] >= d[2]) ? 0 : (d[1] >= d[2] ? 1 : 2);
  const int ia = i0 == 0 ? 1 : 0, ib = i0 == 2 ? 1 : 2;
  const int i1 = d[ia] >= d[ib] ? ia : ib;
  double v0[3], v1[3], v2[3], u0[3], u1[3], u2[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    v0[k] = i0 == 0 ? V[k][0] : (i0 == 1 ? V[k][1] : V[k][2]);
    v1[k] = i1 == 0 ? V[k][0] : (i1 == 1 ? V[k][1] : V[k][2]);
  }
  cross3(v0, v1, v2);
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    u0[k] = E[3 * k] * v0[0] + E[3 * k + 1] * v0[1] + E[3 * k + 2] * v0[2];
    u1[k] = E[3 * k] * v1[0] + E[3 * k + 1] * v1[1] + E[3 * k + 2] * v1[2];
  }
  const double n0 = 1.0 / sqrt(u0[0] * u0[0] + u0[1] * u0[1] + u0[2] * u0[2]);
#pragma unroll
  for (int k = 0; k < 3; ++k) u0[k] *= n0;
  const double pr = u1[0] * u0[0] + u1[1] * u0[1] + u1[2] * u0[2];   // re-orthogonalise (sigma_1 = sigma_2 for a true E)
#pragma unroll
  for (int k = 0; k < 3; ++k) u1[k] -= pr * u0[k];
  const double n1 = 1.0 / sqrt(u1[0] * u1[0] + u1[1] * u1[1] + u1[2] * u1[2]);
#pragma unroll
  for (int k = 0; k < 3; ++k) u1[k] *= n1;
  cross3(u0, u1, u2);
#pragma unroll
  for (int i = 0; i < 3; ++i) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const double a = u1[i] * v0[j] - u0[i] * v1[j], c = u2[i] * v2[j];
      R1[3 * i + j] = a + c;
      R2[3 * i + j] = -a + c;
    }
    t[i] = u2[i];
  }
}

__device__ __forceinline__ void pose_decompose(const double (&E)[9], double (&R1)[9], double (&R2)[9], double (&t)[3], int svd) {
  if (svd) svd_decompose(E, R1, R2, t);
  else horn_decompose<double>(E, R1, R2, t);
}

// evaluate_R_t_tensor (cv_utils.py:361-380) in degrees (eval_essential_matrix :525); tg = t_gt / (|t_gt| + 1e-8)
template <typename S>
__device__ __forceinline__ void rt_error(const S (&R)[9], const S (&t)[3], const double (&Rg)[9], const double (&tg)[3],
                                         S &err_q, S &err_t) {
  const double eps = 1e-8, deg = 180.0 / 3.14159265358979323846;
  S c = Rg[0] * R[0];
#pragma unroll
  for (int i = 1; i < 9; ++i) c = c + Rg[i] * R[i];
  c = 0.5 * (c - lift(c, 1.0));
  // torch.max(torch.min(c, 1), -1): the clamped branch has zero gradient
  if (val(c) > 1.0) c = lift(c, 1.0);
  if (val(c) < -1.0) c = lift(c, -1.0);
  err_q = deg * dacos(c);
  const S dt = tg[0] * t[0] + tg[1] * t[1] + tg[2] * t[2];
  S loss_t = lift(dt, 1.0) - dt * dt;
  if (val(loss_t) < eps) loss_t = lift(dt, eps);
  err_t = deg * dacos(dsqrt(lift(dt, 1.0 + eps) - loss_t));
}

// DLT triangulation for cameras [I|0] and [R|t], point (x1,y1) <-> (x2,y2): X = eigenvector of the smallest eigenvalue
// of G = A^T A (= the right singular vector OpenCV's SVD returns), any scale / sign.  Closed form instead of an
// iteration on vectors (inverse iteration needs (s3/s4)^2 >> 1, false for outlier matches): coefficients of the
// characteristic quartic from 2x2 / 3x3 minors, its smallest root by Newton from 0 (monotone from below for a
// polynomial with real roots, each step covers >= 1/4 of the distance), then one row of adj(G - lambda I).
// Agrees with the SVD to 1e-15 on the fixtures (scratch/tri_emul.py).
struct Minors2 {
  double m01, m02, m03, m12, m13, m23;
};
__device__ __forceinline__ Minors2 minors2(const double (&r)[4], const double (&s)[4]) {   // 2x2 minors of two rows
  return {r[0] * s[1] - r[1] * s[0], r[0] * s[2] - r[2] * s[0], r[0] * s[3] - r[3] * s[0],
          r[1] * s[2] - r[2] * s[1], r[1] * s[3] - r[3] * s[1], r[2] * s[3] - r[3] * s[2]};
}

__device__ __forceinline__ void triangulate(const double (&R)[9], const double (&t)[3], double x1, double y1, double x2,
                                            double y2, double (&X)[4]) {
  const double a[4] = {x2 * R[6] - R[0], x2 * R[7] - R[1], x2 * R[8] - R[2], x2 * t[2] - t[0]};
  const double c[4] = {y2 * R[6] - R[3], y2 * R[7] - R[4], y2 * R[8] - R[5], y2 * t[2] - t[1]};
  double g0[4], g1[4], g2[4], g3[4];   // rows of G (symmetric: 10 products, mirrored)
  g0[0] = a[0] * a[0] + c[0] * c[0] + 1.0;
  g0[1] = a[0] * a[1] + c[0] * c[1];
  g0[2] = a[0] * a[2] + c[0] * c[2] - x1;
  g0[3] = a[0] * a[3] + c[0] * c[3];
  g1[1] = a[1] * a[1] + c[1] * c[1] + 1.0;
  g1[2] = a[1] * a[2] + c[1] * c[2] - y1;
  g1[3] = a[1] * a[3] + c[1] * c[3];
  g2[2] = a[2] * a[2] + c[2] * c[2] + (x1 * x1 + y1 * y1);
  g2[3] = a[2] * a[3] + c[2] * c[3];
  g3[3] = a[3] * a[3] + c[3] * c[3];
  g1[0] = g0[1]; g2[0] = g0[2]; g2[1] = g1[2]; g3[0] = g0[3]; g3[1] = g1[3]; g3[2] = g2[3];
  const Minors2 p = minors2(g0, g1), q = minors2(g2, g3);
  const double c3 = g0[0] + g1[1] + g2[2] + g3[3];
  const double c2 = (g0[0] * g1[1] - g0[1] * g0[1]) + (g0[0] * g2[2] - g0[2] * g0[2]) + (g0[0] * g3[3] - g0[3] * g0[3]) +
                    (g1[1] * g2[2] - g1[2] * g1[2]) + (g1[1] * g3[3] - g1[3] * g1[3]) + (g2[2] * g3[3] - g2[3] * g2[3]);
  const double c1 = (g1[1] * q.m23 - g1[2] * q.m13 + g1[3] * q.m12) + (g0[0] * q.m23 - g0[2] * q.m03 + g0[3] * q.m02) +
                    (g3[0] * p.m13 - g3[1] * p.m03 + g3[3] * p.m01) + (g2[0] * p.m12 - g2[1] * p.m02 + g2[2] * p.m01);
  const double c0 = p.m01 * q.m23 - p.m02 * q.m13 + p.m03 * q.m12 + p.m12 * q.m03 - p.m13 * q.m02 + p.m23 * q.m01;
  // Newton from below; v_rcp_f64 instead of the IEEE division sequence: an inexact step is corrected by the next one
  // and the final accuracy is that of the last evaluation of the polynomial (10 steps: the distance to the root
  // shrinks by >= 1/4 per step until the quadratic phase, which takes 3 steps from 1e-2 to 1e-16)
  double lam = 0.0;
#pragma unroll 1
  for (int it = 0; it < 10; ++it) {
    const double pv = (((lam - c3) * lam + c2) * lam - c1) * lam + c0;
    const double dp = ((4.0 * lam - 3.0 * c3) * lam + 2.0 * c2) * lam - c1;
    lam -= (dp != 0.0) ? pv * __builtin_amdgcn_rcp(dp) : 0.0;
  }
  g0[0] -= lam; g1[1] -= lam; g2[2] -= lam; g3[3] -= lam;
  const Minors2 u = minors2(g0, g1), w = minors2(g2, g3);
  // cofactors of the symmetric matrix B = G - lambda I (upper triangle)
  const double C00 = g1[1] * w.m23 - g1[2] * w.m13 + g1[3] * w.m12;
  const double C01 = -(g1[0] * w.m23 - g1[2] * w.m03 + g1[3] * w.m02);
  const double C02 = g1[0] * w.m13 - g1[1] * w.m03 + g1[3] * w.m01;
  const double C03 = -(g1[0] * w.m12 - g1[1] * w.m02 + g1[2] * w.m01);
  const double C11 = g0[0] * w.m23 - g0[2] * w.m03 + g0[3] * w.m02;
  const double C12 = -(g0[0] * w.m13 - g0[1] * w.m03 + g0[3] * w.m01);
  const double C13 = g0[0] * w.m12 - g0[1] * w.m02 + g0[2] * w.m01;
  const double C22 = g3[0] * u.m13 - g3[1] * u.m03 + g3[3] * u.m01;
  const double C23 = -(g3[0] * u.m12 - g3[1] * u.m02 + g3[2] * u.m01);
  const double C33 = g2[0] * u.m12 - g2[1] * u.m02 + g2[2] * u.m01;
  // adj(B) = const * v v^T: take the row with the largest diagonal entry
  const double d0 = fabs(C00), d1 = fabs(C11), d2 = fabs(C22), d3 = fabs(C33);
  const int k = (d0 >= d1 && d0 >= d2 && d0 >= d3) ? 0 : ((d1 >= d2 && d1 >= d3) ? 1 : (d2 >= d3 ? 2 : 3));
  X[0] = k == 0 ? C00 : (k == 1 ? C01 : (k == 2 ? C02 : C03));
  X[1] = k == 0 ? C01 : (k == 1 ? C11 : (k == 2 ? C12 : C13));
  X[2] = k == 0 ? C02 : (k == 1 ? C12 : (k == 2 ? C22 : C23));
  X[3] = k == 0 ? C03 : (k == 1 ? C13 : (k == 2 ? C23 : C33));
}

constexpr int kPoseThreads = 256, kPosePts = 8, kPoseTile = 4;

template <typename T>
__global__ __launch_bounds__(kPoseThreads) void pose_error_kernel(const T *__restrict__ matches, const T *__restrict__ models,
                                                                 const T *__restrict__ gt_R, const T *__restrict__ gt_t, int M,
                                                                 int N, double dist_thr, T *__restrict__ err_R,
                                                                 T *__restrict__ err_t, int32_t *__restrict__ which,
                                                                 int32_t *__restrict__ votes_out, int svd) {
  __shared__ int s_votes[kPoseTile][4];
  const int p = blockIdx.y, m0 = blockIdx.x * kPoseTile, tid = threadIdx.x;
  const int mcount = min(kPoseTile, M - m0);
  if (tid < kPoseTile * 4) (&s_votes[0][0])[tid] = 0;
  __syncthreads();
  const T *mt = matches + (size_t)p * N * 4;
#pragma unroll 1
  for (int c0 = 0; c0 < N; c0 += kPoseThreads * kPosePts) {
    double px1[kPosePts], py1[kPosePts], px2[kPosePts], py2[kPosePts];
    bool have[kPosePts];
#pragma unroll
    for (int j = 0; j < kPosePts; ++j) {
      const int n = c0 + j * kPoseThreads + tid;   // coalesced: consecutive lanes, consecutive points
      have[j] = n < N;
      const int nc = have[j] ? n : 0;
      px1[j] = (double)mt[nc * 4 + 0]; py1[j] = (double)mt[nc * 4 + 1];
      px2[j] = (double)mt[nc * 4 + 2]; py2[j] = (double)mt[nc * 4 + 3];
    }
#pragma unroll 1
    for (int ml = 0; ml < mcount; ++ml) {
      double E[9], R1[9], R2[9], t[3];
#pragma unroll
      for (int q = 0; q < 9; ++q) E[q] = (double)models[((size_t)p * M + m0 + ml) * 9 + q];
      pose_decompose(E, R1, R2, t, svd);
      int v[4] = {0, 0, 0, 0};
#pragma unroll
      for (int j = 0; j < kPosePts; ++j) {   // unrolled: the point arrays stay in registers
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          double X[4];
          const double(&R)[9] = r == 0 ? R1 : R2;
          triangulate(R, t, px1[j], py1[j], px2[j], py2[j], X);
          // cheirality_check (cv_utils.py:186): Q[2]*Q[3] > 0 & Qh[2] < thr & (P Qh)[2] > 0 & (P Qh)[2] < thr with
          // Qh = Q / Q[3]; all four compared after multiplying by Q[3]^2 > 0 (no division)
          const double zw = X[2] * X[3];                                                   // z  = zw / X3^2
          const double dw = (R[6] * X[0] + R[7] * X[1] + R[8] * X[2] + t[2] * X[3]) * X[3];   // d2 = dw / X3^2
          const double lim = dist_thr * (X[3] * X[3]);
          const bool pos = have[j] && zw > 0 && zw < lim && dw > 0 && dw < lim;       // (R,  t)
          const bool neg = have[j] && zw < 0 && -zw < lim && dw < 0 && -dw < lim;     // (R, -t): X3 -> -X3
          v[r] += pos ? 1 : 0;
          v[2 + r] += neg ? 1 : 0;
        }
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int s = wave_sum(v[c]);
        if ((tid & 63) == 0 && s) atomicAdd(&s_votes[ml][c], s);
      }
    }
  }
  __syncthreads();
  if (tid < mcount) {
    const int ml = tid;
    const size_t o = (size_t)p * M + m0 + ml;
    int best = 0, bv = s_votes[ml][0];
#pragma unroll
    for (int c = 1; c < 4; ++c)
      if (s_votes[ml][c] > bv) { bv = s_votes[ml][c]; best = c; }   // torch.argmax: first maximum
    double E[9], R1[9], R2[9], t[3], Rg[9], tg[3];
#pragma unroll
    for (int q = 0; q < 9; ++q) { E[q] = (double)models[o * 9 + q]; Rg[q] = (double)gt_R[(size_t)p * 9 + q]; }
    double tn = 0;
#pragma unroll
    for (int q = 0; q < 3; ++q) { tg[q] = (double)gt_t[(size_t)p * 3 + q]; tn += tg[q] * tg[q]; }
    tn = 1.0 / (sqrt(tn) + 1e-8);
#pragma unroll
    for (int q = 0; q < 3; ++q) tg[q] *= tn;
    pose_decompose(E, R1, R2, t, svd);
    double R[9], ts[3], eq, et;
#pragma unroll
    for (int q = 0; q < 9; ++q) R[q] = (best & 1) ? R2[q] : R1[q];
#pragma unroll
    for (int q = 0; q < 3; ++q) ts[q] = (best >= 2) ? -t[q] : t[q];
    rt_error<double>(R, ts, Rg, tg, eq, et);
    err_R[o] = (T)eq;
    err_t[o] = (T)et;
    which[o] = best;
    if (votes_out) {
#pragma unroll
      for (int c = 0; c < 4; ++c) votes_out[o * 4 + c] = s_votes[ml][c];
    }
  }
}

template <typename T>
__global__ __launch_bounds__(64) void pose_error_bwd_kernel(const T *__restrict__ models, const T *__restrict__ gt_R,
                                                           const T *__restrict__ gt_t, const int32_t *__restrict__ which,
                                                           const T *__restrict__ g_err_R, const T *__restrict__ g_err_t,
                                                           int P, int M, T *__restrict__ grad_models) {
  const size_t o = (size_t)blockIdx.x * 64 + threadIdx.x;
  if (o >= (size_t)P * M) return;
  const int p = (int)(o / M);
  double Rg[9], tg[3], tn = 0;
#pragma unroll
  for (int q = 0; q < 9; ++q) Rg[q] = (double)gt_R[(size_t)p * 9 + q];
#pragma unroll
  for (int q = 0; q < 3; ++q) { tg[q] = (double)gt_t[(size_t)p * 3 + q]; tn += tg[q] * tg[q]; }
  tn = 1.0 / (sqrt(tn) + 1e-8);
#pragma unroll
  for (int q = 0; q < 3; ++q) tg[q] *= tn;
  const int best = which[o];
  const double gq = (double)g_err_R[o], gt = (double)g_err_t[o];
#pragma unroll 1
  for (int j = 0; j < 9; ++j) {
    Dual E[9], R1[9], R2[9], t[3];
#pragma unroll
    for (int q = 0; q < 9; ++q) E[q] = {(double)models[o * 9 + q], q == j ? 1.0 : 0.0};
    horn_decompose<Dual>(E, R1, R2, t);
    Dual R[9], ts[3], eq, et;
#pragma unroll
    for (int q = 0; q < 9; ++q) R[q] = (best & 1) ? R2[q] : R1[q];
#pragma unroll
    for (int q = 0; q < 3; ++q) ts[q] = (best >= 2) ? -t[q] : t[q];
    rt_error<Dual>(R, ts, Rg, tg, eq, et);
    const double g = gq * eq.d + gt * et.d;
    grad_models[o * 9 + j] = (T)(is_finite(g) ? g : 0.0);
  }
}

// ---- the inlier mask cv2.recoverPose returns (loss.py:99: `_, R, t, gt_inliers = cv2.recoverPose(gt_E, pts1, pts2)`;
// the reference's MatchLoss / ClassificationLoss use it as the ground-truth inlier mask): the points that pass the
// cheirality test of the winning candidate.  One block per (pair, model) -- meant for the ground-truth model of each pair
// (M = 1): pass 1 votes for the four candidates like pose_error_kernel, pass 2 re-triangulates with the winner only.
template <typename T>
__global__ __launch_bounds__(kPoseThreads) void recover_pose_mask_kernel(const T *__restrict__ matches, const T *__restrict__ models,
                                                                        int M, int N, double dist_thr, int32_t *__restrict__ which,
                                                                        uint8_t *__restrict__ mask) {
  __shared__ int s_votes[4];
  __shared__ int s_best;
  const int p = blockIdx.y, m = blockIdx.x, tid = threadIdx.x;
  if (tid < 4) s_votes[tid] = 0;
  __syncthreads();
  const T *mt = matches + (size_t)p * N * 4;
  double E[9], R1[9], R2[9], t[3];
#pragma unroll
  for (int q = 0; q < 9; ++q) E[q] = (double)models[((size_t)p * M + m) * 9 + q];
  horn_decompose<double>(E, R1, R2, t);
  auto test = [&](const double(&R)[9], int n, bool &pos, bool &neg) {
    double X[4];
    triangulate(R, t, (double)mt[n * 4], (double)mt[n * 4 + 1], (double)mt[n * 4 + 2], (double)mt[n * 4 + 3], X);
    const double zw = X[2] * X[3];
    const double dw = (R[6] * X[0] + R[7] * X[1] + R[8] * X[2] + t[2] * X[3]) * X[3];
    const double lim = dist_thr * (X[3] * X[3]);
    pos = zw > 0 && zw < lim && dw > 0 && dw < lim;
    neg = zw < 0 && -zw < lim && dw < 0 && -dw < lim;
  };
  int v[4] = {0, 0, 0, 0};
  for (int n = tid; n < N; n += kPoseThreads) {
    bool a, b;
    test(R1, n, a, b);
    v[0] += a; v[2] += b;
    test(R2, n, a, b);
    v[1] += a; v[3] += b;
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int sum = wave_sum(v[c]);
    if ((tid & 63) == 0 && sum) atomicAdd(&s_votes[c], sum);
  }
  __syncthreads();
  if (tid == 0) {
    int best = 0, bv = s_votes[0];
    for (int c = 1; c < 4; ++c)
      if (s_votes[c] > bv) { bv = s_votes[c]; best = c; }
    s_best = best;
    if (which) which[(size_t)p * M + m] = best;
  }
  __syncthreads();
  const int best = s_best;
  for (int n = tid; n < N; n += kPoseThreads) {
    bool a, b;
    if (best & 1) test(R2, n, a, b); else test(R1, n, a, b);
    mask[((size_t)p * M + m) * N + n] = (best >= 2) ? b : a;
  }
}

template <typename T>
int pose_error_launch(const T *matches, const T *models, const T *gt_R, const T *gt_t, int P, int M, int N,
                      double dist_thr, T *err_R, T *err_t, int32_t *which, int32_t *votes, hipStream_t st, int svd = 0) {
  dim3 grid((M + kPoseTile - 1) / kPoseTile, P);
  hipLaunchKernelGGL((pose_error_kernel<T>), grid, dim3(kPoseThreads), 0, st, matches, models, gt_R, gt_t, M, N, dist_thr,
                     err_R, err_t, which, votes, svd);
  return check_launch("pose_error_kernel");
}

template <typename T>
int pose_error_bwd_launch(const T *models, const T *gt_R, const T *gt_t, const int32_t *which, const T *g_err_R,
                          const T *g_err_t, int P, int M, T *grad_models, hipStream_t st) {
  const size_t total = (size_t)P * M;
  hipLaunchKernelGGL((pose_error_bwd_kernel<T>), dim3((unsigned)((total + 63) / 64)), dim3(64), 0, st, models, gt_R, gt_t,
                     which, g_err_R, g_err_t, P, M, grad_models);
  return check_launch("pose_error_bwd_kernel");
}

}  // namespace dr

extern "C" {

int dr_pose_error_fwd_f32(const float *matches, const float *models, const float *gt_R, const float *gt_t, int P, int M,
                          int N, double distance_threshold, float *err_R, float *err_t, int32_t *which, int32_t *votes,
                          void *stream) {
  DR_REQUIRE(P > 0 && M > 0 && N > 0 && P <= 65535, "bad sizes");
  DR_REQUIRE(matches && models && gt_R && gt_t && err_R && err_t && which, "null pointer");
  return dr::pose_error_launch<float>(matches, models, gt_R, gt_t, P, M, N, distance_threshold, err_R, err_t, which, votes,
                                      (hipStream_t)stream);
}

int dr_pose_error_fwd_f64(const double *matches, const double *models, const double *gt_R, const double *gt_t, int P,
                          int M, int N, double distance_threshold, double *err_R, double *err_t, int32_t *which,
                          int32_t *votes, void *stream) {
  DR_REQUIRE(P > 0 && M > 0 && N > 0 && P <= 65535, "bad sizes");
  DR_REQUIRE(matches && models && gt_R && gt_t && err_R && err_t && which, "null pointer");
  return dr::pose_error_launch<double>(matches, models, gt_R, gt_t, P, M, N, distance_threshold, err_R, err_t, which,
                                       votes, (hipStream_t)stream);
}

int dr_pose_error_svd_fwd_f32(const float *matches, const float *models, const float *gt_R, const float *gt_t, int P, int M,
                              int N, double distance_threshold, float *err_R, float *err_t, int32_t *which, int32_t *votes,
                              void *stream) {
  DR_REQUIRE(P > 0 && M > 0 && N > 0 && P <= 65535, "bad sizes");
  DR_REQUIRE(matches && models && gt_R && gt_t && err_R && err_t && which, "null pointer");
  return dr::pose_error_launch<float>(matches, models, gt_R, gt_t, P, M, N, distance_threshold, err_R, err_t, which, votes,
                                      (hipStream_t)stream, 1);
}

int dr_pose_error_svd_fwd_f64(const double *matches, const double *models, const double *gt_R, const double *gt_t, int P,
                              int M, int N, double distance_threshold, double *err_R, double *err_t, int32_t *which,
                              int32_t *votes, void *stream) {
  DR_REQUIRE(P > 0 && M > 0 && N > 0 && P <= 65535, "bad sizes");
  DR_REQUIRE(matches && models && gt_R && gt_t && err_R && err_t && which, "null pointer");
  return dr::pose_error_launch<double>(matches, models, gt_R, gt_t, P, M, N, distance_threshold, err_R, err_t, which,
                                       votes, (hipStream_t)stream, 1);
}

int dr_pose_error_bwd_f32(const float *models, const float *gt_R, const float *gt_t, const int32_t *which,
                          const float *grad_err_R, const float *grad_err_t, int P, int M, float *grad_models,
                          void *stream) {
  DR_REQUIRE(P > 0 && M > 0, "bad sizes");
  DR_REQUIRE(models && gt_R && gt_t && which && grad_err_R && grad_err_t && grad_models, "null pointer");
  return dr::pose_error_bwd_launch<float>(models, gt_R, gt_t, which, grad_err_R, grad_err_t, P, M, grad_models,
                                          (hipStream_t)stream);
}

int dr_pose_error_bwd_f64(const double *models, const double *gt_R, const double *gt_t, const int32_t *which,
                          const double *grad_err_R, const double *grad_err_t, int P, int M, double *grad_models,
                          void *stream) {
  DR_REQUIRE(P > 0 && M > 0, "bad sizes");
  DR_REQUIRE(models && gt_R && gt_t && which && grad_err_R && grad_err_t && grad_models, "null pointer");
  return dr::pose_error_bwd_launch<double>(models, gt_R, gt_t, which, grad_err_R, grad_err_t, P, M, grad_models,
                                           (hipStream_t)stream);
}

int dr_recover_pose_mask_f32(const float *matches, const float *models, int P, int M, int N, double distance_threshold,
                             int32_t *which, uint8_t *mask, void *stream) {
  DR_REQUIRE(P > 0 && M > 0 && N > 0 && P <= 65535, "bad sizes");
  DR_REQUIRE(matches && models && mask, "null pointer");
  hipLaunchKernelGGL((dr::recover_pose_mask_kernel<float>), dim3(M, P), dim3(dr::kPoseThreads), 0, (hipStream_t)stream, matches,
                     models, M, N, distance_threshold, which, mask);
  return dr::check_launch("recover_pose_mask_kernel");
}

int dr_recover_pose_mask_f64(const double *matches, const double *models, int P, int M, int N, double distance_threshold,
                             int32_t *which, uint8_t *mask, void *stream) {
  DR_REQUIRE(P > 0 && M > 0 && N > 0 && P <= 65535, "bad sizes");
  DR_REQUIRE(matches && models && mask, "null pointer");
  hipLaunchKernelGGL((dr::recover_pose_mask_kernel<double>), dim3(M, P), dim3(dr::kPoseThreads), 0, (hipStream_t)stream, matches,
                     models, M, N, distance_threshold, which, mask);
  return dr::check_launch("recover_pose_mask_kernel");
}

}  // extern "C"
