// Building blocks of the minimal solvers (K3).  One lane = one minimal sample; arithmetic in f64
// (full-rate FMA on CDNA4, and the only way the f32 entry points meet the 1e-4 model tolerance:
// the reference's own f32 five-point output is off by >1e-3 for 10 % of its solutions).
//
// Per-lane matrices that need run-time (pivot-dependent) indexing live in LDS in an
// element-major / lane-minor layout: element e of lane l is at ws[e*64 + l].  Consecutive lanes
// are 8 bytes apart, so every ds_read_b64/ds_write_b64 of a wave is bank-conflict free whatever
// element index each lane uses (e*512 bytes is a multiple of the 256-byte bank row).
#pragma once
#include "dr_common.hpp"
#include "sturm_eval_asm.hpp"

namespace dr {

// view of one lane's slice of the block's LDS workspace
struct LaneWs {
  double *base;      // &ws[lane]  (or &ws[lane >> 1] when two lanes share one sample)
  int stride = 64;   // lanes (samples) per block row
  __device__ __forceinline__ double &operator[](int e) const { return base[e * stride]; }
};

// orders the LDS traffic of a single-wave block (compiler-level: the hardware executes a wave's LDS instructions in order)
__device__ __forceinline__ void wave_lds_order() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ double dsign(double a, double b) { return b >= 0 ? fabs(a) : -fabs(a); }

__device__ __forceinline__ double rsqrt_nr(double x) {
  double y = __builtin_amdgcn_rsq(x);
  y = y * fma(-0.5 * x, y * y, 1.5);
  return y * fma(-0.5 * x, y * y, 1.5);
}
__device__ __forceinline__ double rcp_nr(double x) {
  double z = __builtin_amdgcn_rcp(x);
  z = z * fma(-x, z, 2.0);
  return z * fma(-x, z, 2.0);
}

// Round 6 (DR_K3_FAST_DIV): divisions and square roots of the five-point stages by v_rcp_f64 / v_rsq_f64 + two Newton steps (5-6
// instructions, results within 1-2 ulp) instead of the IEEE sequences (div_scale x 2, rcp, 4-5 fma, div_fmas, div_fixup: 10-12
// instructions in one dependent chain; sqrt likewise) -- ~110 of them per sample, a tenth of the kernels' instruction stream.
// Same behaviour at the edges the callers test for: b = 0 or non-finite gives a non-finite quotient (NaN instead of inf: every caller
// asks is_finite), fsqrt(0) = 0.
#ifndef DR_K3_FAST_DIV
#define DR_K3_FAST_DIV 1
#endif
__device__ __forceinline__ double frcp(double b) { return DR_K3_FAST_DIV ? rcp_nr(b) : 1.0 / b; }
__device__ __forceinline__ double fdiv(double a, double b) { return DR_K3_FAST_DIV ? a * rcp_nr(b) : a / b; }
__device__ __forceinline__ double fsqrt(double x) { return DR_K3_FAST_DIV ? (x > 0 ? x * rsqrt_nr(x) : x) : sqrt(x); }
__device__ __forceinline__ double frsqrt(double x) { return DR_K3_FAST_DIV ? rsqrt_nr(x) : 1.0 / sqrt(x); }


// ------------------------------------------------------------------------------------------------
// Null space of a K x 9 matrix (rows = equations) by Householder QR of its transpose; registers only.
// A[r][c] is overwritten.  nb[i] (i < 9-K) are orthonormal vectors spanning null(A) when rank A = K.
// ------------------------------------------------------------------------------------------------
template <int K>
__device__ __forceinline__ void null_space_qr(double (&A)[K][9], double (&nb)[9 - K][9]) {
  double beta[K];
#pragma unroll
  for (int j = 0; j < K; ++j) {
    // column j of A^T = row j of A, entries j..8
    double nrm2 = 0;
#pragma unroll
    for (int i = j; i < 9; ++i) nrm2 += A[j][i] * A[j][i];
    const double nrm = fsqrt(nrm2);
    const double alpha = -dsign(nrm, A[j][j]);
    const double v0 = A[j][j] - alpha;
    // v = (v0, A[j][j+1..8]);  beta = 2 / v^T v = -1/(alpha*v0)
    const double vtv = v0 * v0 + (nrm2 - A[j][j] * A[j][j]);
    beta[j] = vtv > 0 ? 2.0 * frcp(vtv) : 0.0;
    A[j][j] = v0;
#pragma unroll
    for (int c = j + 1; c < K; ++c) {
      double dot = 0;
#pragma unroll
      for (int i = j; i < 9; ++i) dot += A[j][i] * A[c][i];
      dot *= beta[j];
#pragma unroll
      for (int i = j; i < 9; ++i) A[c][i] -= dot * A[j][i];
    }
  }
#pragma unroll
  for (int t = 0; t < 9 - K; ++t) {
    double q[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) q[i] = (i == K + t) ? 1.0 : 0.0;
#pragma unroll
    for (int j = K - 1; j >= 0; --j) {
      double dot = 0;
#pragma unroll
      for (int i = j; i < 9; ++i) dot += A[j][i] * q[i];
      dot *= beta[j];
#pragma unroll
      for (int i = j; i < 9; ++i) q[i] -= dot * A[j][i];
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) nb[t][i] = q[i];
  }
}

// ------------------------------------------------------------------------------------------------
// All real roots of a degree-D polynomial (coefficients ascending), robustly in plain arithmetic.
//   * |z| <= 1 : roots of p in [-1, 1];   |z| > 1 : z = 1/w with w a root of the reversed polynomial in (-1, 1).
//     Both searches live on [-1, 1], so there is no root bound to estimate and no huge outer bracket to crawl through.
//   * On [-1, 1] the real roots of q' split the interval into pieces on which q is monotone; going from the linear
//     (D-1)-th derivative up to the polynomial itself, every root is bracketed by the previous level's points.
//     Each level emits exactly d sorted points (roots, or the right end of an empty bracket -- a harmless extra
//     breakpoint), so nothing is ever compacted and all indexing is static.
//   * Every bracket is refined by a FIXED schedule -- kBis bisections, then kNewt safeguarded Newton steps -- so the
//     64 samples of a wave never wait for a slow one.  A root that Newton does not finish is still inside a bracket
//     of width 2^(1-kBis); the five-point callers polish (x, y, z) on the defining constraints afterwards.
// Replaces the per-sample companion-matrix eigvals of nister.py:361-370 / the Sturm recursion of math_utils.py.
// ------------------------------------------------------------------------------------------------
// DR_ROOT_F32_LOW = 1: the levels below the polynomial itself (which only provide the breakpoints between which the next
// level looks for sign changes) run in f32 -- half the cycles per FMA and per select on gfx950.  A breakpoint y~ = y* + e
// (y* a critical point of the next level's polynomial g, g'(y*) = 0) is used only through sign g(y~) = sign(g(y*) +
// g''(y*) e^2 / 2): second order in e, so f32 breakpoints bracket the same roots unless two of them are closer than ~1e-6
// (nearly a double root, where the f64 search is at the mercy of rounding as well).  The last level, which produces the
// roots, is always f64.
#ifndef DR_ROOT_F32_LOW
#define DR_ROOT_F32_LOW 0
#endif
#ifndef DR_K3_STURM
#define DR_K3_STURM 1   // 1: the two-lanes-per-sample kernels isolate the roots by a Sturm sequence (real_roots_half_sturm)
#endif
#ifndef DR_K3_STURM_DRAIN
#define DR_K3_STURM_DRAIN 0   // 1: emitting an isolated interval does not cost a step of the isolation loop (round 5: 13 -> 11 steps for
                              // the slowest lane of a wave, but every step pays the extra loop: 165.1 -> 170.9 us at 131 072 samples; off)
#endif
#ifndef DR_K3_STURM_FALLBACK
#define DR_K3_STURM_FALLBACK 1   // 1: a lane whose Sturm chain loses a degree has its roots found by the whole wave on a grid (round 5)
#endif
#ifndef DR_K3_SYMG
#define DR_K3_SYMG 1          // 1: the six distinct entries of E E^T once in Nister's lane-pair kernel (constraint_rows; round 5)
#endif
#ifndef DR_K3_ISO_FLAT
#define DR_K3_ISO_FLAT 1      // 1: the isolation step as predicates + selects instead of an if / else-if chain (round 5)
#endif
#ifndef DR_ROOT_BIS_LOW
#define DR_ROOT_BIS_LOW 6
#define DR_ROOT_NEWT_LOW 4
#endif
__device__ __forceinline__ double root_rcp(double v) { return __builtin_amdgcn_rcp(v); }
// The safeguarded Newton step of the fixed schedules: y - step when that stays inside the bracket [a, b], else the midpoint --
// EXCEPT when the step has already shrunk to rounding level.  After `a = y` (or `b = y`) a converged iterate sits ON the end of
// its bracket, and a last-bit step (zero, or one ulp outwards: the sign of f is noise there) used to fail the strict test
// `a < yn < b` and send the iterate to the midpoint -- half a bracket away from the root it had found, which the remaining
// steps of the schedule then only bisected back (rounds 1-2: 22 % of the roots left the refinement with a backward error above
// 1e-13, up to 1e-2, and the Gauss-Newton polish repaired them; tests/test_gpu_roots.py).  DR_ROOT_KEEP_CONVERGED=0: old rule.
#ifndef DR_ROOT_KEEP_CONVERGED
#define DR_ROOT_KEEP_CONVERGED 1
#endif
template <typename F>
__device__ __forceinline__ F safeguarded_newton(F y, F step, F a, F b) {
  F yn = y - step;
#if DR_ROOT_KEEP_CONVERGED
  const F tiny = (sizeof(F) == 8 ? (F)4e-16 : (F)5e-7) * ((F)1 + fabs(y));
  if (!(yn >= a && yn <= b)) yn = (fabs(step) <= tiny) ? y : (F)0.5 * (a + b);
#else
  if (!(yn > a && yn < b)) yn = (F)0.5 * (a + b);
#endif
  return yn;
}
__device__ __forceinline__ float root_rcp(float v) { return __builtin_amdgcn_rcpf(v); }

template <int D, int kBisLast, int kNewtLast>
__device__ __forceinline__ void roots_in_unit(const double (&c)[D + 1], double (&x)[D], unsigned &mask, double tail_tol) {
  double pts[D + 1];
#pragma unroll
  for (int i = 0; i <= D; ++i) pts[i] = 1.0;
  pts[0] = -1.0;
  mask = 0;
  // one level: F = arithmetic type of the level, d = degree (a constant after unrolling)
  auto level = [&](auto tag, const int d) {
    using F = decltype(tag);
    F q[D + 1];  // q = p^(D-d): degree d, ascending
#pragma unroll
    for (int i = 0; i <= D; ++i) q[i] = 0;
#pragma unroll
    for (int i = 0; i <= d; ++i) {
      double f = 1;
#pragma unroll
      for (int t = 0; t < D - d; ++t) f *= (double)(i + D - d - t);
      q[i] = (F)(c[i + D - d] * f);
    }
    auto evalf = [&](F t) {
      F fx = q[d];
#pragma unroll
      for (int i = d - 1; i >= 0; --i) fx = fx * t + q[i];
      return fx;
    };
    auto eval2 = [&](F t, F &fx, F &dfx) {
      fx = q[d];
      dfx = 0;
#pragma unroll
      for (int i = d - 1; i >= 0; --i) {
        dfx = dfx * t + fx;
        fx = fx * t + q[i];
      }
    };
    F a[D], b[D], y[D];
    bool neg_a[D];
    unsigned has = 0;
    F fprev = evalf((F)pts[0]);
#pragma unroll
    for (int i = 0; i < d; ++i) {
      const F lo = (F)pts[i], hi = (i == d - 1) ? (F)1 : (F)pts[i + 1];
      const F fhi = evalf(hi);
      if (((fprev < 0) != (fhi < 0)) && (hi > lo)) has |= 1u << i;
      a[i] = lo;
      b[i] = hi;
      neg_a[i] = fprev < 0;
      fprev = fhi;
    }
#ifdef DR_PROFILE_STAGES
    if (D == 10) atomicAdd(&::dr::g_stage_cycles[5 + d], (unsigned long long)__popc(has));   // brackets with a sign change, per level
#endif
    const int kBis = (d == D) ? kBisLast : DR_ROOT_BIS_LOW;
    const int kNewt = (d == D) ? kNewtLast : DR_ROOT_NEWT_LOW;
#pragma unroll 1
    for (int it = 0; it < kBis; ++it) {
#pragma unroll
      for (int i = 0; i < d; ++i) {
        const F m = (F)0.5 * (a[i] + b[i]);
        const F fm = evalf(m);
        const bool left = (fm < 0) == neg_a[i], hit = fm == (F)0;   // a midpoint that IS the root closes the bracket on it
        a[i] = (left || hit) ? m : a[i];
        b[i] = (left && !hit) ? b[i] : m;
      }
    }
#pragma unroll
    for (int i = 0; i < d; ++i) y[i] = (F)0.5 * (a[i] + b[i]);
#pragma unroll 1
    for (int it = 0; it < kNewt; ++it) {
#pragma unroll
      for (int i = 0; i < d; ++i) {
        F fx, dfx;
        eval2(y[i], fx, dfx);
        const bool left = (fx < 0) == neg_a[i];
        a[i] = left ? y[i] : a[i];
        b[i] = left ? b[i] : y[i];
        const F yn = safeguarded_newton<F>(y[i], fx * root_rcp(dfx), a[i], b[i]);
        y[i] = (fx == 0) ? y[i] : yn;
      }
    }
    if (d == D && tail_tol > 0) {
      // optional (callers without a polish step, i.e. the 7-point cubic): brackets that the fixed schedule did not
      // bring below `tail_tol` keep alternating safeguarded Newton / bisection.  The five-point solvers skip this --
      // over 64 samples x 10 brackets some bracket is always slow and the whole wave would wait for it; their roots
      // are refined by the Gauss-Newton polish on the defining constraints instead, which is well conditioned.
      const F tol = (F)tail_tol;
      unsigned live = 0;
#pragma unroll
      for (int i = 0; i < d; ++i) {
        F fx, dfx;
        eval2(y[i], fx, dfx);
        const F step = fabs(fx * root_rcp(dfx));
        if (((has >> i) & 1u) && !(step <= tol * ((F)1 + fabs(y[i]))) && (b[i] - a[i]) > tol) live |= 1u << i;
      }
      for (int it = 0; it < 100 && __any(live != 0); ++it) {
#pragma unroll
        for (int i = 0; i < d; ++i) {
          if (!((live >> i) & 1u)) continue;
          F fx, dfx;
          eval2(y[i], fx, dfx);
          const bool left = (fx < 0) == neg_a[i];
          a[i] = left ? y[i] : a[i];
          b[i] = left ? b[i] : y[i];
          F yn = y[i] - fx / dfx;
          if (!(yn > a[i] && yn < b[i]) || (it & 1)) yn = (F)0.5 * (a[i] + b[i]);
          const F dx = fabs(yn - y[i]);
          y[i] = yn;
          if (dx <= tol * ((F)1 + fabs(yn)) || fx == 0 || (b[i] - a[i]) <= tol * ((F)1 + fabs(yn))) live &= ~(1u << i);
        }
      }
    }
    // breakpoints of the next level (kept in double so that an empty bracket's right end stays exact)
    double nxt[D];
#pragma unroll
    for (int i = 0; i < d; ++i) {
      const double hi = (i == d - 1) ? 1.0 : pts[i + 1];
      nxt[i] = ((has >> i) & 1u) ? (double)y[i] : hi;
    }
#pragma unroll
    for (int i = 0; i < d; ++i) pts[i + 1] = nxt[i];
    if (d == D) {
#pragma unroll
      for (int i = 0; i < D; ++i) x[i] = nxt[i];
      mask = has;
    }
  };
#pragma unroll
  for (int d = 1; d <= D; ++d) {
    if (DR_ROOT_F32_LOW && d < D) level(float{}, d);
    else level(double{}, d);
  }
}

// One half of the search (two lanes per sample: the even lane takes |z| <= 1, the odd lane |z| > 1 through the reversed
// polynomial).  roots[0..count-1] dense.
#ifndef DR_ROOT_BIS_LAST
#define DR_ROOT_BIS_LAST 10
#define DR_ROOT_NEWT_LAST 6
#endif
template <int D, int kBisLast = DR_ROOT_BIS_LAST, int kNewtLast = DR_ROOT_NEWT_LAST>
__device__ __forceinline__ void real_roots_half(const double (&c)[D + 1], bool outer, double (&roots)[D], int &count) {
  double cmax = 0;
#pragma unroll
  for (int i = 0; i <= D; ++i) cmax = fmax(cmax, fabs(c[i]));
  const bool ok = is_finite(cmax) && cmax > 0;
  const double sc = ok ? 1.0 / cmax : 0.0;
  double ch[D + 1];
#pragma unroll
  for (int i = 0; i <= D; ++i) {
    const double a = ok ? c[i] * sc : (i == 0 ? 1.0 : 0.0);
    const double b = ok ? c[D - i] * sc : (i == 0 ? 1.0 : 0.0);
    ch[i] = outer ? b : a;
  }
  double x[D];
  unsigned mk;
  roots_in_unit<D, kBisLast, kNewtLast>(ch, x, mk, 0.0);
  if (!ok) mk = 0;
  count = 0;
#pragma unroll
  for (int i = 0; i < D; ++i) roots[i] = 0.0;
#pragma unroll
  for (int k = 0; k < D; ++k) {
    const double v = outer ? 1.0 / x[k] : x[k];
    const bool take = ((mk >> k) & 1u) && (!outer || (fabs(x[k]) > 1e-9 && fabs(x[k]) < 1.0));
    if (take) {
#pragma unroll
      for (int t = 0; t < D; ++t) roots[t] = (t == count) ? v : roots[t];
    }
    count += take ? 1 : 0;
  }
}

// ------------------------------------------------------------------------------------------------
// The same search, organised per WAVE instead of per lane (blocks of exactly one wave; LDS workspace `RootWs`).
// roots_in_unit refines all d brackets of level d in every lane, although on the five-point polynomials only 0.6 (level 1)
// to 2.3 (levels 8-10) of them hold a sign change: 3520 bracket refinements per wave, 1083 of them useful.  Here a lane
// only evaluates its level's polynomial at its breakpoints (d (d + 1) FMAs); the brackets that do change sign become
// tasks (source lane, bracket) in an LDS queue and the 64 lanes take them R at a time (R interleaved tasks per lane hide
// the f64 FMA latency the d brackets of the per-lane version used to hide): 1-3 rounds per level instead of d.  A task
// gathers its source lane's coefficients and bracket ends from LDS, runs the SAME bisection / Newton schedule on them as
// roots_in_unit (so the results are bit-identical) and writes the root into the source lane's breakpoint list of the
// next level (double-buffered: tasks of one level must all see the old breakpoints).
// (Evaluating a task's polynomial as even + odd part in t^2 -- two chains of half the length -- is not faster, 73.4 vs
// 72.6 us, and not bit-identical: dropped.  Measured f64 FMA on gfx950: 5.1 clk issue, 8.5 clk dependent issue.)
// ------------------------------------------------------------------------------------------------
template <int D>
struct RootWs {
  double *pts[2];    // (D + 1) x 64 each: breakpoint i of lane l at [i * 64 + l]
  double *q;         // (D + 1) x 64: coefficient i of lane l's current polynomial
  uint16_t *queue;   // D x 64 tasks: source lane | bracket << 6 | (sign at the left end) << 10
#ifdef DR_PROFILE_STAGES
  unsigned long long *prof;   // 16 cycle accumulators of the block's (single) wave: [0] per-lane part, [d] level d
  static constexpr int kDoubles = 3 * (D + 1) * 64 + (D * 64 + 3) / 4 + 16;
  __device__ __forceinline__ explicit RootWs(double *ws)
      : pts{ws, ws + (D + 1) * 64}, q(ws + 2 * (D + 1) * 64), queue(reinterpret_cast<uint16_t *>(ws + 3 * (D + 1) * 64)),
        prof(reinterpret_cast<unsigned long long *>(ws + 3 * (D + 1) * 64 + (D * 64 + 3) / 4)) {}
#else
  static constexpr int kDoubles = 3 * (D + 1) * 64 + (D * 64 + 3) / 4;
  __device__ __forceinline__ explicit RootWs(double *ws)
      : pts{ws, ws + (D + 1) * 64}, q(ws + 2 * (D + 1) * 64), queue(reinterpret_cast<uint16_t *>(ws + 3 * (D + 1) * 64)) {}
#endif
};

template <int D, int d, int R>
__device__ __forceinline__ void root_tasks(const RootWs<D> &ws, const double *__restrict__ cur, double *__restrict__ nxt, int total, int lane,
                                           int kBis, int kNewt) {
#pragma unroll 1
  for (int base = 0; base < total; base += 64 * R) {
    double a[R], b[R], y[R], qq[R][d + 1];
    bool neg[R], val[R];
    int dst[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int t = base + 64 * r + lane;
      val[r] = t < total;
      const unsigned m = ws.queue[val[r] ? t : base];
      const int src = m & 63, i = (m >> 6) & 15;
      neg[r] = (m >> 10) & 1u;
      a[r] = cur[i * 64 + src];
      b[r] = cur[(i + 1) * 64 + src];
      dst[r] = (i + 1) * 64 + src;
#pragma unroll
      for (int k = 0; k <= d; ++k) qq[r][k] = ws.q[k * 64 + src];
    }
#pragma unroll 1
    for (int it = 0; it < kBis; ++it) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const double m = 0.5 * (a[r] + b[r]);
        double fx = qq[r][d];
#pragma unroll
        for (int k = d - 1; k >= 0; --k) fx = fx * m + qq[r][k];
        const bool left = (fx < 0) == neg[r], hit = fx == 0.0;   // a midpoint that IS the root closes the bracket on it
        a[r] = (left || hit) ? m : a[r];
        b[r] = (left && !hit) ? b[r] : m;
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) y[r] = 0.5 * (a[r] + b[r]);
#pragma unroll 1
    for (int it = 0; it < kNewt; ++it) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        double fx = qq[r][d], dfx = 0;
#pragma unroll
        for (int k = d - 1; k >= 0; --k) {
          dfx = dfx * y[r] + fx;
          fx = fx * y[r] + qq[r][k];
        }
        const bool left = (fx < 0) == neg[r];
        a[r] = left ? y[r] : a[r];
        b[r] = left ? b[r] : y[r];
        const double yn = safeguarded_newton<double>(y[r], fx * __builtin_amdgcn_rcp(dfx), a[r], b[r]);
        y[r] = (fx == 0.0) ? y[r] : yn;
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
      if (val[r]) nxt[dst[r]] = y[r];
  }
}

template <int D, int d, int kBisLast, int kNewtLast>
__device__ __forceinline__ void root_levels_wave(const double (&c)[D + 1], const RootWs<D> &ws, int lane, unsigned &mask) {
  const double *cur = ws.pts[(d - 1) & 1];
  double *nxt = ws.pts[d & 1];
#ifdef DR_PROFILE_STAGES
  const unsigned long long _lvl_t0 = __builtin_readcyclecounter();
#endif
  double q[d + 1];  // q = p^(D-d): degree d, ascending
#pragma unroll
  for (int i = 0; i <= d; ++i) {
    double f = 1;
#pragma unroll
    for (int t = 0; t < D - d; ++t) f *= (double)(i + D - d - t);
    q[i] = c[i + D - d] * f;
    ws.q[i * 64 + lane] = q[i];
  }
  auto evalf = [&](double t) {
    double fx = q[d];
#pragma unroll
    for (int i = d - 1; i >= 0; --i) fx = fx * t + q[i];
    return fx;
  };
  double p[d + 1];
#pragma unroll
  for (int i = 0; i <= d; ++i) p[i] = cur[i * 64 + lane];
  // an empty bracket hands its right end to the next level: that is the old breakpoint itself (or the interval's end)
#pragma unroll
  for (int i = 1; i < d; ++i) nxt[i * 64 + lane] = p[i];
  unsigned has = 0;
  int offs = 0;
  double fprev = evalf(p[0]);
#pragma unroll
  for (int i = 0; i < d; ++i) {
    const double lo = p[i], hi = (i == d - 1) ? 1.0 : p[i + 1];
    const double fhi = evalf(hi);
    const bool h = ((fprev < 0) != (fhi < 0)) && (hi > lo);
    const unsigned long long bm = __ballot(h);
    const int pos = offs + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bm, 0u));
    if (h) {
      has |= 1u << i;
      ws.queue[pos] = (uint16_t)(lane | (i << 6) | ((fprev < 0) ? 1 << 10 : 0));
    }
    offs += __popcll(bm);
    fprev = fhi;
  }
  wave_lds_order();
#ifdef DR_PROFILE_STAGES
  if (lane == 0) ws.prof[0] += __builtin_readcyclecounter() - _lvl_t0;   // per-lane part, all levels
#endif
  const int kBis = (d == D) ? kBisLast : DR_ROOT_BIS_LOW;
  const int kNewt = (d == D) ? kNewtLast : DR_ROOT_NEWT_LOW;
  if (offs <= 64) root_tasks<D, d, 1>(ws, cur, nxt, offs, lane, kBis, kNewt);
  else if (offs <= 128) root_tasks<D, d, 2>(ws, cur, nxt, offs, lane, kBis, kNewt);
  else root_tasks<D, d, 3>(ws, cur, nxt, offs, lane, kBis, kNewt);
  wave_lds_order();
#ifdef DR_PROFILE_STAGES
  if (lane == 0) ws.prof[d] += __builtin_readcyclecounter() - _lvl_t0;   // whole level d
#endif
  if constexpr (d < D) root_levels_wave<D, d + 1, kBisLast, kNewtLast>(c, ws, lane, mask);
  else mask = has;
}

// wave-cooperative real_roots_half: every lane of the (single-wave) block must call it
template <int D, int kBisLast = DR_ROOT_BIS_LAST, int kNewtLast = DR_ROOT_NEWT_LAST>
__device__ __forceinline__ void real_roots_half_wave(const double (&c)[D + 1], bool outer, double (&roots)[D], int &count, double *lds_ws,
                                                     int lane) {
  const RootWs<D> ws(lds_ws);
  double cmax = 0;
#pragma unroll
  for (int i = 0; i <= D; ++i) cmax = fmax(cmax, fabs(c[i]));
  const bool ok = is_finite(cmax) && cmax > 0;
  const double sc = ok ? 1.0 / cmax : 0.0;
  double ch[D + 1];
#pragma unroll
  for (int i = 0; i <= D; ++i) {
    const double a = ok ? c[i] * sc : (i == 0 ? 1.0 : 0.0);
    const double b = ok ? c[D - i] * sc : (i == 0 ? 1.0 : 0.0);
    ch[i] = outer ? b : a;
  }
#pragma unroll
  for (int i = 0; i <= D; ++i) {
    ws.pts[0][i * 64 + lane] = i == 0 ? -1.0 : 1.0;
    ws.pts[1][i * 64 + lane] = i == 0 ? -1.0 : 1.0;
  }
  unsigned mk;
#ifdef DR_PROFILE_STAGES
  if (lane < 16) ws.prof[lane] = 0;
#endif
  root_levels_wave<D, 1, kBisLast, kNewtLast>(ch, ws, lane, mk);
#ifdef DR_PROFILE_STAGES
  if (D == 10 && lane <= 10) atomicAdd(&::dr::g_stage_cycles[lane == 0 ? 1 : 5 + lane], ws.prof[lane]);
#endif
  if (!ok) mk = 0;
  const double *res = ws.pts[D & 1];
  count = 0;
#pragma unroll
  for (int i = 0; i < D; ++i) roots[i] = 0.0;
#pragma unroll
  for (int k = 0; k < D; ++k) {
    const double xk = res[(k + 1) * 64 + lane];
    const double v = outer ? 1.0 / xk : xk;
    const bool take = ((mk >> k) & 1u) && (!outer || (fabs(xk) > 1e-9 && fabs(xk) < 1.0));
    if (take) {
#pragma unroll
      for (int t = 0; t < D; ++t) roots[t] = (t == count) ? v : roots[t];
    }
    count += take ? 1 : 0;
  }
}

// ------------------------------------------------------------------------------------------------
// Root ISOLATION by a Sturm sequence (DR_K3_STURM): the derivative chain above isolates the roots of p by finding, level by
// level, ALL roots of p', p'', ... in [-1, 1] first -- 45 k of the 56 k cycles a wave spends in the root search, although a
// sample-half holds 2.3 real roots on average.  Here a lane builds the Sturm chain of its polynomial once (f0 = p, f1 = p',
// f_{k+1} = -rem(f_{k-1}, f_k) as division-free pseudo-remainders with positive multipliers, each renormalised to max |coef| = 1
// by an approximate reciprocal -- any positive scale keeps the signs), keeps it in registers (66 doubles) and bisects (-1, 1]
// on the sign-variation count V: an interval (a, b] holds V(a) - V(b) distinct real roots.  Intervals wait on a per-lane
// stack in LDS (left part on top, so isolated intervals come off in ascending order); they share one array of D entries with
// the output list (isolated + pending <= number of roots <= D).  4.4 evaluations of V per polynomial on RANSAC samples, 12-14
// for the slowest lane of a wave (scratch/sturm_proto.py, against numpy's eigenvalues: 0-5e-5 of the real roots missed, the
// same as the idealised derivative chain misses).  The isolated roots are then refined by the SAME wave-wide task rounds on p
// (bisection + Newton on a bracket with a sign change), so the final accuracy is the one of the level-D tasks.
// ------------------------------------------------------------------------------------------------
#ifndef DR_K3_VAR_SCHED
#define DR_K3_VAR_SCHED 0
#endif
// 1 (round 6): the chain evaluation of the isolation step as one hand-scheduled asm block (sturm_eval_asm.hpp): bit-identical roots,
// 66 instead of ~130 instructions per evaluation; Nister 159.7 -> 156.3 us, Stewenius 192.6 -> 189.2 us at 131 072 samples (same box)
#ifndef DR_K3_VAR_ASM
#define DR_K3_VAR_ASM 1
#endif
#ifndef DR_K3_TASK_ESTRIN
#define DR_K3_TASK_ESTRIN 0
#endif
template <int D>
struct SturmWs {
  double *lo, *hi;     // D x 64 each: entry e of lane l at [e * 64 + l]; after the refine tasks lo holds the root
  uint32_t *vv;        // D x 64: V(lo) | sign(p(lo)) << 4 | V(hi) << 8 | sign(p(hi)) << 12
  double *q;           // (D + 1) x 64: coefficient i of lane l's polynomial (for the tasks)
  uint16_t *queue;     // D x 64 tasks: source lane | entry << 6 | (p(lo) < 0) << 10
  static constexpr int kDoubles = 2 * D * 64 + D * 64 / 2 + (D + 1) * 64 + (D * 64 + 3) / 4;
  __device__ __forceinline__ explicit SturmWs(double *ws)
      : lo(ws), hi(ws + D * 64), vv(reinterpret_cast<uint32_t *>(ws + 2 * D * 64)), q(ws + 2 * D * 64 + D * 64 / 2),
        queue(reinterpret_cast<uint16_t *>(ws + 2 * D * 64 + D * 64 / 2 + (D + 1) * 64)) {}
};

// Estrin's scheme for the refine tasks (round 6, DR_K3_TASK_ESTRIN): a bracket's bisection / Newton steps are one dependent chain per
// task, and a wave alone on its SIMD waits ~8.5 clocks for every dependent v_fma_f64 -- Horner's rule for degree 10 is ten of them in
// a row.  Pairing the coefficients (q0 + q1 x) + x^2 (q2 + q3 x) ... costs three squarings and three more FMAs and is four levels deep.
template <int D>
__device__ __forceinline__ double estrin_eval(const double (&q)[D + 1], double x) {
  static_assert(D == 10, "written out for the degree-10 polynomials of the five-point solvers");
  const double x2 = x * x, x4 = x2 * x2, x8 = x4 * x4;
  const double a0 = __builtin_fma(q[1], x, q[0]), a1 = __builtin_fma(q[3], x, q[2]), a2 = __builtin_fma(q[5], x, q[4]);
  const double a3 = __builtin_fma(q[7], x, q[6]), a4 = __builtin_fma(q[9], x, q[8]);
  const double b0 = __builtin_fma(a1, x2, a0), b1 = __builtin_fma(a3, x2, a2), b2 = __builtin_fma(q[10], x2, a4);
  const double c0 = __builtin_fma(b1, x4, b0);
  return __builtin_fma(b2, x8, c0);
}
// value and derivative: p' = sum (j + 1) q[j + 1] x^j, degree 9, by the same pairing
template <int D>
__device__ __forceinline__ void estrin_eval_d(const double (&q)[D + 1], double x, double &fx, double &dfx) {
  static_assert(D == 10, "written out for the degree-10 polynomials of the five-point solvers");
  const double x2 = x * x, x4 = x2 * x2, x8 = x4 * x4;
  const double a0 = __builtin_fma(q[1], x, q[0]), a1 = __builtin_fma(q[3], x, q[2]), a2 = __builtin_fma(q[5], x, q[4]);
  const double a3 = __builtin_fma(q[7], x, q[6]), a4 = __builtin_fma(q[9], x, q[8]);
  const double b0 = __builtin_fma(a1, x2, a0), b1 = __builtin_fma(a3, x2, a2), b2 = __builtin_fma(q[10], x2, a4);
  fx = __builtin_fma(b2, x8, __builtin_fma(b1, x4, b0));
  const double e0 = __builtin_fma(2.0 * q[2], x, q[1]), e1 = __builtin_fma(4.0 * q[4], x, 3.0 * q[3]);
  const double e2 = __builtin_fma(6.0 * q[6], x, 5.0 * q[5]), e3 = __builtin_fma(8.0 * q[8], x, 7.0 * q[7]);
  const double e4 = __builtin_fma(10.0 * q[10], x, 9.0 * q[9]);
  const double g0 = __builtin_fma(e1, x2, e0), g1 = __builtin_fma(e3, x2, e2);
  dfx = __builtin_fma(e4, x8, __builtin_fma(g1, x4, g0));
}

// the refine rounds: root_tasks with the bracket ends in separate arrays and the polynomial always of degree D
template <int D, int R>
__device__ __forceinline__ void sturm_tasks(const SturmWs<D> &ws, int total, int lane, int kBis, int kNewt) {
#pragma unroll 1
  for (int base = 0; base < total; base += 64 * R) {
    double a[R], b[R], y[R], qq[R][D + 1];
    bool neg[R], val[R];
    int dst[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int t = base + 64 * r + lane;
      val[r] = t < total;
      const unsigned m = ws.queue[val[r] ? t : base];
      const int src = m & 63, e = (m >> 6) & 15;
      neg[r] = (m >> 10) & 1u;
      dst[r] = e * 64 + src;
      a[r] = ws.lo[dst[r]];
      b[r] = ws.hi[dst[r]];
#pragma unroll
      for (int k = 0; k <= D; ++k) qq[r][k] = ws.q[k * 64 + src];
    }
#pragma unroll 1
    for (int it = 0; it < kBis; ++it) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const double m = 0.5 * (a[r] + b[r]);
#if DR_K3_TASK_ESTRIN
        const double fx = estrin_eval<D>(qq[r], m);
#else
        double fx = qq[r][D];
#pragma unroll
        for (int k = D - 1; k >= 0; --k) fx = fx * m + qq[r][k];
#endif
        const bool left = (fx < 0) == neg[r], hit = fx == 0.0;   // a midpoint that IS the root closes the bracket on it
        a[r] = (left || hit) ? m : a[r];
        b[r] = (left && !hit) ? b[r] : m;
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) y[r] = 0.5 * (a[r] + b[r]);
#pragma unroll 1
    for (int it = 0; it < kNewt; ++it) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
#if DR_K3_TASK_ESTRIN
        double fx, dfx;
        estrin_eval_d<D>(qq[r], y[r], fx, dfx);
#else
        double fx = qq[r][D], dfx = 0;
#pragma unroll
        for (int k = D - 1; k >= 0; --k) {
          dfx = dfx * y[r] + fx;
          fx = fx * y[r] + qq[r][k];
        }
#endif
        const bool left = (fx < 0) == neg[r];
        a[r] = left ? y[r] : a[r];
        b[r] = left ? b[r] : y[r];
        const double yn = safeguarded_newton<double>(y[r], fx * __builtin_amdgcn_rcp(dfx), a[r], b[r]);
        y[r] = (fx == 0.0) ? y[r] : yn;
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
      if (val[r]) ws.lo[dst[r]] = y[r];
  }
}

// ------------------------------------------------------------------------------------------------
// Fallback of the Sturm isolation (round 5) for a lane whose chain lost a degree: the real roots of ONE polynomial in (-1, 1],
// found by the WHOLE WAVE.  (First version: the lane itself walked a rolled derivative chain -- correct, and a catastrophe: tens
// of thousands of dependent LDS reads, ~1 ms for the wave, and with one block per SIMD the launch ends when its slowest block
// does: 0.15 -> 0.62 ms on 131 072 RANSAC samples, of which 1e-4 take this path.)  Here the 64 lanes evaluate q and q' on a grid of
// 1024 cells (16 per lane, ~340 instructions); a cell over which q changes sign brackets a root; a cell over which only q'
// changes sign holds an extremum of q -- the wave bisects q' there, and the extremum either separates two roots (q has the other
// sign there), or is an even-multiplicity root (|q| at rounding level), or nothing.  Brackets are refined by the fixed schedule of
// the refine tasks.  No division by a leading coefficient, no assumption about the degree; ~2-5 us per polynomial.
// q: wave-uniform coefficients (max |q_i| = 1), ascending.  out (LDS, >= D doubles): the roots ascending; returns their number.
// lo_ws / hi_ws: LDS scratch of 64 doubles each, cnt_ws: 64 ints.
// ------------------------------------------------------------------------------------------------
template <int D>
__device__ __forceinline__ int wave_grid_roots(const double (&q)[D + 1], double *out, double *lo_ws, double *hi_ws, int *kind_ws, int lane) {
  auto ev = [&](double x, double &fx, double &dfx) {
    fx = q[D];
    dfx = 0;
#pragma unroll
    for (int i = D - 1; i >= 0; --i) {
      dfx = dfx * x + fx;
      fx = fx * x + q[i];
    }
  };
  constexpr int kCells = 16;
  const double h = 2.0 / (64 * kCells);
  const double x0 = -1.0 + (double)(kCells * lane) * h;
  // cells of this lane: sign change of q (kind 1), or of q' only with |q| small enough to matter (kind 2)
  unsigned ma = 0, mb = 0;
  double f0, d0;
  ev(x0, f0, d0);
#pragma unroll 1
  for (int i = 0; i < kCells; ++i) {
    const double x1 = (lane == 63 && i == kCells - 1) ? 1.0 : x0 + (double)(i + 1) * h;
    double f1, d1;
    ev(x1, f1, d1);
    const bool a = (f0 < 0) != (f1 < 0);
    // an extremum inside the cell moves q by at most |q'| h ~ (|d0| + |d1|) h from its end values (q'' changes it further only in
    // second order of h = 2e-3): cells whose end values are far above that cannot hide a root
    const bool bb = !a && ((d0 < 0) != (d1 < 0)) && fmin(fabs(f0), fabs(f1)) <= 4.0 * h * (fabs(d0) + fabs(d1)) + 1e-10;
    ma |= a ? (1u << i) : 0u;
    mb |= bb ? (1u << i) : 0u;
    f0 = f1;
    d0 = d1;
  }
  // task list in grid order (ascending x): exclusive prefix of the per-lane counts
  const int mine = __popc(ma | mb);
  int incl = mine;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int o = __shfl_up(incl, d, 64);
    incl += (lane >= d) ? o : 0;
  }
  const int total = min(__shfl(incl, 63, 64), 64);
  int pos = incl - mine;
  unsigned m = ma | mb;
  while (m) {
    const int i = __builtin_ctz(m);
    m &= m - 1;
    if (pos < 64) {
      lo_ws[pos] = x0 + (double)i * h;
      hi_ws[pos] = (lane == 63 && i == kCells - 1) ? 1.0 : x0 + (double)(i + 1) * h;
      kind_ws[pos] = ((ma >> i) & 1u) ? 1 : 2;
    }
    ++pos;
  }
  wave_lds_order();
  // one task per lane
  const bool has = lane < total;
  double lo = has ? lo_ws[lane] : 0.0, hi = has ? hi_ws[lane] : 0.0;
  const int kind = has ? kind_ws[lane] : 0;
  double r0 = 0, r1 = 0;
  int nr = 0;
  double flo, dlo;
  ev(lo, flo, dlo);
  double mid_ext = lo;
  bool two = false;
  if (__any(kind == 2)) {
    // extremum of q in the cell: bisection on q'
    double a = lo, b = hi;
    const bool dneg = dlo < 0;
#pragma unroll 1
    for (int it = 0; it < 28; ++it) {   // a cell is 2e-3 wide: 2^-28 of it is below the rounding of the extremum's position
      const double mm = 0.5 * (a + b);
      double fm, dm;
      ev(mm, fm, dm);
      const bool left = (dm < 0) == dneg;
      a = left ? mm : a;
      b = left ? b : mm;
    }
    mid_ext = 0.5 * (a + b);
    double fe, de;
    ev(mid_ext, fe, de);
    if (kind == 2) {
      two = (fe < 0) != (flo < 0) && fe != 0.0;           // the extremum lies on the other side: two simple roots around it
      if (!two && fabs(fe) <= 1e-11) { r0 = mid_ext; nr = 1; }   // q touches zero: a root of even multiplicity
    }
  }
  // brackets with a sign change: [lo, hi] (kind 1), or [lo, ext] and [ext, hi] (kind 2, two)
  auto refine = [&](double a, double b, bool want) -> double {
    double fa, da;
    ev(a, fa, da);
    const bool neg = fa < 0;
#pragma unroll 1
    for (int it = 0; it < 12; ++it) {
      const double mm = 0.5 * (a + b);
      double fm, dm;
      ev(mm, fm, dm);
      const bool left = (fm < 0) == neg, hit = fm == 0.0;
      a = (left || hit) ? mm : a;
      b = (left && !hit) ? b : mm;
    }
    double y = 0.5 * (a + b);
#pragma unroll 1
    for (int it = 0; it < 6; ++it) {
      double fx, dfx;
      ev(y, fx, dfx);
      const bool left = (fx < 0) == neg;
      a = left ? y : a;
      b = left ? b : y;
      const double yn = safeguarded_newton<double>(y, fx * __builtin_amdgcn_rcp(dfx), a, b);
      y = (fx == 0.0) ? y : yn;
    }
    (void)want;
    return y;
  };
  if (__any(kind == 1 || two)) {
    const double y = refine(lo, two ? mid_ext : hi, kind == 1 || two);
    if (kind == 1 || two) { r0 = y; nr = 1; }
  }
  if (__any(two)) {
    const double y = refine(mid_ext, hi, two);
    if (two) { r1 = y; nr = 2; }
  }
  // roots in ascending order: prefix of the per-task counts
  int inc2 = nr;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int o = __shfl_up(inc2, d, 64);
    inc2 += (lane >= d) ? o : 0;
  }
  const int nall = __shfl(inc2, 63, 64);
  const int p0 = inc2 - nr;
  if (nr >= 1 && p0 < D) out[p0] = r0;
  if (nr >= 2 && p0 + 1 < D) out[p0 + 1] = r1;
  wave_lds_order();
  return min(nall, D);
}

// wave-cooperative (block = one wave): every lane must call it.  Same outputs as real_roots_half_wave.
template <int D, int kBisLast = DR_ROOT_BIS_LAST, int kNewtLast = DR_ROOT_NEWT_LAST>
__device__ __forceinline__ void real_roots_half_sturm(const double (&c)[D + 1], bool outer, double (&roots)[D], int &count, double *lds_ws,
                                                      int lane) {
  const SturmWs<D> ws(lds_ws);
#ifdef DR_PROFILE_STAGES
  unsigned long long _st0 = __builtin_readcyclecounter();
  int _iters = 0;
#endif
  double cmax = 0;
#pragma unroll
  for (int i = 0; i <= D; ++i) cmax = fmax(cmax, fabs(c[i]));
  const bool ok = is_finite(cmax) && cmax > 0;
  const double sc = ok ? frcp(cmax) : 0.0;
  // chain F[k] = polynomial of degree D - k, ascending coefficients F[k][0 .. D - k]
  double F[D + 1][D + 1];
#pragma unroll
  for (int i = 0; i <= D; ++i) {
    const double a = ok ? c[i] * sc : (i == 0 ? 1.0 : 0.0);
    const double b = ok ? c[D - i] * sc : (i == 0 ? 1.0 : 0.0);
    F[0][i] = outer ? b : a;
    ws.q[i * 64 + lane] = F[0][i];
  }
  auto renorm = [](double (&f)[D + 1], int deg) {   // positive scale only: an approximate reciprocal is enough
    double mx = 0;
#pragma unroll
    for (int i = 0; i <= D; ++i)
      if (i <= deg) mx = fmax(mx, fabs(f[i]));
    const double s = (mx > 0 && is_finite(mx)) ? (double)__builtin_amdgcn_rcpf((float)mx) : 1.0;
#pragma unroll
    for (int i = 0; i <= D; ++i)
      if (i <= deg) f[i] *= s;
  };
#pragma unroll
  for (int i = 0; i < D; ++i) F[1][i] = F[0][i + 1] * (double)(i + 1);
  renorm(F[1], D - 1);
  // Round 5 (review of rounds 3-4): the static-degree pseudo-remainder below divides -- in effect -- by the leading coefficient b of
  // every chain member.  Where b is at rounding level against the member's largest coefficient (1 after renorm) the member has
  // lost a degree (p itself of lower degree: a vanishing leading coefficient; or a common factor of p and p': a multiple root) and
  // the signs that follow are noise: the counts come out inconsistent and the interval logic below DROPS roots.  Such a lane
  // (`degenerate`) takes its polynomial through the derivative chain afterwards (real_roots_half_wave: no division, no degree
  // assumption), with the critical points at which p vanishes reported as (even-multiplicity) roots.
  bool degenerate = false;
#pragma unroll
  for (int k = 1; k < D; ++k) {
    // A = F[k-1] (degree n), B = F[k] (degree n - 1):  F[k+1] = -(b^2 A - (a_n b x + (a_{n-1} b - a_n b_{n-2})) B), degree n - 2
    const int n = D - k + 1;
    const double an = F[k - 1][n], an1 = F[k - 1][n - 1], b = F[k][n - 1], b2 = (n >= 2) ? F[k][n - 2] : 0.0;
    degenerate = degenerate || !(fabs(b) > 1e-11);
#ifdef DR_PROFILE_STAGES
    if (D == 10 && !(fabs(b) > 1e-11)) atomicAdd(&::dr::g_stage_cycles[23], 1ull);
#endif
    double rmax = 0;   // largest coefficient of the raw remainder (inputs: max 1): at rounding level => p and p' share a factor
    const double q1 = an * b, q0 = an1 * b - an * b2, bb = b * b;
#pragma unroll
    for (int i = 0; i <= D; ++i) {
      if (i <= n - 2) {
        double r = q0 * F[k][i] - bb * F[k - 1][i];
        if (i >= 1) r += q1 * F[k][i - 1];
        F[k + 1][i] = r;
        rmax = fmax(rmax, fabs(r));
      }
    }
#ifdef DR_PROFILE_STAGES
    if (D == 10 && fabs(b) > 1e-11 && !(rmax > 1e-12 * (bb + fabs(q0) + fabs(q1)))) atomicAdd(&::dr::g_stage_cycles[24], 1ull);
#endif
    degenerate = degenerate || !(rmax > 1e-12 * (bb + fabs(q0) + fabs(q1)));
    renorm(F[k + 1], n - 2);
  }
  // sign variations of the chain at x (bits 0-3) and the sign of p(x) (bit 4)
  auto variations = [&](double x) -> unsigned {
    // all Horner chains advance together (step i touches every polynomial that still has a coefficient left): eleven independent
    // dependency chains instead of one after the other
#if DR_K3_VAR_ASM
    if constexpr (D == 10) {
      // round 6: the 55 FMAs as one hand-scheduled asm block (csrc/sturm_eval_asm.hpp), the sign bits by v_alignbit_b32
      const unsigned w = sturm_signs10(F, x);
      const unsigned ch = (w ^ (w >> 1)) & ((1u << D) - 1u);
      return (unsigned)__popc(ch) | ((w & 1u) << 4);
    }
#endif
    double val[D + 1];
#pragma unroll
    for (int k = 0; k <= D; ++k) val[k] = F[k][D - k];
#pragma unroll
    for (int i = 1; i <= D; ++i) {
#pragma unroll
      for (int k = 0; k + i <= D; ++k) val[k] = val[k] * x + F[k][D - k - i];
#if DR_K3_VAR_SCHED
      // round 6: the scheduler, short of registers, otherwise runs the chains one after the other (a dependent v_fma_f64 every
      // ~8.5 clocks at one wave per SIMD instead of an independent one every ~4.3): nothing moves across a Horner level
      __builtin_amdgcn_sched_barrier(0);
#endif
    }
    unsigned w = 0;
#pragma unroll
    for (int k = 0; k <= D; ++k) w |= ((unsigned)__double2hiint(val[k]) >> 31) << k;
    const unsigned ch = (w ^ (w >> 1)) & ((1u << D) - 1u);
    return (unsigned)__popc(ch) | ((w & 1u) << 4);
  };
  const unsigned vm = variations(-1.0), vp = variations(1.0);
#ifdef DR_PROFILE_STAGES
  if (D == 10 && lane == 0) atomicAdd(&::dr::g_stage_cycles[6], __builtin_readcyclecounter() - _st0);
  _st0 = __builtin_readcyclecounter();
#endif
  // entries [0, nout) = isolated intervals in ascending order; entries [top, D) = pending, the leftmost on top; the interval a
  // lane is working on stays in registers (LDS is read only when an interval is popped: once per root, not once per step)
  int nout = 0, top = D;
  bool have = ok && (int)(vm & 15u) - (int)(vp & 15u) >= 1;
  double l = -1.0, h = 1.0;
  unsigned v = vm | (vp << 8);
  auto pop = [&]() {
    have = top < D;
    const int tc = have ? top : D - 1;
    l = ws.lo[tc * 64 + lane];
    h = ws.hi[tc * 64 + lane];
    v = ws.vv[tc * 64 + lane];
    top += have ? 1 : 0;
  };
#if DR_K3_ISO_FLAT
  // Round 5: the step without control flow.  Rounds 3-4 walked an if / else-if chain (isolated -> emit + pop | cannot split -> pop |
  // left part isolated -> emit, go right | left part holds roots -> push right, go left | go right); with 64 lanes in different
  // states the wave executed every arm one after the other, ~250 instructions per step where the evaluation of the chain is ~100.
  // Here a lane derives its action as predicates, writes at most ONE record (the emitted interval or the pushed right part),
  // updates its interval with selects and pops under one mask.  Same intervals, same order: the roots are bit-identical.
#pragma unroll 1
  for (int guard = 0; guard < 64 * D; ++guard) {
    if (!__any(have)) break;
#ifdef DR_PROFILE_STAGES
    ++_iters;
#endif
    const int cl = (int)(v & 15u), ch = (int)((v >> 8) & 15u);
    const bool iso = have && (cl - ch == 1);
    // split a hair off the centre: a root AT a split point would be counted with the sign of +0 (nice inputs have nice roots:
    // 0, 1/2, 1/4 ... are exactly where plain halving of (-1, 1] looks)
    const double mid = __builtin_fma(h - l, 0.49999952316284180, l);
    const unsigned vmid = variations(mid) & 31u;   // (a lane that emits or idles in this step evaluates for nothing: the wave pays anyway)
    const int cm = (int)(vmid & 15u);
    const int nl = cl - cm, nr = cm - ch;
    // an interval that cannot be split any more (a multiple root to rounding) or whose halves both come out empty
    // (inconsistent counts of a degenerate chain) is dropped
    const bool splittable = mid > l && mid < h && (h - l) > 1e-12;
    const bool room = top - 2 >= nout;
    const bool split = have && !iso;
    const bool drop = split && (!splittable || (nl < 1 && nr < 1));
    const bool go = split && !drop;
    const bool emit_left = go && nl == 1 && nr >= 1 && room;   // the left part is isolated: straight to the output list, go on with the right
    const bool to_left = go && !emit_left && nl >= 1;          // the left part holds roots: it is next, the right part waits on the stack
    const bool push_right = to_left && nr >= 1 && room;        // (dropped if there is no room: degenerate counts)
    const bool to_right = go && !emit_left && !to_left;
    const bool emit = iso || emit_left;
    if (emit || push_right) {
      const int we = emit ? nout : top - 1;
      ws.lo[we * 64 + lane] = push_right ? mid : l;
      ws.hi[we * 64 + lane] = emit_left ? mid : h;
      ws.vv[we * 64 + lane] = iso ? v : (emit_left ? ((v & 31u) | (vmid << 8)) : (vmid | (v & 0xff00u)));
    }
    nout += emit ? 1 : 0;
    top -= push_right ? 1 : 0;
    const bool move_l = emit_left || to_right;
    l = move_l ? mid : l;
    h = to_left ? mid : h;
    v = move_l ? (vmid | (v & 0xff00u)) : (to_left ? ((v & 31u) | (vmid << 8)) : v);
    if (iso || drop) pop();
  }
#else
#pragma unroll 1
  for (int guard = 0; guard < 64 * D; ++guard) {
    if (!__any(have)) break;
#ifdef DR_PROFILE_STAGES
    ++_iters;
#endif
#if DR_K3_STURM_DRAIN
    // isolated intervals go to the output list and the next pending one comes off the stack (LDS only) WITHOUT costing the wave a
    // step: rounds 3-4 spent one step of the loop -- i.e. one evaluation of the chain by every other lane -- per emitted interval
    // (4-5 of the ~13 steps the slowest lane of a wave needs); same intervals in the same order, so the roots are bit-identical
    while (have && ((int)(v & 15u) - (int)((v >> 8) & 15u) == 1)) {
      ws.lo[nout * 64 + lane] = l;
      ws.hi[nout * 64 + lane] = h;
      ws.vv[nout * 64 + lane] = v;
      ++nout;
      pop();
    }
    if (!__any(have)) break;
    const bool iso = false;
#else
    // one action per lane and step: an isolated interval goes to the output list and the next pending one comes off the stack
    // (LDS only), any other interval is split at its midpoint (one evaluation of the chain)
    const bool iso = have && ((int)(v & 15u) - (int)((v >> 8) & 15u) == 1);
#endif
    if (iso) {
      ws.lo[nout * 64 + lane] = l;
      ws.hi[nout * 64 + lane] = h;
      ws.vv[nout * 64 + lane] = v;
      ++nout;
      pop();
    } else if (have) {
      // split a hair off the centre: a root AT a split point would be counted with the sign of +0 (nice inputs have nice roots:
      // 0, 1/2, 1/4 ... are exactly where plain halving of (-1, 1] looks)
      const double mid = __builtin_fma(h - l, 0.49999952316284180, l);
      const unsigned vmid = variations(mid);
      const int nl = (int)(v & 15u) - (int)(vmid & 15u), nr = (int)(vmid & 15u) - (int)((v >> 8) & 15u);
      // an interval that cannot be split any more (a multiple root to rounding) or whose halves both come out empty
      // (inconsistent counts of a degenerate chain) is dropped
      const bool splittable = mid > l && mid < h && (h - l) > 1e-12;
      if (!splittable || (nl < 1 && nr < 1)) {
        pop();
      } else if (nl == 1 && nr >= 1 && top - 2 >= nout) {
        // the left part is isolated: straight to the output list (it is the leftmost interval of this lane), go on with the right
        ws.lo[nout * 64 + lane] = l;
        ws.hi[nout * 64 + lane] = mid;
        ws.vv[nout * 64 + lane] = (v & 31u) | ((vmid & 31u) << 8);
        ++nout;
        l = mid;
        v = (vmid & 31u) | (v & 0xff00u);
      } else if (nl >= 1) {
        if (nr >= 1 && top - 2 >= nout) {   // right part waits on the stack (dropped if there is no room: degenerate counts)
          --top;
          ws.lo[top * 64 + lane] = mid;
          ws.hi[top * 64 + lane] = h;
          ws.vv[top * 64 + lane] = (vmid & 31u) | (v & 0xff00u);
        }
        h = mid;
        v = (v & 31u) | ((vmid & 31u) << 8);
      } else {
        l = mid;
        v = (vmid & 31u) | (v & 0xff00u);
      }
    }
  }
#endif
  wave_lds_order();
#ifdef DR_PROFILE_STAGES
  if (D == 10 && lane == 0) { atomicAdd(&::dr::g_stage_cycles[7], __builtin_readcyclecounter() - _st0); atomicAdd(&::dr::g_stage_cycles[9], (unsigned long long)_iters); atomicMax(&::dr::g_stage_cycles[10], (unsigned long long)_iters); }
  _st0 = __builtin_readcyclecounter();
#endif
  // tasks: the isolated intervals whose ends differ in the sign of p
  unsigned has_mask = 0;
  int offs = 0;
#pragma unroll
  for (int e = 0; e < D; ++e) {
    const unsigned v = ws.vv[(e < nout ? e : 0) * 64 + lane];
    const bool h = e < nout && (((v >> 4) ^ (v >> 12)) & 1u);
    const unsigned long long bm = __ballot(h);
    if (bm) {
      const int pos = offs + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bm, 0u));
      if (h) {
        has_mask |= 1u << e;
        ws.queue[pos] = (uint16_t)(lane | (e << 6) | (((v >> 4) & 1u) << 10));
      }
      offs += __popcll(bm);
    }
  }
  wave_lds_order();
  if (offs <= 64) sturm_tasks<D, 1>(ws, offs, lane, kBisLast, kNewtLast);
  else if (offs <= 128) sturm_tasks<D, 2>(ws, offs, lane, kBisLast, kNewtLast);
  else sturm_tasks<D, 3>(ws, offs, lane, kBisLast, kNewtLast);
  wave_lds_order();
#ifdef DR_PROFILE_STAGES
  if (D == 10 && lane == 0) atomicAdd(&::dr::g_stage_cycles[8], __builtin_readcyclecounter() - _st0);
#endif
  // dense list: the roots that count are written back over the lane's own entries (entry `count` <= k: already read) and read again
  // with static indices -- rounds 3-4 compacted in registers with a 10 x 10 select network (300 instructions)
  count = 0;
#pragma unroll
  for (int k = 0; k < D; ++k) {
    const double xk = ws.lo[k * 64 + lane];
    const bool take = ((has_mask >> k) & 1u) && (!outer || (fabs(xk) > 1e-9 && fabs(xk) < 1.0));
    if (__any(take)) {
      const double vv_ = outer ? frcp(xk) : xk;
      if (take) ws.lo[count * 64 + lane] = vv_;
    }
    count += take ? 1 : 0;
  }
#pragma unroll
  for (int i = 0; i < D; ++i) {
    const double r = ws.lo[i * 64 + lane];
    roots[i] = (i < count) ? r : 0.0;
  }
#if DR_K3_STURM_FALLBACK
  // (1 + z^D is the polynomial the callers hand to lanes WITHOUT a sample -- partial blocks, few samples per block on small grids,
  // rank-deficient systems: its chain loses eight degrees at once and it has no real root: nothing to look for.  Forgetting this
  // sent every idle lane of a one-pair call through the fallback: 30 -> 189 us.)
  bool dummy = true;
#pragma unroll
  for (int i = 1; i < D; ++i) dummy = dummy && c[i] == 0.0;
  degenerate = degenerate && ok && !dummy;
#ifdef DR_PROFILE_STAGES
  if (D == 10) { if (degenerate) atomicAdd(&::dr::g_stage_cycles[21], 1ull); if (lane == 0 && __any(degenerate)) atomicAdd(&::dr::g_stage_cycles[22], 1ull); }
#endif
  unsigned long long todo = __ballot(degenerate);
  if (todo) {
    // rare (1e-4 of the lanes on RANSAC samples): one polynomial at a time, by the whole wave (wave_grid_roots)
    wave_lds_order();
    double *gout = lds_ws, *glo = lds_ws + 64, *ghi = lds_ws + 128;
    int *gkind = reinterpret_cast<int *>(lds_ws + 192);
    static_assert(192 + 32 <= SturmWs<D>::kDoubles, "the fallback works in the isolation's own workspace");
    while (todo) {
      const int src = __builtin_ctzll(todo);
      todo &= todo - 1;
      double qq[D + 1];
#pragma unroll
      for (int i = 0; i <= D; ++i) qq[i] = __shfl(F[0][i], src, 64);   // the lane's normalised (and, for its outer half, reversed) polynomial
      const int nfound = wave_grid_roots<D>(qq, gout, glo, ghi, gkind, lane);
      const bool src_outer = __shfl((int)outer, src, 64) != 0;
      if (lane == src) {
        count = 0;
#pragma unroll
        for (int i = 0; i < D; ++i) roots[i] = 0.0;
      }
#pragma unroll 1
      for (int kk = 0; kk < nfound; ++kk) {
        const double xk = gout[kk];
        const bool take = !src_outer || (fabs(xk) > 1e-9 && fabs(xk) < 1.0);
        const double vv_ = src_outer ? 1.0 / xk : xk;
        if (lane == src && take) {
#pragma unroll
          for (int t = 0; t < D; ++t) roots[t] = (t == count) ? vv_ : roots[t];
          ++count;
        }
      }
      wave_lds_order();
    }
  }
#endif
}

// roots[0..count-1] = all real roots found (|z| <= 1 ascending first, then the |z| > 1 ones); count <= D
template <int D, int kBisLast = 10, int kNewtLast = 6>
__device__ __forceinline__ void real_roots(const double (&c)[D + 1], double (&roots)[D], int &count, double tail_tol = 0.0) {
  double cmax = 0;
#pragma unroll
  for (int i = 0; i <= D; ++i) cmax = fmax(cmax, fabs(c[i]));
  const bool ok = is_finite(cmax) && cmax > 0;
  double cn[D + 1], cr[D + 1];
  const double sc = ok ? 1.0 / cmax : 0.0;
#pragma unroll
  for (int i = 0; i <= D; ++i) {
    cn[i] = ok ? c[i] * sc : (i == 0 ? 1.0 : 0.0);
    cr[D - i] = cn[i];
  }
  double xin[D], xout[D];
  unsigned min_, mout;
  roots_in_unit<D, kBisLast, kNewtLast>(cn, xin, min_, tail_tol);
  roots_in_unit<D, kBisLast, kNewtLast>(cr, xout, mout, tail_tol);
  if (!ok) { min_ = 0; mout = 0; }
  // per-lane compaction into a dense list (static select network: no dynamic register indexing)
  count = 0;
#pragma unroll
  for (int i = 0; i < D; ++i) roots[i] = 0.0;
#pragma unroll
  for (int i = 0; i < 2 * D; ++i) {
    const bool inner = i < D;
    const int k = inner ? i : i - D;
    const double v = inner ? xin[k] : 1.0 / xout[k];
    // a reversed-polynomial root w with |w| ~ 0 is a root at infinity (vanishing leading coefficient); |w| = 1 is
    // already covered by the inner search
    const bool take = inner ? ((min_ >> k) & 1u) : (((mout >> k) & 1u) && fabs(xout[k]) > 1e-9 && fabs(xout[k]) < 1.0);
    if (take && count < D) {
#pragma unroll
      for (int t = 0; t < D; ++t) roots[t] = (t == count) ? v : roots[t];
    }
    count += (take && count < D) ? 1 : 0;
  }
}

// ------------------------------------------------------------------------------------------------
// Cyclic Jacobi eigen-decomposition of a symmetric n x n matrix held in LDS (A: n*n, V: n*n).
// On return A's diagonal holds the eigenvalues and the columns of V the eigenvectors.
// ------------------------------------------------------------------------------------------------
template <int N>
__device__ __forceinline__ void jacobi_eig_lds(const LaneWs &A, const LaneWs &V) {
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < N; ++j) V[i * N + j] = (i == j) ? 1.0 : 0.0;
  double prev_off = INFINITY;
  for (int sweep = 0; sweep < 30; ++sweep) {
    double off = 0, dg = 0;
    for (int i = 0; i < N; ++i) {
      dg += A[i * N + i] * A[i * N + i];
      for (int j = i + 1; j < N; ++j) off += A[i * N + j] * A[i * N + j];
    }
    // converged, or stagnating at the rounding floor of a badly scaled matrix (1e-34 was never reached by Gram matrices
    // with a 1e6 eigenvalue spread: all 30 sweeps ran)
    const bool done = !(off > 1e-30 * dg) || (sweep >= 4 && off > 0.25 * prev_off);
    prev_off = off;
    if (__all(done)) break;
    for (int p = 0; p < N - 1; ++p) {
      for (int q = p + 1; q < N; ++q) {
        const double apq = A[p * N + q];
        const double app = A[p * N + p], aqq = A[q * N + q];
        double c = 1.0, s = 0.0;
        if (fabs(apq) > 1e-300 && !done) {
          const double theta = (aqq - app) / (2.0 * apq);
          const double t = dsign(1.0, theta) / (fabs(theta) + sqrt(theta * theta + 1.0));
          c = 1.0 / sqrt(t * t + 1.0);
          s = t * c;
        }
        for (int k = 0; k < N; ++k) {
          const double akp = A[k * N + p], akq = A[k * N + q];
          A[k * N + p] = c * akp - s * akq;
          A[k * N + q] = s * akp + c * akq;
        }
        for (int k = 0; k < N; ++k) {
          const double apk = A[p * N + k], aqk = A[q * N + k];
          A[p * N + k] = c * apk - s * aqk;
          A[q * N + k] = s * apk + c * aqk;
        }
        for (int k = 0; k < N; ++k) {
          const double vkp = V[k * N + p], vkq = V[k * N + q];
          V[k * N + p] = c * vkp - s * vkq;
          V[k * N + q] = s * vkp + c * vkq;
        }
      }
    }
  }
}

// Cyclic Jacobi of a symmetric 9x9 matrix entirely in VGPRs (162 doubles; for kernels that run one wave per SIMD).
// The (p,q) schedule is static, so nothing needs run-time indexing; used where a single wave would otherwise crawl
// through LDS latency (K7 refit: one sample per block).  column(V, c) extraction is a 9-way select.
__device__ __forceinline__ void jacobi_eig9_reg(double (&A)[9][9], double (&V)[9][9]) {
#pragma unroll
  for (int i = 0; i < 9; ++i)
#pragma unroll
    for (int j = 0; j < 9; ++j) V[i][j] = (i == j) ? 1.0 : 0.0;
  double prev_off = INFINITY;
#pragma unroll 1
  for (int sweep = 0; sweep < 30; ++sweep) {
    double off = 0, dg = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      dg += A[i][i] * A[i][i];
#pragma unroll
      for (int j = i + 1; j < 9; ++j) off += A[i][j] * A[i][j];
    }
    const bool done = !(off > 1e-30 * dg) || (sweep >= 4 && off > 0.25 * prev_off);
    prev_off = off;
    if (__all(done)) break;
#pragma unroll
    for (int p = 0; p < 8; ++p) {
#pragma unroll
      for (int q = p + 1; q < 9; ++q) {
        const double apq = A[p][q];
        double c = 1.0, s = 0.0;
        if (fabs(apq) > 1e-300 && !done) {
          const double theta = (A[q][q] - A[p][p]) / (2.0 * apq);
          const double t = dsign(1.0, theta) / (fabs(theta) + sqrt(theta * theta + 1.0));
          c = 1.0 / sqrt(t * t + 1.0);
          s = t * c;
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) {
          const double akp = A[k][p], akq = A[k][q];
          A[k][p] = c * akp - s * akq;
          A[k][q] = s * akp + c * akq;
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) {
          const double apk = A[p][k], aqk = A[q][k];
          A[p][k] = c * apk - s * aqk;
          A[q][k] = s * apk + c * aqk;
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) {
          const double vkp = V[k][p], vkq = V[k][q];
          V[k][p] = c * vkp - s * vkq;
          V[k][q] = s * vkp + c * vkq;
        }
      }
    }
  }
}

// index of the smallest diagonal entry not yet in `used` (bit mask), and that eigenvector
__device__ __forceinline__ int smallest_eigvec9(const double (&A)[9][9], const double (&V)[9][9], unsigned &used,
                                                double (&out)[9]) {
  int best = 0;
  double bv = INFINITY;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const bool free_ = !((used >> i) & 1u);
    if (free_ && A[i][i] < bv) { bv = A[i][i]; best = i; }
  }
  used |= 1u << best;
#pragma unroll
  for (int r = 0; r < 9; ++r) {
    double v = V[r][0];
#pragma unroll
    for (int c = 1; c < 9; ++c) v = (c == best) ? V[r][c] : v;
    out[r] = v;
  }
  return best;
}

// Parallel cyclic Jacobi for ONE symmetric 9x9 matrix held in LDS (A, V: 81 doubles each, row-major), executed by one
// whole wave: round-robin ordering -- 9 rounds per sweep, 4 disjoint rotations per round (player r sits out in round r)
// -- lanes 0-3 compute the four (c, s), then 36 lanes apply the column updates of A, 36 those of V, 36 the row updates.
// ~700 cycles per round instead of ~27 k cycles per sweep of the serial register version above; on return the diagonal
// of A holds the eigenvalues and the columns of V the eigenvectors.  `cs` = 16 doubles of LDS scratch.
// f64 reciprocal square root / reciprocal from the hardware estimates (v_rsq_f64 / v_rcp_f64, ~26 bits) + two Newton steps: what the
// rotation angles below need -- finite, well-scaled, positive arguments; none of the scaling / special-case code of sqrt() and "/"
// Round 5: one parallel-order round (four disjoint rotations; index r rests) is TWO phases instead of four: (0) lanes 0..8 work
// out the rotation of the pair their index belongs to -- partner i' = (2 r - i) mod 9 -- as (C[i], S[i]) with the sign of "p or q"
// folded into S, by rsq / rcp + Newton instead of three divisions and two square roots; (1) every lane forms its elements of
// J^T A J (45 lanes: the upper triangle, mirrored on the way out) and of V J directly from the OLD matrices:
//   A'[i][j] = C[j] (C[i] A[i][j] + S[i] A[i'][j]) + S[j] (C[i] A[i][j'] + S[i] A[i'][j']),   V'[k][j] = C[j] V[k][j] + S[j] V[k][j'].
// (was: rotation by four lanes, columns of A and V, rows of A -- ~160 issued instructions and four LDS round trips per round
// against ~70 and three; the refit kernel runs next to the sampler and the solver of the same call and every instruction it
// issues there waits for the vector ALU)
__device__ __forceinline__ void jacobi_eig9_wave(double *A, double *V, double *cs, int lane) {
  double *C = cs, *S = cs + 9;
  for (int i = lane; i < 81; i += 64) V[i] = (i % 10 == 0) ? 1.0 : 0.0;
  // this lane's element of the upper triangle (lanes 0..44): entry q of the row-major upper triangle -> (ui, uj)
  int ui = 0, uj = 0;
  {
    int rem = lane < 45 ? lane : 0;
    while (rem >= 9 - ui) { rem -= 9 - ui; ++ui; }
    uj = ui + rem;
  }
  const int f1 = lane + 64;                         // second V element of lanes 0..16
  const int vk0 = lane / 9, vj0 = lane % 9, vk1 = f1 / 9, vj1 = f1 % 9;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
  double prev_off = INFINITY;
#pragma unroll 1
  for (int sweep = 0; sweep < 30; ++sweep) {
    double off = 0, dg = 0;
    for (int i = lane; i < 81; i += 64) {
      const double v = A[i] * A[i];
      if (i % 10 == 0) dg += v; else off += v;
    }
    off = wave_sum(off);
    dg = wave_sum(dg);
    // converged (off-diagonal mass at rounding level), or stagnating at the rounding floor of a badly scaled matrix
    if (!(off > 2e-30 * dg) || (sweep >= 4 && off > 0.25 * prev_off)) break;
    prev_off = off;
#pragma unroll 1
    for (int r = 0; r < 9; ++r) {
      if (lane < 9) {
        const int i = lane, ip = (2 * r - i + 18) % 9;
        double c = 1.0, ss = 0.0;
        if (ip != i) {
          const int p = min(i, ip), q = max(i, ip);
          const double apq = A[p * 9 + q];
          if (fabs(apq) > 1e-300) {
            const double a = A[q * 9 + q] - A[p * 9 + p], b2 = 2.0 * apq;
            const double r2 = fma(a, a, b2 * b2);
            const double rr = r2 * rsqrt_nr(r2);                       // sqrt(a^2 + b^2)
            const double t = (a >= 0 ? b2 : -b2) * rcp_nr(fabs(a) + rr);   // tan(phi) = sgn(theta) / (|theta| + sqrt(theta^2 + 1))
            c = rsqrt_nr(fma(t, t, 1.0));
            const double sn = t * c;
            ss = (i < ip) ? -sn : sn;                                  // column p: c col_p - s col_q ; column q: s col_p + c col_q
          }
        }
        C[i] = c;
        S[i] = ss;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
      __builtin_amdgcn_wave_barrier();
      double na = 0.0, nv0 = 0.0, nv1 = 0.0;
      if (lane < 45) {
        const int i = ui, j = uj, ip = (2 * r - i + 18) % 9, jp = (2 * r - j + 18) % 9;
        const double ci = C[i], si = S[i], cj = C[j], sj = S[j];
        const double rij = fma(si, A[ip * 9 + j], ci * A[i * 9 + j]);
        const double rijp = fma(si, A[ip * 9 + jp], ci * A[i * 9 + jp]);
        na = fma(sj, rijp, cj * rij);
      }
      {
        const int jp0 = (2 * r - vj0 + 18) % 9;
        nv0 = fma(S[vj0], V[vk0 * 9 + jp0], C[vj0] * V[vk0 * 9 + vj0]);
        if (f1 < 81) {
          const int jp1 = (2 * r - vj1 + 18) % 9;
          nv1 = fma(S[vj1], V[vk1 * 9 + jp1], C[vj1] * V[vk1 * 9 + vj1]);
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
      __builtin_amdgcn_wave_barrier();
      if (lane < 45) {
        A[ui * 9 + uj] = na;
        A[uj * 9 + ui] = na;
      }
      V[lane] = nv0;
      if (f1 < 81) V[f1] = nv1;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
      __builtin_amdgcn_wave_barrier();
    }
  }
}

// the `count` smallest eigenpairs after jacobi_eig9_wave: out[t] = eigenvector of the (t+1)-th smallest eigenvalue
template <int kCount>
__device__ __forceinline__ void smallest_eigvecs9_lds(const double *A, const double *V, double (&out)[kCount][9]) {
  double d[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) d[i] = A[i * 10];
  unsigned used = 0;
#pragma unroll
  for (int t = 0; t < kCount; ++t) {
    int best = 0;
    double bv = INFINITY;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      const bool free_ = !((used >> i) & 1u);
      if (free_ && d[i] < bv) { bv = d[i]; best = i; }
    }
    used |= 1u << best;
#pragma unroll
    for (int r = 0; r < 9; ++r) out[t][r] = V[r * 9 + best];
  }
}

// Eigenvector of the SMALLEST eigenvalue of a symmetric positive semi-definite 9x9 matrix by inverse iteration on a
// Cholesky factor (A + 1e-14 tr(A) I, so that an exactly singular A -- noise-free data -- still factors): ~200 flops
// for the factor + 162 per iteration, against ~29 k for the cyclic Jacobi above, which it replaces where only this one
// vector is needed (LSQ fundamental matrix).  Error after m iterations ~ (lambda_9 / lambda_8)^m; 24 iterations.
__device__ __forceinline__ void smallest_eigvec9_invit(const double (&A)[9][9], double (&x)[9]) {
  double L[9][9];
  double tr = 0;
#pragma unroll
  for (int i = 0; i < 9; ++i) tr += A[i][i];
  const double shift = 1e-14 * tr;
  double inv[9];
#pragma unroll
  for (int j = 0; j < 9; ++j) {
    double d = A[j][j] + shift;
#pragma unroll
    for (int k = 0; k < j; ++k) d -= L[j][k] * L[j][k];
    d = fmax(d, 1e-300);
    const double r = 1.0 / sqrt(d);
    inv[j] = r;
#pragma unroll
    for (int i = j + 1; i < 9; ++i) {
      double v = A[i][j];
#pragma unroll
      for (int k = 0; k < j; ++k) v -= L[i][k] * L[j][k];
      L[i][j] = v * r;
    }
  }
#pragma unroll
  for (int i = 0; i < 9; ++i) x[i] = 1.0 / 3.0 + 0.01 * i;   // fixed start, not orthogonal to anything in particular
#pragma unroll 1
  for (int it = 0; it < 24; ++it) {
    double y[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) {        // L y = x
      double v = x[i];
#pragma unroll
      for (int k = 0; k < i; ++k) v -= L[i][k] * y[k];
      y[i] = v * inv[i];
    }
    double nn = 0;
#pragma unroll
    for (int i = 8; i >= 0; --i) {       // L^T z = y
      double v = y[i];
#pragma unroll
      for (int k = i + 1; k < 9; ++k) v -= L[k][i] * x[k];
      x[i] = v * inv[i];
      nn += x[i] * x[i];
    }
    const double sc = 1.0 / sqrt(nn);
#pragma unroll
    for (int i = 0; i < 9; ++i) x[i] *= sc;
  }
}

// symmetric 3x3 Jacobi in registers; eigenvalues in d[], eigenvectors = columns of V
#ifndef DR_JACOBI3_FAST
#define DR_JACOBI3_FAST 1
#endif
__device__ __forceinline__ void jacobi_eig3(double (&A)[3][3], double (&V)[3][3], double (&d)[3]) {
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) V[i][j] = (i == j) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 24; ++sweep) {
    const double off = A[0][1] * A[0][1] + A[0][2] * A[0][2] + A[1][2] * A[1][2];
    const double dg = A[0][0] * A[0][0] + A[1][1] * A[1][1] + A[2][2] * A[2][2];
    const bool done = !(off > 1e-36 * dg);
    if (__all(done)) break;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
#pragma unroll
      for (int q = p + 1; q < 3; ++q) {
        const double apq = A[p][q];
        double c = 1.0, s = 0.0;
        if (fabs(apq) > 1e-300 && !done) {
#if DR_JACOBI3_FAST
          // round 6: t = sgn(theta) / (|theta| + sqrt(theta^2 + 1)) with theta = h / (2 apq) is sgn(h) 2 apq / (|h| + sqrt(h^2 + 4 apq^2)):
          // one reciprocal square root for the root, one reciprocal, one reciprocal square root for c (rsq / rcp + two Newton steps
          // each) instead of three IEEE divisions and two square roots -- the rotation is a dependent chain one lane waits for
          const double h = A[q][q] - A[p][p];
          const double x = __builtin_fma(h, h, 4.0 * apq * apq);
          if (x > 1e-280) {
            const double root = x * rsqrt_nr(x);
            const double t = (h >= 0 ? 2.0 * apq : -2.0 * apq) * rcp_nr(fabs(h) + root);   // sgn(h) with sgn(0) = +1, as dsign(1, theta)
            c = rsqrt_nr(__builtin_fma(t, t, 1.0));
            s = t * c;
          }
#else
          const double theta = (A[q][q] - A[p][p]) / (2.0 * apq);
          const double t = dsign(1.0, theta) / (fabs(theta) + sqrt(theta * theta + 1.0));
          c = 1.0 / sqrt(t * t + 1.0);
          s = t * c;
#endif
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const double akp = A[k][p], akq = A[k][q];
          A[k][p] = c * akp - s * akq;
          A[k][q] = s * akp + c * akq;
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const double apk = A[p][k], aqk = A[q][k];
          A[p][k] = c * apk - s * aqk;
          A[q][k] = s * apk + c * aqk;
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const double vkp = V[k][p], vkq = V[k][q];
          V[k][p] = c * vkp - s * vkq;
          V[k][q] = s * vkp + c * vkq;
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) d[i] = A[i][i];
}

// ---- multivariate polynomial tables (x, y, z), shared by the two five-point solvers -----------------
struct Mono { int x, y, z; };
__host__ __device__ constexpr bool mono_eq(Mono a, Mono b) { return a.x == b.x && a.y == b.y && a.z == b.z; }

struct Tables {
  // degree-1 order (x, y, z, 1)
  Mono e1[4] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}, {0, 0, 0}};
  Mono e2[10];
  Mono e3[20];
  int t11[4][4] = {};   // product of two degree-1 monomials -> index into e2
  int t21[10][4] = {};  // degree-2 x degree-1 -> index into e3
  constexpr Tables(const Mono (&m2)[10], const Mono (&m3)[20]) : e2{}, e3{} {
    for (int i = 0; i < 10; ++i) e2[i] = m2[i];
    for (int i = 0; i < 20; ++i) e3[i] = m3[i];
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) {
        Mono s{e1[i].x + e1[j].x, e1[i].y + e1[j].y, e1[i].z + e1[j].z};
        for (int o = 0; o < 10; ++o)
          if (mono_eq(e2[o], s)) t11[i][j] = o;
      }
    for (int i = 0; i < 10; ++i)
      for (int j = 0; j < 4; ++j) {
        Mono s{e2[i].x + e1[j].x, e2[i].y + e1[j].y, e2[i].z + e1[j].z};
        for (int o = 0; o < 20; ++o)
          if (mono_eq(e3[o], s)) t21[i][j] = o;
      }
  }
};

// Nister's orders (nister.py:410-430)
constexpr Mono kN2[10] = {{2, 0, 0}, {1, 1, 0}, {1, 0, 1}, {1, 0, 0}, {0, 2, 0}, {0, 1, 1}, {0, 1, 0}, {0, 0, 2}, {0, 0, 1}, {0, 0, 0}};
constexpr Mono kN3[20] = {{3, 0, 0}, {0, 3, 0}, {2, 1, 0}, {1, 2, 0}, {2, 0, 1}, {2, 0, 0}, {0, 2, 1}, {0, 2, 0}, {1, 1, 1}, {1, 1, 0},
                          {1, 0, 2}, {1, 0, 1}, {1, 0, 0}, {0, 1, 2}, {0, 1, 1}, {0, 1, 0}, {0, 0, 3}, {0, 0, 2}, {0, 0, 1}, {0, 0, 0}};
// GrevLex orders of the Stewenius solver (stewenius.py:134-172)
constexpr Mono kG2[10] = {{2, 0, 0}, {1, 1, 0}, {0, 2, 0}, {1, 0, 1}, {0, 1, 1}, {0, 0, 2}, {1, 0, 0}, {0, 1, 0}, {0, 0, 1}, {0, 0, 0}};
constexpr Mono kG3[20] = {{3, 0, 0}, {2, 1, 0}, {1, 2, 0}, {0, 3, 0}, {2, 0, 1}, {1, 1, 1}, {0, 2, 1}, {1, 0, 2}, {0, 1, 2}, {0, 0, 3},
                          {2, 0, 0}, {1, 1, 0}, {0, 2, 0}, {1, 0, 1}, {0, 1, 1}, {0, 0, 2}, {1, 0, 0}, {0, 1, 0}, {0, 0, 1}, {0, 0, 0}};

struct NisterOrder { static constexpr Tables T{kN2, kN3}; };
struct GrevlexOrder { static constexpr Tables T{kG2, kG3}; };

template <class Ord>
__device__ __forceinline__ void pmul11(const double (&a)[4], const double (&b)[4], double (&o)[10], double scale = 1.0,
                                       bool accumulate = false) {
  if (!accumulate) {
#pragma unroll
    for (int i = 0; i < 10; ++i) o[i] = 0;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) o[Ord::T.t11[i][j]] += scale * a[i] * b[j];
}

template <class Ord>
__device__ __forceinline__ void pmul21_acc(const double (&a)[10], const double (&b)[4], double (&o)[20], double scale) {
#pragma unroll
  for (int i = 0; i < 10; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) o[Ord::T.t21[i][j]] += scale * a[i] * b[j];
}

// ------------------------------------------------------------------------------------------------
// Constraint system [A | B] (10 x 20) -> selected rows of X = A^-1 B, without pivoting and without run-time
// indexing: A (left 10 x 10) stays in VGPRs and is factored by Householder QR (backward stable with no
// pivoting, unlike the Gauss-Jordan it replaces); B (right 10 x 10) is parked in LDS row by row as the
// constraints are generated and then pulled back one COLUMN at a time: reflectors applied in registers,
// back-substitution down to the first needed row, results written to X.  LDS per lane: 100 doubles
// (51 KiB per 64-lane block => three blocks per CU instead of one), ~200 LDS accesses per sample instead of ~6000.
// kFirstRow = smallest row index the caller needs (rows kFirstRow..9 of X are produced: X[r - kFirstRow][c]).
// ------------------------------------------------------------------------------------------------
// kSplit: two lanes build and factor the same system (same LDS slot); each then solves five of the ten right-hand-side
// columns (`half` = 0 / 1) and both read all results back -- block = one wave, so program order is enough.
// The ten cubic constraints on E(x,y,z) = x B0 + y B1 + z B2 + B3: rows 0-8 = entries (row-major i,j) of
// s*(E E^T E - 1/2 tr(E E^T) E) (s = 2 for Stewenius' 2EE^TE - tr(EE^T)E), row 9 = det E;
// e[i][j][0..3] = entry polynomial (i,j) in (x,y,z,1).  Left block (first ten monomials) -> A, right block -> put_b(row, t, value)
// kSymmetricG: the six distinct entries of E E^T once (round 5).  Chosen per kernel by measurement: Nister's lane-pair kernel gains
// (156.8 -> 153.4 us at 131 072 samples); Nister's two-phase kernel and both Stewenius kernels spill more with it and lose
// (two-phase 132.9 -> 140.1 us; Stewenius lane pairs 1 201 -> 1 520 accumulation-register moves).
template <class Ord, bool kSymmetricG, class PutB>
__device__ __forceinline__ void constraint_rows(const double (&e)[3][3][4], double s, double (&A)[10][10], PutB put_b) {
  auto emit_rows = [&](int i, const double (&g)[3][10]) {   // rows (i, 0..2) from row i of E E^T - 1/2 tr I
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      double row[20];
#pragma unroll
      for (int t = 0; t < 20; ++t) row[t] = 0;
#pragma unroll
      for (int k = 0; k < 3; ++k) pmul21_acc<Ord>(g[k], e[k][j], row, s);
#pragma unroll
      for (int t = 0; t < 10; ++t) {
        A[3 * i + j][t] = row[t];
        put_b(3 * i + j, t, row[10 + t]);
      }
    }
  };
  if constexpr (kSymmetricG) {
    // G = E E^T is symmetric: its six distinct entry polynomials once (rounds 1-4 formed all nine, and the diagonal twice)
    double G[3][3][10];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int k = i; k < 3; ++k) {
        pmul11<Ord>(e[i][0], e[k][0], G[i][k]);
        pmul11<Ord>(e[i][1], e[k][1], G[i][k], 1.0, true);
        pmul11<Ord>(e[i][2], e[k][2], G[i][k], 1.0, true);
      }
#pragma unroll
    for (int t = 0; t < 10; ++t) {
      const double tr = 0.5 * (G[0][0][t] + G[1][1][t] + G[2][2][t]);
      G[0][0][t] -= tr; G[1][1][t] -= tr; G[2][2][t] -= tr;
      G[1][0][t] = G[0][1][t]; G[2][0][t] = G[0][2][t]; G[2][1][t] = G[1][2][t];
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) emit_rows(i, G[i]);
  } else {
    double tr[10];
    {
      double d0[10], d1[10], d2[10];
      pmul11<Ord>(e[0][0], e[0][0], d0); pmul11<Ord>(e[0][1], e[0][1], d0, 1.0, true); pmul11<Ord>(e[0][2], e[0][2], d0, 1.0, true);
      pmul11<Ord>(e[1][0], e[1][0], d1); pmul11<Ord>(e[1][1], e[1][1], d1, 1.0, true); pmul11<Ord>(e[1][2], e[1][2], d1, 1.0, true);
      pmul11<Ord>(e[2][0], e[2][0], d2); pmul11<Ord>(e[2][1], e[2][1], d2, 1.0, true); pmul11<Ord>(e[2][2], e[2][2], d2, 1.0, true);
#pragma unroll
      for (int t = 0; t < 10; ++t) tr[t] = 0.5 * (d0[t] + d1[t] + d2[t]);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      // (E E^T - 1/2 tr I) row i : three degree-2 polynomials
      double g[3][10];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        pmul11<Ord>(e[i][0], e[k][0], g[k]);
        pmul11<Ord>(e[i][1], e[k][1], g[k], 1.0, true);
        pmul11<Ord>(e[i][2], e[k][2], g[k], 1.0, true);
      }
#pragma unroll
      for (int t = 0; t < 10; ++t) g[i][t] -= tr[t];
      emit_rows(i, g);
    }
  }
  {
    double row[20];
#pragma unroll
    for (int t = 0; t < 20; ++t) row[t] = 0;
    double m[10], m2[10];
    pmul11<Ord>(e[0][1], e[1][2], m); pmul11<Ord>(e[0][2], e[1][1], m2);
#pragma unroll
    for (int t = 0; t < 10; ++t) m[t] -= m2[t];
    pmul21_acc<Ord>(m, e[2][0], row, 1.0);
    pmul11<Ord>(e[0][2], e[1][0], m); pmul11<Ord>(e[0][0], e[1][2], m2);
#pragma unroll
    for (int t = 0; t < 10; ++t) m[t] -= m2[t];
    pmul21_acc<Ord>(m, e[2][1], row, 1.0);
    pmul11<Ord>(e[0][0], e[1][1], m); pmul11<Ord>(e[0][1], e[1][0], m2);
#pragma unroll
    for (int t = 0; t < 10; ++t) m[t] -= m2[t];
    pmul21_acc<Ord>(m, e[2][2], row, 1.0);
#pragma unroll
    for (int t = 0; t < 10; ++t) {
      A[9][t] = row[t];
      put_b(9, t, row[10 + t]);
    }
  }
}

// Householder QR of A in place: reflector j lives in A[j..9][j] (v, with v_j = A[j][j]), R above the diagonal, 1 / r_jj in rinv.
// Returns false when |r_jj| underflows against the largest one (nister.py:154-157: rank-deficient left block => sample dropped).
// (LAPACK storage -- v_j = 1 implicit, 1 / r_jj on the diagonal, 20 registers fewer -- was built in round 5: the compiler spills
// MORE with it, 349 instead of 261 accumulation-register moves in the Nister pair kernel, 1 813 instead of 1 312 in Stewenius'.)
__device__ __forceinline__ bool householder_qr10(double (&A)[10][10], double (&beta)[10], double (&rinv)[10]) {
  double rdiag[10];
  double amax = 0;
  bool ok = true;
#pragma unroll
  for (int j = 0; j < 10; ++j) {
    double nrm2 = 0;
#pragma unroll
    for (int i = j; i < 10; ++i) nrm2 += A[i][j] * A[i][j];
    const double nrm = fsqrt(nrm2);
    const double alpha = -dsign(nrm, A[j][j]);
    const double v0 = A[j][j] - alpha;
    const double vtv = v0 * v0 + (nrm2 - A[j][j] * A[j][j]);
    beta[j] = vtv > 0 ? 2.0 * frcp(vtv) : 0.0;
    rdiag[j] = alpha;
    A[j][j] = v0;
    amax = fmax(amax, fabs(alpha));
#pragma unroll
    for (int c = j + 1; c < 10; ++c) {
      double dot = 0;
#pragma unroll
      for (int i = j; i < 10; ++i) dot += A[i][j] * A[i][c];
      dot *= beta[j];
#pragma unroll
      for (int i = j; i < 10; ++i) A[i][c] -= dot * A[i][j];
    }
  }
#pragma unroll
  for (int j = 0; j < 10; ++j) ok = ok && (fabs(rdiag[j]) > 1e-13 * amax);
  ok = ok && is_finite(amax) && amax > 0;
#pragma unroll
  for (int j = 0; j < 10; ++j) rinv[j] = ok ? frcp(rdiag[j]) : 0.0;
  return ok;
}

// one right-hand-side column through the factorisation: reflectors, then back-substitution R x = b for rows 9 .. kFirstRow
template <int kFirstRow>
__device__ __forceinline__ void qr10_solve_column(const double (&A)[10][10], const double (&beta)[10], const double (&rinv)[10],
                                                  double (&b)[10], double (&x)[10]) {
#pragma unroll
  for (int j = 0; j < 10; ++j) {
    double dot = 0;
#pragma unroll
    for (int i = j; i < 10; ++i) dot += A[i][j] * b[i];
    dot *= beta[j];
#pragma unroll
    for (int i = j; i < 10; ++i) b[i] -= dot * A[i][j];
  }
#pragma unroll
  for (int r = 9; r >= kFirstRow; --r) {
    double acc = b[r];
#pragma unroll
    for (int k = r + 1; k < 10; ++k) acc -= A[r][k] * x[k];
    x[r] = acc * rinv[r];
  }
}

template <class Ord, int kFirstRow, bool kSplit = false>
__device__ __forceinline__ bool constraints_reduce(const double (&e)[3][3][4], const LaneWs &Bw, double s,
                                   double (&X)[10 - kFirstRow][10], int half = 0) {
  double A[10][10];
  // A-part to registers, B-part to LDS
  constraint_rows<Ord, DR_K3_SYMG != 0 && kFirstRow == 4>(e, s, A, [&](int r, int t, double v) { Bw[r * 10 + t] = v; });   // kFirstRow == 4: Nister
  // ---- Householder QR of A: reflector j lives in A[j..9][j] (v, with v_j = A[j][j]), R above the diagonal + rdiag
  // (kept inline here, not through householder_qr10 / qr10_solve_column: the shipped pair kernels' register allocation is
  // sensitive to the form -- 16 579 -> 16 766 instructions for Stewenius' through the helpers)
  double beta[10], rdiag[10];
  double amax = 0;
  bool ok = true;
#pragma unroll
  for (int j = 0; j < 10; ++j) {
    double nrm2 = 0;
#pragma unroll
    for (int i = j; i < 10; ++i) nrm2 += A[i][j] * A[i][j];
    const double nrm = fsqrt(nrm2);
    const double alpha = -dsign(nrm, A[j][j]);
    const double v0 = A[j][j] - alpha;
    const double vtv = v0 * v0 + (nrm2 - A[j][j] * A[j][j]);
    beta[j] = vtv > 0 ? 2.0 * frcp(vtv) : 0.0;
    rdiag[j] = alpha;
    A[j][j] = v0;
    amax = fmax(amax, fabs(alpha));
#pragma unroll
    for (int c = j + 1; c < 10; ++c) {
      double dot = 0;
#pragma unroll
      for (int i = j; i < 10; ++i) dot += A[i][j] * A[i][c];
      dot *= beta[j];
#pragma unroll
      for (int i = j; i < 10; ++i) A[i][c] -= dot * A[i][j];
    }
  }
#pragma unroll
  for (int j = 0; j < 10; ++j) ok = ok && (fabs(rdiag[j]) > 1e-13 * amax);
  ok = ok && is_finite(amax) && amax > 0;
  double rinv[10];
#pragma unroll
  for (int j = 0; j < 10; ++j) rinv[j] = ok ? frcp(rdiag[j]) : 0.0;
  // ---- one right-hand-side column at a time
  if (kSplit) wave_lds_order();
#pragma unroll 1
  for (int c0 = 0; c0 < (kSplit ? 5 : 10); ++c0) {
    const int c = kSplit ? c0 + 5 * half : c0;
    double b[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) b[i] = Bw[i * 10 + c];
#pragma unroll
    for (int j = 0; j < 10; ++j) {
      double dot = 0;
#pragma unroll
      for (int i = j; i < 10; ++i) dot += A[i][j] * b[i];
      dot *= beta[j];
#pragma unroll
      for (int i = j; i < 10; ++i) b[i] -= dot * A[i][j];
    }
    // back-substitution R x = b, rows 9 .. kFirstRow
    double x[10];
#pragma unroll
    for (int r = 9; r >= kFirstRow; --r) {
      double acc = b[r];
#pragma unroll
      for (int k = r + 1; k < 10; ++k) acc -= A[r][k] * x[k];
      x[r] = acc * rinv[r];
    }
    // park the result where the column came from; it is re-read with static indices below
#pragma unroll
    for (int r = kFirstRow; r < 10; ++r) Bw[r * 10 + c] = x[r];
  }
  if (kSplit) wave_lds_order();
#pragma unroll
  for (int r = kFirstRow; r < 10; ++r)
#pragma unroll
    for (int c = 0; c < 10; ++c) X[r - kFirstRow][c] = Bw[r * 10 + c];
  return ok;
}

// The same reduction for the FRONT stage of the two-phase kernels (round 5): ONE lane per sample, so the block's 64 lanes park
// 64 right blocks -- 100 doubles each would be 51 KiB, one block too many for four blocks per CU.  Rows 0 .. kLdsRows-1 wait in
// LDS (stride 64), the last rows stay in registers; the column loop is unrolled (static register indices) and every lane
// solves all ten columns.  Results of the LDS rows are parked where the column came from and re-read at the end, like above.
// a double parked in two ACCUMULATION registers (gfx950: 256 AGPRs next to the 256 VGPRs at one wave per SIMD) by explicit
// v_accvgpr_write / read: one move per dword and direction, once -- the compiler's own spilling moved the parked rows back and forth
// (2 300 instead of 260 moves in the kernel)
struct AccDouble {
  unsigned lo, hi;   // "a"-class values
  __device__ __forceinline__ void put(double v) {
    const unsigned l = (unsigned)__double2loint(v), h = (unsigned)__double2hiint(v);
    asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(lo) : "v"(l));
    asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(hi) : "v"(h));
  }
  __device__ __forceinline__ double get() const {
    unsigned l, h;
    asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(l) : "a"(lo));
    asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(h) : "a"(hi));
    return __hiloint2double((int)h, (int)l);
  }
};

template <class Ord, int kFirstRow, int kLdsRows = 7>
__device__ __forceinline__ bool constraints_reduce_front(const double (&e)[3][3][4], const LaneWs &Bw, double s,
                                                         double (&X)[10 - kFirstRow][10]) {
  static_assert(kLdsRows >= kFirstRow && kLdsRows <= 10, "rows kept in registers must be result rows");
  double A[10][10];
  AccDouble Br[kLdsRows < 10 ? 10 - kLdsRows : 1][10];
  constraint_rows<Ord, false>(e, s, A, [&](int r, int t, double v) {
    if (r < kLdsRows) Bw[r * 10 + t] = v;
    else Br[r - kLdsRows][t].put(v);
  });
  double beta[10], rinv[10];
  const bool ok = householder_qr10(A, beta, rinv);
#pragma unroll
  for (int c = 0; c < 10; ++c) {
    double b[10], x[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) b[i] = (i < kLdsRows) ? Bw[i * 10 + c] : Br[i - kLdsRows][c].get();
    qr10_solve_column<kFirstRow>(A, beta, rinv, b, x);
#pragma unroll
    for (int r = kFirstRow; r < 10; ++r) {
      if (r < kLdsRows) Bw[r * 10 + c] = x[r];
      else X[r - kFirstRow][c] = x[r];
    }
  }
#pragma unroll
  for (int r = kFirstRow; r < kLdsRows; ++r)
#pragma unroll
    for (int c = 0; c < 10; ++c) X[r - kFirstRow][c] = Bw[r * 10 + c];
  return ok;
}

// rows (x1x2, x1y2, x1, y1x2, y1y2, y1, x2, y2, 1) of the five-point solvers (nister.py:87-115)
__device__ __forceinline__ void epipolar_row_5pt(double x1, double y1, double x2, double y2, double w, double (&r)[9]) {
  r[0] = w * x1 * x2; r[1] = w * x1 * y2; r[2] = w * x1;
  r[3] = w * y1 * x2; r[4] = w * y1 * y2; r[5] = w * y1;
  r[6] = w * x2; r[7] = w * y2; r[8] = w;
}
// rows (x1x2, x2y1, x2, y2x1, y2y1, y2, x1, y1, 1) of the fundamental-matrix solvers (fundamental…:243-246)
__device__ __forceinline__ void epipolar_row_f(double x1, double y1, double x2, double y2, double w, double (&r)[9]) {
  r[0] = w * x1 * x2; r[1] = w * x2 * y1; r[2] = w * x2;
  r[3] = w * y2 * x1; r[4] = w * y2 * y1; r[5] = w * y2;
  r[6] = w * x1; r[7] = w * y1; r[8] = w;
}

}  // namespace dr
