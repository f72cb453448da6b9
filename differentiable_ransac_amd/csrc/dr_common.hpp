// Shared helpers for the gfx950 kernels of libdransac.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/dransac.h"

namespace dr {

constexpr int kWave = 64;  // CDNA4 wavefront

void set_error(const char *fmt, ...);

inline int check_launch(const char *what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return DR_ELAUNCH;
  }
  return DR_OK;
}

#define DR_REQUIRE(cond, msg)            \
  do {                                   \
    if (!(cond)) {                       \
      ::dr::set_error("%s: %s", __func__, msg); \
      return DR_EINVAL;                  \
    }                                    \
  } while (0)

// Per-pair gate of the multi-round test-mode drivers (round 5: device-side termination): a pair whose iteration counter has
// reached its adaptive bound (ransac.py:135-144, kept on the device by dr_ransac_update) has terminated; the kernels of later rounds
// return at once for its blocks -- nothing about a round is read back by the host, so a whole multi-round call is one HIP graph.
struct PairGate {
  const int32_t *iters = nullptr;
  const double *max_iters = nullptr;
  __device__ __forceinline__ bool closed(int p) const { return iters && (double)iters[p] >= max_iters[p]; }
  // every pair of [first, last] closed (a block of samples may span more than two pairs when ransac_batch_size is below the
  // samples per block: the pairs in the middle must be closed too)
  __device__ __forceinline__ bool closed_range(int first, int last) const {
    if (!iters) return false;
    for (int p = first; p <= last; ++p)
      if (!closed(p)) return false;
    return true;
  }
};

// The per-call sampler key of the batched drivers, advanced on the device (dr_seed_next_n; also the tail of dr_ransac_init when it
// is handed a seed state): state[0] = base, state[1] = calls so far -> out[i] = base * 0x9E3779B97F4A7C15 + calls + i, calls += n.
// Called by ONE block; returns after a block-level barrier.
__device__ __forceinline__ void seed_next_block(uint64_t *__restrict__ state, uint64_t *__restrict__ out, int n) {
  const uint64_t s0 = state[0] * 0x9E3779B97F4A7C15ull + state[1];
  for (int i = threadIdx.x; i < n; i += blockDim.x) out[i] = s0 + (uint64_t)i;
  __syncthreads();
  if (threadIdx.x == 0) state[1] += (uint64_t)n;
}

// Per-pair weights of the exponential-race sampler (gumbel_topk.hip, round 6): w_n = exp(lmax - logit_n) into ws [P,N]; then three words
// per pair: the "tame" flag (all logits finite, span <= 80) at ws[P * N + p]; -1 / c_p at ws[P * N + P + p], c_p = ln 2 sum_n 1 / w_n
// (the expected number of keys >= t is -t c_p: the sampler's threshold search starts from it); lmax at ws[P * N + 2 P + p] (train
// mode's log-sum-exp).  ws holds (N + 32) * P floats.  One 256-thread block per pair.
__device__ __forceinline__ void race_weights_block(const float *__restrict__ logits, int N, int P, int p, float *__restrict__ ws) {
  __shared__ float s_mx[4], s_mn[4], s_rate[4];
  __shared__ int s_bad[4];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const float *lg = logits + (size_t)p * N;
  // rows the register sampler serves (N <= 2048): the thread's eight logits stay in registers between the two passes -- one trip to
  // memory, all loads in flight at once (the launch is a chain of latencies: 6.8 us with the second pass re-reading them)
  constexpr int kRegs = 8;
  const bool in_regs = N <= kRegs * 256;
  float lr[kRegs];
  float mx = -INFINITY, mn = INFINITY;
  int bad = 0;
  if (in_regs) {
#pragma unroll
    for (int u = 0; u < kRegs; ++u) lr[u] = (tid + 256 * u < N) ? lg[tid + 256 * u] : 0.f;
#pragma unroll
    for (int u = 0; u < kRegs; ++u) {
      const bool have = tid + 256 * u < N, fin = fabsf(lr[u]) < INFINITY;   // fin: false for NaN and +-inf
      bad |= (have && !fin) ? 1 : 0;
      mx = (have && fin) ? fmaxf(mx, lr[u]) : mx;
      mn = (have && fin) ? fminf(mn, lr[u]) : mn;
    }
  } else {
    for (int n = tid; n < N; n += 256) {
      const float l = lg[n];
      const bool fin = fabsf(l) < INFINITY;
      bad |= fin ? 0 : 1;
      mx = fin ? fmaxf(mx, l) : mx;
      mn = fin ? fminf(mn, l) : mn;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    mn = fminf(mn, __shfl_xor(mn, o, 64));
    bad |= __shfl_xor(bad, o, 64);
  }
  if (lane == 0) { s_mx[wv] = mx; s_mn[wv] = mn; s_bad[wv] = bad; }
  __syncthreads();
  mx = fmaxf(fmaxf(s_mx[0], s_mx[1]), fmaxf(s_mx[2], s_mx[3]));
  mn = fminf(fminf(s_mn[0], s_mn[1]), fminf(s_mn[2], s_mn[3]));
  bad = s_bad[0] | s_bad[1] | s_bad[2] | s_bad[3];
  const bool tame = !bad && (mx - mn) <= 80.0f;
  float *w = ws + (size_t)p * N;
  float rate = 0.f;   // sum_n 1 / w_n: the density of the pair's keys (the sampler's threshold search starts from it)
  if (in_regs) {
#pragma unroll
    for (int u = 0; u < kRegs; ++u) {
      if (tid + 256 * u < N) {
        const float d = (mx - lr[u]) * 1.44269504088896340736f;
        w[tid + 256 * u] = tame ? __builtin_amdgcn_exp2f(d) : 0.f;
        rate += tame ? __builtin_amdgcn_exp2f(-d) : 0.f;
      }
    }
  } else {
    for (int n = tid; n < N; n += 256) {
      const float d = (mx - lg[n]) * 1.44269504088896340736f;
      w[n] = tame ? __builtin_amdgcn_exp2f(d) : 0.f;
      rate += tame ? __builtin_amdgcn_exp2f(-d) : 0.f;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) rate += __shfl_xor(rate, o, 64);
  if (lane == 0) s_rate[wv] = rate;
  __syncthreads();
  if (tid == 0) {
    reinterpret_cast<int *>(ws + (size_t)P * N)[p] = tame ? 1 : 0;
    ws[(size_t)P * N + P + p] = -1.0f / (0.69314718055994530942f * ((s_rate[0] + s_rate[1]) + (s_rate[2] + s_rate[3])));
    ws[(size_t)P * N + 2 * (size_t)P + p] = mx;   // (train mode: lse = lmax + ln(sum 1 / -key) - ln ln 2)
  }
}

// ---- wave64 reductions (ds_swizzle/DPP chosen by the compiler from the xor pattern) ----
template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// Sum over the 64 lanes by DPP only (six VALU instructions, no LDS-pipe traffic; __shfl_xor compiles to ds_bpermute):
// butterfly inside each row of 16 lanes (quad_perm, row_half_mirror, row_mirror), then row_bcast:15 / row_bcast:31.
// The total is valid in LANE 63 ONLY.
__device__ __forceinline__ float wave_sum_lane63(float v) {
  v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
  v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
  v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x141, 0xF, 0xF, true));   // row_half_mirror
  v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x140, 0xF, 0xF, true));   // row_mirror: row sums
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0xA, 0xF, false));   // row_bcast:15 -> rows 1, 3
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x143, 0xC, 0xF, false));   // row_bcast:31 -> rows 2, 3
  return v;
}

// Maximum over the 64 lanes by DPP, broadcast to every lane through an SGPR (v_readlane of lane 63).
// f64 variant: the two halves travel through DPP separately; total valid in lane 63 only.
template <int kCtrl, int kRowMask, bool kBound>
__device__ __forceinline__ double dpp_add_step(double x) {
  const long long xi = __double_as_longlong(x);
  const int lo = (int)xi, hi = (int)(xi >> 32);
  const int tlo = kBound ? __builtin_amdgcn_mov_dpp(lo, kCtrl, kRowMask, 0xF, true) : __builtin_amdgcn_update_dpp(0, lo, kCtrl, kRowMask, 0xF, false);
  const int thi = kBound ? __builtin_amdgcn_mov_dpp(hi, kCtrl, kRowMask, 0xF, true) : __builtin_amdgcn_update_dpp(0, hi, kCtrl, kRowMask, 0xF, false);
  return x + __longlong_as_double(((long long)thi << 32) | (unsigned int)tlo);
}
__device__ __forceinline__ double wave_sum_lane63(double v) {
  v = dpp_add_step<0xB1, 0xF, true>(v);
  v = dpp_add_step<0x4E, 0xF, true>(v);
  v = dpp_add_step<0x141, 0xF, true>(v);
  v = dpp_add_step<0x140, 0xF, true>(v);
  v = dpp_add_step<0x142, 0xA, false>(v);
  v = dpp_add_step<0x143, 0xC, false>(v);
  return v;
}

template <int kCtrl, int kRowMask>
__device__ __forceinline__ float dpp_max_step(float x) {
  const int xi = __float_as_int(x);
  return fmaxf(x, __int_as_float(__builtin_amdgcn_update_dpp(xi, xi, kCtrl, kRowMask, 0xF, false)));
}
__device__ __forceinline__ float wave_max_bcast(float v) {
  v = dpp_max_step<0xB1, 0xF>(v);    // quad_perm [1,0,3,2]
  v = dpp_max_step<0x4E, 0xF>(v);    // quad_perm [2,3,0,1]
  v = dpp_max_step<0x141, 0xF>(v);   // row_half_mirror
  v = dpp_max_step<0x140, 0xF>(v);   // row_mirror: row maxima
  v = dpp_max_step<0x142, 0xA>(v);   // row_bcast:15 -> rows 1, 3
  v = dpp_max_step<0x143, 0xC>(v);   // row_bcast:31 -> rows 2, 3
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_sum_bcast(float v) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wave_sum_lane63(v)), 63));
}

template <typename T>
__device__ __forceinline__ T wave_max(T v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    T w = __shfl_xor(v, o, 64);
    v = w > v ? w : v;
  }
  return v;
}

__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ double fast_rcp(double x) { return 1.0 / x; }

// NOTE: never test finiteness with (x - x) == 0: under -ffp-contract=fast the backend may fuse the producer
// of x into the subtraction (fma(a, b, c - x)) and the "difference" is then a rounding residue, not zero.
template <typename T>
__device__ __forceinline__ bool is_finite(T x) {
  return __builtin_isfinite(x);
}

// Philox4x32-7 (Salmon et al., SC'11: the counter-based generator of Random123; seven rounds is the paper's smallest
// "Crush-resistant" Philox4x32 -- it passes BigCrush -- and ten is its default with a safety margin).  Round 3: 10 -> 7 rounds
// and the two three-way XORs of a round as ONE v_bitop3_b32 each (gfx950 has no v_xor3_b32; the compiler does not select the
// three-input boolean op by itself): 4 vector instructions per round instead of 6, 28 per call instead of 60.  The in-kernel
// stream is not part of any parity contract (bit-exactness is defined on EXPLICIT noise); its integer side is pinned by the
// numpy restatement tests/philox_ref.py, its law by the chi-square / inclusion tests.
struct Philox {
  static constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
  __host__ __device__ static inline void mulhilo(uint32_t a, uint32_t b, uint32_t &hi, uint32_t &lo) {
    uint64_t p = (uint64_t)a * b;
    hi = (uint32_t)(p >> 32);
    lo = (uint32_t)p;
  }
  __host__ __device__ static inline uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(DR_PHILOX_NO_BITOP3)
    uint32_t r;
    // truth table of a ^ b ^ c; the third operand is the round key -- wave-uniform in every kernel (derived from the seed), so
    // it stays in an SGPR (a "v" constraint costs one v_mov_b32 per use: 6 instructions per round again)
    asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x96" : "=v"(r) : "v"(a), "v"(b), "s"(c));
    return r;
#else
    return a ^ b ^ c;
#endif
  }
  __host__ __device__ static inline void gen(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                             uint32_t out[4]) {
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#ifndef DR_PHILOX_ROUNDS
#define DR_PHILOX_ROUNDS 7
#endif
#pragma unroll
    for (int r = 0; r < DR_PHILOX_ROUNDS; ++r) {
      uint32_t h0, l0, h1, l1;
      mulhilo(M0, c0, h0, l0);
      mulhilo(M1, c2, h1, l1);
      uint32_t n0 = xor3(h1, c1, k0), n1 = l1, n2 = xor3(h0, c3, k1), n3 = l0;
      c0 = n0; c1 = n1; c2 = n2; c3 = n3;
      k0 += W0; k1 += W1;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
  }
};

// u32 -> Gumbel(0,1) sample, mirroring torch.distributions.Gumbel: u = tiny + r*(1-eps-tiny), r uniform on a 2^-24 grid of
// [0, 1].  r = RNE-to-24-bits(bits) * 2^-32 (one v_cvt_f32_u32: no shift) and u = fma(cvt, 2^-32 * (1-eps-tiny), tiny): the
// scale factor is folded exactly (a power of two times the constant), so u equals fl(r * c + tiny) bit for bit and costs two
// instructions instead of four (shift, convert, multiply, fma).
__device__ __forceinline__ float gumbel_from_bits(uint32_t bits) {
  constexpr float kScale = 2.3283064365386963e-10f * (1.0f - 1.1920928955078125e-07f - 1.17549435e-38f);   // 2^-32 * c
  float u = __builtin_fmaf((float)bits, kScale, 1.17549435e-38f);   // in [2^-126, 1 - eps]
  // -ln u in [1.2e-7, 87.4]: both log arguments are normal numbers, so the raw v_log_f32 (no denormal fix-up sequence) is
  // exact to its 1-ulp spec
  const float kLn2 = 0.69314718055994530942f;
#if defined(DR_K1_NOISE_EXPERIMENT) && DR_K1_NOISE_EXPERIMENT == 1   // timing experiment: no logarithm at all
  return u;
#elif defined(DR_K1_NOISE_EXPERIMENT) && DR_K1_NOISE_EXPERIMENT == 2   // timing experiment: one logarithm
  return -kLn2 * __builtin_amdgcn_logf(u);
#endif
  // -ln(-ln u) = -ln2 * log2(-ln2 * log2 u) = -ln2 * log2(-log2 u) - ln2 * log2(ln2): the inner scale factor leaves the
  // logarithm as a constant, the sign of log2 u is a free source modifier of the second v_log_f32 -- five instructions per
  // sample (convert, fma, log, log, fma) instead of six
  const float t = __builtin_amdgcn_logf(u);                                  // log2 u in [-126, -1.7e-7]
  float g = __builtin_fmaf(-kLn2, __builtin_amdgcn_logf(-t), 0.36651292058166432701f);   // -ln2 * log2(ln2) = -ln(ln 2)
  // the sample is an f32 VALUE: rounded here, then added to the logit by the caller with a second rounding -- exactly what the
  // oracle does with the noise tensor the general kernel reports.  The empty asm hides the multiply from -ffp-contract=fast,
  // which otherwise fuses it into the caller's `logit + noise` (one rounding): a top-k decided by the last bit then differs
  // from the oracle's on the reported noise (seen as soon as the generator changed: tests/test_gpu_round2.py, 2000 x 1024 x 5)
  asm("" : "+v"(g));
  return g;
}

// ---- optional per-stage cycle accounting (profiling builds only: -DDR_PROFILE_STAGES) --------------------
#ifdef DR_PROFILE_STAGES
static __device__ unsigned long long g_stage_cycles[32];   // one copy per translation unit (no RDC)
#define DR_STAGE_BEGIN() unsigned long long _t_prev = __builtin_readcyclecounter()
#define DR_STAGE(i)                                                        \
  do {                                                                     \
    unsigned long long _t = __builtin_readcyclecounter();                  \
    if ((threadIdx.x & 63) == 0) atomicAdd(&::dr::g_stage_cycles[i], _t - _t_prev); \
    _t_prev = __builtin_readcyclecounter();                                \
  } while (0)
#define DR_DEFINE_STAGE_READER(name)                                                                          \
  extern "C" int name(unsigned long long *out16) {   /* 32 entries */                                         \
    unsigned long long zero[32] = {0};                                                                        \
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(::dr::g_stage_cycles), sizeof(zero)) != hipSuccess) return -2; \
    if (hipMemcpyToSymbol(HIP_SYMBOL(::dr::g_stage_cycles), zero, sizeof(zero)) != hipSuccess) return -2;    \
    return 0;                                                                                                 \
  }
#else
#define DR_DEFINE_STAGE_READER(name)
#define DR_STAGE_BEGIN() do {} while (0)
#define DR_STAGE(i) do {} while (0)
#endif

}  // namespace dr
