// PROTOTYPE (not part of libdransac.so): MSAC scoring with the five linear forms a = M^T x2 (3), b = M x1 (2) on the bf16
// matrix cores, every f32 operand split exactly into three bf16 values (hi + mid + lo) and every coefficient x coordinate
// product formed from six bf16 products (hh, hm, mh, hl, lh, mm) accumulated in f32 -- DESIGN.md section 9, item 3.
//
//   one form  c0*x + c1*y + c2   =  a K = 16 dot product:  k 0-7  (lanes 0-31 of the operand registers):  x-terms + c2.hi, c2.mid
//                                                         k 8-15 (lanes 32-63):                         y-terms + c2.lo, 0
//   MFMA-1 rows = (model, a-form) : 10 models x 3;   MFMA-2 rows = (model, b-form) : 10 models x 2
//   column n of iteration t = point 16 n + t of the wave's 512-point span, so a lane ends with 16 consecutive mask bytes per
//   model; C rows (reg&3) + 8 (reg>>2) + 4 (lane>>5): each half-wave owns five of the ten models.
//
// Build + run: scratch/k4_bf16.py.  Needs N % 16 == 0.
#include <type_traits>

#include "../differentiable_ransac_amd/csrc/dr_common.hpp"

namespace dr {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#ifndef DR_Q_THREADS
#define DR_Q_THREADS 256     // A/B knobs of the prototype: block size, occupancy cap, in-flight window
#endif
#ifndef DR_Q_WAVES
#define DR_Q_WAVES 2
#endif
#ifndef DR_Q_TIE
#define DR_Q_TIE 1
#endif
#ifndef DR_Q_PIPE
#define DR_Q_PIPE 0    // 1: step s+1's MFMAs are issued before step s's epilogue (explicit software pipeline; UNTESTED on the
                       // device at the time of writing -- compiles, 0 spills, MFMAs sit in front of the epilogues in the ISA)
#endif
#ifndef DR_Q_NOEPI
#define DR_Q_NOEPI 0   // 1: timing decomposition only -- the epilogue reduced to one add per model (wrong results)
#endif
constexpr int kQT = DR_Q_THREADS, kQTile = 64, kQG = 10, kQWavePts = 512, kQChunk = (kQT / 64) * kQWavePts;

// f32 -> (hi, mid, lo) bf16 bit patterns (low 16 bits of each result) by round-to-nearest splits: hi + mid + lo == x exactly
// (8 + 8 + <= 8 significant bits), |mid| <= 2^-9 |x|, |lo| <= 2^-18 |x|, so the three products the K = 16 budget drops
// (mid*lo, lo*mid, lo*lo) are below 2^-26 of the term.  (Truncating splits: 2^-24 -- measured 2.4e-4 on a score.)
__device__ __forceinline__ uint32_t bf16_bits(float x) { return (uint32_t)__builtin_bit_cast(unsigned short, (__bf16)x); }
__device__ __forceinline__ void split3(float x, uint32_t &h, uint32_t &m, uint32_t &l) {
  h = bf16_bits(x);
  const float r1 = x - __uint_as_float(h << 16);
  m = bf16_bits(r1);
  const float r2 = r1 - __uint_as_float(m << 16);
  l = bf16_bits(r2);
}
// K order of one form (six products + the constant):  point side  (h, m, l, h, m, h, 1, 1)
//                                                      model side  (H, H, H, M, M, L, k, k')   ->  hH mH lH hM mM hL
// point-side fragment of one coordinate, two registers: w0 = h | m << 16, w1 = l | h << 16; the third dword of the operand
// (m | h << 16) is w0 rotated by 16, the fourth the constant (1, 1)
__device__ __forceinline__ void point_frag(float x, uint32_t (&w)[2]) {
  uint32_t h, m, l;
  split3(x, h, m, l);
  w[0] = h | (m << 16);
  w[1] = l | (h << 16);
}
__device__ __forceinline__ bf16x8 point_operand(const uint32_t (&w)[2]) {
  const u32x4 b = {w[0], w[1], __builtin_amdgcn_alignbit(w[0], w[0], 16), 0x3F803F80u};
  return __builtin_bit_cast(bf16x8, b);
}
// model-side fragment: variable coefficient c -> (H, H, H, M, M, L); constant k -> (k.h, k.m) in the lower k-half, (k.l, 0)
// in the upper one
__device__ __forceinline__ u32x4 model_operand(float c, float k, int khalf, bool live) {
  uint32_t h, m, l, kh, km, kl;
  split3(c, h, m, l);
  split3(k, kh, km, kl);
  u32x4 f;
  f[0] = h | (h << 16);
  f[1] = h | (m << 16);
  f[2] = m | (l << 16);
  f[3] = khalf == 0 ? (kh | (km << 16)) : kl;
  if (!live) f = (u32x4){0u, 0u, 0u, 0u};
  return f;
}
// mask bytes: byte q of `acc` <- 0xFF if the sign bit of `bits` is set, else 0x00 (v_perm_b32 selector 11 = sign of S0[31])
__device__ __forceinline__ uint32_t sign_into_byte(int q, uint32_t bits, uint32_t acc) {   // q folds after unrolling
  const uint32_t sel = (0x03020100u & ~(0xffu << (8 * q))) | (0x0Bu << (8 * q));
  return __builtin_amdgcn_perm(bits, acc, sel);
}

// v2: points outermost (a lane's sixteen points are fetched four at a time, never all resident), up to three groups of ten
// models innermost (their operands stay in registers), mask bytes staged per four points in wave-private LDS and written as
// 16-byte row pieces at the end of the pass.  v1 (all sixteen points' fragments + all accumulators resident) needed > 256
// registers: 442 spills, 3.3 ms.
constexpr int kQGP = 3;   // groups per pass
__global__ __launch_bounds__(kQT) __attribute__((amdgpu_waves_per_eu(DR_Q_WAVES, DR_Q_WAVES))) void msac_score_bf16x3_kernel(
    const float *__restrict__ matches, const float *__restrict__ models, const uint8_t *__restrict__ valid,
    const float *__restrict__ thr, int M, int N, float *__restrict__ scores, uint8_t *__restrict__ masks, int use_atomic) {
  __shared__ int s_slot[kQTile + kQG * kQGP];
  __shared__ int s_fin[kQTile];
  __shared__ int s_nv;
  __shared__ uint32_t s_inv[2];
  __shared__ float part[kQT / 64][kQTile];
  __shared__ uint32_t stage[kQT / 64][kQGP * 5][4][64];   // [wave][model of the lane's half][four-point word][lane]
  const int p = blockIdx.z, m0 = blockIdx.x * kQTile;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int n = lane & 31, half = lane >> 5;
  const int mcount = min(kQTile, M - m0);
  const float t15 = 1.5f * thr[p];
  const float inv_thr2 = 1.0f / (t15 * t15);
  const float4 *mt = reinterpret_cast<const float4 *>(matches + (size_t)p * N * 4);
  const float *md = models + ((size_t)p * M + m0) * 9;

  for (int i = tid; i < (kQT / 64) * kQTile; i += kQT) (&part[0][0])[i] = 0.f;
  if (tid < kQTile + kQG * kQGP) s_slot[tid] = -1;
  __syncthreads();
  if (wv == 0) {
    const bool v = lane < mcount && (!valid || valid[(size_t)p * M + m0 + lane] != 0);
    const unsigned long long b = __ballot(v);
    if (v) s_slot[__popcll(b & ((1ull << lane) - 1ull))] = lane;
    if (lane == 0) {
      s_nv = __popcll(b);
      const unsigned long long inv = ~b & (mcount >= 64 ? ~0ull : ((1ull << mcount) - 1ull));
      s_inv[0] = (uint32_t)inv;
      s_inv[1] = (uint32_t)(inv >> 32);
    }
    bool fin = true;
    if (lane < mcount)
      for (int q = 0; q < 9; ++q) fin = fin && is_finite(md[lane * 9 + q]);
    s_fin[lane] = fin ? 1 : 0;
  }
  __syncthreads();
  const int nv = __builtin_amdgcn_readfirstlane(s_nv);   // block-uniform: keep the group branches on the scalar unit

  const int n0 = blockIdx.y * kQChunk + wv * kQWavePts + 16 * n;   // the lane's sixteen points (both half-waves: the same)
  const bool have = n0 < N;

  // which (model, form) this lane's operand ROW is: row R = n  <->  C register `reg` of half-wave hc
  const int hc = (n >> 2) & 1, reg = (n & 3) + 4 * (n >> 3);
  const int j1 = reg / 3, f1 = reg % 3;   // MFMA-1: a_f = m[f] x2 + m[3+f] y2 + m[6+f]
  const int j2 = reg >> 1, f2 = reg & 1;  // MFMA-2: b_f = m[3f] x1 + m[3f+1] y1 + m[3f+2]
  const bool live1 = reg < 15, live2 = reg < 10;

  // one pass over <= kQGP groups, specialised on the group count (no per-group branches inside: they cost exec-mask
  // bookkeeping and register copies at every merge)
  auto pass = [&](int g0, auto ngc) {
    constexpr int ng = decltype(ngc)::value;
    u32x4 A1[ng], A2[ng];
    float acc[ng][5];
#pragma unroll
    for (int g = 0; g < ng; ++g) {
      const int base = (g0 + g) * kQG + 5 * hc;
      const int s1 = live1 ? s_slot[base + j1] : -1;
      const int s2 = live2 ? s_slot[base + j2] : -1;
      float c1v = 0.f, c1k = 0.f, c2v = 0.f, c2k = 0.f;
      if (s1 >= 0) {
        c1v = md[s1 * 9 + (half ? 3 + f1 : f1)];
        c1k = md[s1 * 9 + 6 + f1];
      }
      if (s2 >= 0) {
        c2v = md[s2 * 9 + 3 * f2 + (half ? 1 : 0)];
        c2k = md[s2 * 9 + 3 * f2 + 2];
      }
      A1[g] = model_operand(c1v, c1k, half, s1 >= 0 && s_fin[max(s1, 0)] != 0);
      A2[g] = model_operand(c2v, c2k, half, s2 >= 0 && s_fin[max(s2, 0)] != 0);
#pragma unroll
      for (int jj = 0; jj < 5; ++jj) acc[g][jj] = 0.f;
    }
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;

#if DR_Q_PIPE
    // ---- software pipeline over the 4 * ng (point, group) steps of a four-point batch
    auto epilogue = [&](const f32x16 &C1, const f32x16 &C2, float x1, float y1, int ti, float (&ac)[5], uint32_t (&mw)[5]) {
#pragma unroll
      for (int jj = 0; jj < 5; ++jj) {
        const float a0 = C1[3 * jj], a1 = C1[3 * jj + 1], a2 = C1[3 * jj + 2];
        const float b0 = C2[2 * jj], b1f = C2[2 * jj + 1];
        const float r = fmaf(x1, a0, fmaf(y1, a1, a2));
        const float J = fmaf(a0, a0, fmaf(a1, a1, fmaf(b0, b0, b1f * b1f)));
        const float sv = fmaf((r * r) * __builtin_amdgcn_rcpf(J), inv_thr2, -1.0f);
        const uint32_t bits = __float_as_uint(sv);
        ac[jj] += __int_as_float(min((int)bits, 0));
        mw[jj] = sign_into_byte(ti, bits, mw[jj]);
      }
    };
    float4 nxt[4];   // the next batch's points, requested one batch ahead
#pragma unroll
    for (int ti = 0; ti < 4; ++ti) nxt[ti] = have ? mt[n0 + ti] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 1
    for (int tq = 0; tq < 4; ++tq) {
      float x1[4], y1[4];
      bf16x8 B1[4], B2[4];
#pragma unroll
      for (int ti = 0; ti < 4; ++ti) {
        uint32_t w1[2], w2[2];
        point_frag(half ? nxt[ti].w : nxt[ti].z, w1);
        point_frag(half ? nxt[ti].y : nxt[ti].x, w2);
        B1[ti] = point_operand(w1);
        B2[ti] = point_operand(w2);
        x1[ti] = nxt[ti].x;
        y1[ti] = nxt[ti].y;
      }
      if (tq < 3) {
#pragma unroll
        for (int ti = 0; ti < 4; ++ti) nxt[ti] = have ? mt[n0 + 4 * (tq + 1) + ti] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      uint32_t mkq[ng][5];
#pragma unroll
      for (int g = 0; g < ng; ++g)
#pragma unroll
        for (int jj = 0; jj < 5; ++jj) mkq[g][jj] = 0u;
      constexpr int S = 4 * ng;
      f32x16 Ca1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A1[0]), B1[0], z, 0, 0, 0);
      f32x16 Ca2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A2[0]), B2[0], z, 0, 0, 0);
#pragma unroll
      for (int st = 0; st < S; ++st) {
        const int ti = st / ng, g = st % ng;
        f32x16 Cb1 = Ca1, Cb2 = Ca2;
        if (st + 1 < S) {
          const int tn = (st + 1) / ng, gn = (st + 1) % ng;
          Cb1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A1[gn]), B1[tn], z, 0, 0, 0);
          Cb2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A2[gn]), B2[tn], z, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);   // the machine scheduler keeps the MFMAs of step st+1 above this epilogue
        epilogue(Ca1, Ca2, x1[ti], y1[ti], ti, acc[g], mkq[g]);
        // and the IR keeps step st+2's MFMAs below it: their model operands pass through an empty asm that reads acc
        if (st + 2 < S) {
          const int g2 = (st + 2) % ng;
          asm volatile("" : "+v"(A1[g2]), "+v"(A2[g2]) : "v"(acc[g][0]), "v"(acc[g][4]));
        }
        __builtin_amdgcn_sched_barrier(0);
        Ca1 = Cb1;
        Ca2 = Cb2;
      }
#pragma unroll
      for (int g = 0; g < ng; ++g)
#pragma unroll
        for (int jj = 0; jj < 5; ++jj) stage[wv][g * 5 + jj][tq][lane] = mkq[g][jj] & 0x01010101u;
    }
#else
#pragma unroll 1
    for (int tq = 0; tq < 4; ++tq) {
      float4 pt[4];
#pragma unroll
      for (int ti = 0; ti < 4; ++ti) pt[ti] = have ? mt[n0 + 4 * tq + ti] : make_float4(0.f, 0.f, 0.f, 0.f);
      uint32_t mkq[ng][5];
#pragma unroll
      for (int g = 0; g < ng; ++g)
#pragma unroll
        for (int jj = 0; jj < 5; ++jj) mkq[g][jj] = 0u;
#pragma unroll
      for (int ti = 0; ti < 4; ++ti) {
        uint32_t w1[2], w2[2];
        point_frag(half ? pt[ti].w : pt[ti].z, w1);
        point_frag(half ? pt[ti].y : pt[ti].x, w2);
        const bf16x8 B1 = point_operand(w1), B2 = point_operand(w2);
        const float x1 = pt[ti].x, y1 = pt[ti].y;
#pragma unroll
        for (int g = 0; g < ng; ++g) {
          {
            const f32x16 C1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A1[g]), B1, z, 0, 0, 0);
            const f32x16 C2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A2[g]), B2, z, 0, 0, 0);
#pragma unroll
            for (int jj = 0; jj < 5; ++jj) {
              const float a0 = C1[3 * jj], a1 = C1[3 * jj + 1], a2 = C1[3 * jj + 2];
              const float b0 = C2[2 * jj], b1f = C2[2 * jj + 1];
              if (DR_Q_NOEPI) {
                acc[g][jj] += a0 + b0;
                continue;
              }
              const float r = fmaf(x1, a0, fmaf(y1, a1, a2));
              const float J = fmaf(a0, a0, fmaf(a1, a1, fmaf(b0, b0, b1f * b1f)));
              const float sv = fmaf((r * r) * __builtin_amdgcn_rcpf(J), inv_thr2, -1.0f);
              const uint32_t bits = __float_as_uint(sv);
              acc[g][jj] += __int_as_float(min((int)bits, 0));
              mkq[g][jj] = sign_into_byte(ti, bits, mkq[g][jj]);
            }
            // at most two (point, group) steps in flight: the operands of the step after next wait for this epilogue
            const int kNext = (g + 2) % ng;   // folds after unrolling
            if (DR_Q_TIE) asm volatile("" : "+v"(A1[kNext]), "+v"(A2[kNext]) : "v"(acc[g][0]), "v"(acc[g][4]));
          }
        }
      }
#pragma unroll
      for (int g = 0; g < ng; ++g)
#pragma unroll
        for (int jj = 0; jj < 5; ++jj) stage[wv][g * 5 + jj][tq][lane] = mkq[g][jj] & 0x01010101u;
    }
#endif
    // the staged words are read back by the lane that wrote them: no barrier, only the LDS counter
#pragma unroll
    for (int g = 0; g < ng; ++g) {
      {
#pragma unroll
        for (int jj = 0; jj < 5; ++jj) {
          const int slot = s_slot[(g0 + g) * kQG + 5 * half + jj];
          const bool mine = slot >= 0;
          const bool fin = mine && s_fin[max(slot, 0)] != 0;
          if (masks && mine && have) {
            uint4 q = make_uint4(stage[wv][g * 5 + jj][0][lane], stage[wv][g * 5 + jj][1][lane], stage[wv][g * 5 + jj][2][lane],
                                 stage[wv][g * 5 + jj][3][lane]);
            if (!fin) q = make_uint4(0u, 0u, 0u, 0u);
            *reinterpret_cast<uint4 *>(masks + ((size_t)p * M + m0 + slot) * N + n0) = q;
          }
          // sum over the 32 lanes of the half-wave: rows of 16 by DPP butterflies, then row_bcast:15 into rows 1 and 3
          float v = (mine && have) ? -acc[g][jj] : 0.f;
          v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true));
          v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, true));
          v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x141, 0xF, 0xF, true));
          v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x140, 0xF, 0xF, true));
          v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0xA, 0xF, false));
          if (n == 31 && mine) atomicAdd(&part[wv][slot], fin ? v : NAN);
        }
      }
    }
  };
  for (int g0 = 0; g0 * kQG < nv; g0 += kQGP) {
    const int ng = min(kQGP, (nv - g0 * kQG + kQG - 1) / kQG);   // groups in this pass (block-uniform, scalar)
    if (ng == 3) pass(g0, std::integral_constant<int, 3>{});
    else if (ng == 2) pass(g0, std::integral_constant<int, 2>{});
    else pass(g0, std::integral_constant<int, 1>{});
  }
  // empty mask rows of the invalid slots: the two half-waves take alternate slots
  // (walks the set bits of the tile's invalid-slot mask: a per-slot load + branch here cost more than all the arithmetic)
  if (masks) {
    unsigned long long inv = (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane(s_inv[0]) |
                             ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane(s_inv[1]) << 32);
    int k = 0;
    while (inv) {
      const int ml = __builtin_ctzll(inv);
      inv &= inv - 1;
      if (have && (k++ & 1) == half)
        *reinterpret_cast<uint4 *>(masks + ((size_t)p * M + m0 + ml) * N + n0) = make_uint4(0u, 0u, 0u, 0u);
    }
  }
  __syncthreads();
  for (int i = tid; i < mcount; i += kQT) {
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < kQT / 64; ++w) v += part[w][i];
    float *dst = scores + (size_t)p * M + m0 + i;
    if (use_atomic) atomicAdd(dst, v);
    else *dst = v;
  }
}

}  // namespace dr

extern "C" int dr_msac_score_bf16x3_f32(const float *matches, const float *models, const uint8_t *valid, const float *thr,
                                        int P, int M, int N, float *scores, uint8_t *masks, void *stream) {
  DR_REQUIRE(matches && models && thr && scores, "null pointer");
  DR_REQUIRE(P > 0 && M > 0 && N > 0 && P <= 65535 && N % 16 == 0, "bad sizes (N must be a multiple of 16)");
  const int chunks = (N + dr::kQChunk - 1) / dr::kQChunk;
  if (chunks > 1) (void)hipMemsetAsync(scores, 0, sizeof(float) * (size_t)P * M, (hipStream_t)stream);
  hipLaunchKernelGGL(dr::msac_score_bf16x3_kernel, dim3((M + dr::kQTile - 1) / dr::kQTile, chunks, P), dim3(dr::kQT), 0,
                     (hipStream_t)stream, matches, models, valid, thr, M, N, scores, masks, chunks > 1 ? 1 : 0);
  return dr::check_launch("msac_score_bf16x3_kernel");
}
