// SURVEY 8(f) rank 2 -- the training loss that sits right after the hot path: MatchLoss's clamped symmetric epipolar
// error of EVERY returned model on the ground-truth inlier points (loss.py:107-153, cv_utils.batch_episym :680-695):
//     ys[m,n] = (x2^T M x1)^2 * ( 1/((Mx1)_0^2 + (Mx1)_1^2 + 1e-15) + 1/((M^T x2)_0^2 + (M^T x2)_1^2 + 1e-15) )
//     sums[m] = sum_{n in mask} min(ys[m,n], 1)
// The reference materialises [M, n_in, 3, 3] repeats of the models; here it is the same (model x point) grid as K4:
// points in VGPRs, model coefficients wave-uniform, one partial per (wave, model).  Forward and backward.
#include "dr_common.hpp"

namespace dr {

constexpr int kET = 256, kEP = 8, kEChunk = kET * kEP;
constexpr int kEMW = 2;                          // models a wave evaluates together (their accumulators live in registers)
constexpr int kEMperGroup = (kET / 64) * kEMW;   // 8 models per block and group; a block takes `groups` of them (chosen per launch)
constexpr int kEPw = 16, kEPass = 64 * kEPw;     // a wave holds 16 points per lane per pass
typedef float ev2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ ev2 esplat(float a) { return (ev2){a, a}; }

constexpr int kMeanT = 1024;
// number of non-zero bytes among the 16 of a uint4 (the masks are torch.bool tensors, but the contract is "!= 0")
__device__ __forceinline__ int nonzero_bytes(uint4 v) {
  auto nz = [](uint32_t w) { return __popc((((w & 0x7f7f7f7fu) + 0x7f7f7f7fu) | w) & 0x80808080u); };
  return nz(v.x) + nz(v.y) + nz(v.z) + nz(v.w);
}
// count of non-zero bytes of row[0..n) by one wave: 16-byte loads when the row allows it (every lane's loads are independent:
// the byte-at-a-time loop of the per-pair kernel is a chain of ~30 dependent memory round trips per pair)
__device__ __forceinline__ int wave_count_nonzero(const uint8_t *row, int n, int lane) {
  int c = 0;
  if ((n & 15) == 0 && (reinterpret_cast<uintptr_t>(row) & 15) == 0) {
    const uint4 *r4 = reinterpret_cast<const uint4 *>(row);
    for (int i = lane; i < (n >> 4); i += 64) c += nonzero_bytes(r4[i]);
  } else {
    for (int i = lane; i < n; i += 64) c += row[i] != 0;
  }
  return wave_sum(c);
}
// per-pair means and their mean by the `nwaves` waves of ONE block (fixed reduction order); s_tot: nwaves floats of LDS
__device__ __forceinline__ void match_loss_reduce(const float *__restrict__ sums, const uint8_t *__restrict__ mask,
                                                  const uint8_t *__restrict__ keep, int P, int M, int N, float *__restrict__ per_pair,
                                                  float *__restrict__ coef, float *__restrict__ mean, float *s_tot, int nwaves) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float wave_total = 0.f;
  for (int p = wv; p < P; p += nwaves) {
    float acc = 0.f;
    const float *sp = sums + (size_t)p * M;
    if ((M & 3) == 0 && (reinterpret_cast<uintptr_t>(sp) & 15) == 0) {
      const float4 *s4 = reinterpret_cast<const float4 *>(sp);
      for (int i = lane; i < (M >> 2); i += 64) {
        const float4 v = s4[i];
        acc += (v.x + v.y) + (v.z + v.w);
      }
    } else {
      for (int m = lane; m < M; m += 64) acc += sp[m];
    }
    acc = wave_sum(acc);
    const int n_in = mask ? wave_count_nonzero(mask + (size_t)p * N, N, lane) : N;
    const int n_kept = keep ? wave_count_nonzero(keep + (size_t)p * M, M, lane) : M;
    const float den = fmaxf((float)n_in * (float)n_kept, 1.0f);
    if (lane == 0) {
      per_pair[p] = acc / den;
      coef[p] = 1.0f / den;
    }
    wave_total += acc / den;
  }
  if (lane == 0) s_tot[wv] = wave_total;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < nwaves; ++w) t += s_tot[w];
    mean[0] = t / (float)P;
  }
}


// v4 (v2 = mask compaction + packed f32 + v_rcp_f32; v3 = a wave takes its models over ALL compacted points, one DPP
// reduction per accumulator and wave).  The points selected by the mask (the GT inliers: half of the points at C2) are
// compacted per block, so no lane evaluates a point whose weight is 0.  v3 ran the compaction (mask bytes, wave scans, three
// barriers, the gathered point loads) once per EIGHT models -- 4096 blocks at the train shape, each with ~2 k cycles of
// arithmetic behind ~2 us of prologue: 36 us forward + 59 us backward for 33 M (model, point) evaluations.  Now a block owns
// 8 x `groups` models: every wave takes `groups` groups of two, and when the selected points fit one pass (<= 1024 of <= 2048:
// the training shape) they are loaded ONCE and stay in VGPRs for all groups.  `groups` is chosen per launch so that the grid
// is one round of resident waves where the shape allows it (episym_groups below).
// kMode 0 = forward (sums), 1 = backward (scaled gradient), 2 = round 5: BOTH in one pass -- the loss is a scalar mean, so its
// gradient w.r.t. a model is (one number per pair) x a quantity the forward can write while it holds the residuals in registers:
// out = the UNSCALED gradient d sums[p,m] / d M [P,M,9], out2 = sums [P,M].  The training step then walks the (model x point) grid
// once instead of twice (dr_match_loss_fused_f32 / dr_match_loss_scale_f32 below).  (Built first with the per-pair reduction in the
// SAME launch -- the block that takes the last ticket reduces: 166 us instead of 82 for forward + backward, because the agent-scope
// release every block needs before its ticket is an L2 write-back on a part with eight L2s; the reduction stays a launch of its own.)
template <int kMode>
__global__ __launch_bounds__(kET) void episym_kernel(const float *__restrict__ matches, const uint8_t *__restrict__ mask,
                                                     const float *__restrict__ models, const uint8_t *__restrict__ valid,
                                                     const float *__restrict__ grad_sums, int grad_per_pair, int M,
                                                     int N, float *__restrict__ out, int groups,
                                                     const float *__restrict__ grad_scalar, float scalar_scale,
                                                     float *__restrict__ out2 = nullptr) {
  // forward: out = sums [P,M]; backward: out = grad_models [P,M,9]
  constexpr bool kBackward = kMode != 0;
  constexpr int kV = kMode == 0 ? 1 : (kMode == 1 ? 9 : 10);
  __shared__ int s_list[kEChunk];
  __shared__ int s_wave[kET / 64];
  const int p = blockIdx.z, m0 = blockIdx.x * kEMperGroup * groups;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: the model coefficients below live in SGPRs
  const float *mt = matches + (size_t)p * N * 4;
  const uint8_t *mk = mask ? mask + (size_t)p * N : nullptr;

  // compaction of one 2048-point chunk: s_list[0..T) = indices of the selected points, ascending (block-wide, three barriers)
  auto compact = [&](int c0) -> int {
    int T = min(kEChunk, N - c0);
    __syncthreads();
    if (mk) {
      int cnt = 0;
      uint32_t bits = 0;
#pragma unroll
      for (int j = 0; j < kEP; ++j) {
        const int n = c0 + tid * kEP + j;
        const bool on = n < N && mk[n] != 0;
        bits |= on ? (1u << j) : 0u;
        cnt += on ? 1 : 0;
      }
      int inc = cnt;   // inclusive scan inside the wave
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(inc, o, 64);
        if (lane >= o) inc += v;
      }
      if (lane == 63) s_wave[tid >> 6] = inc;
      __syncthreads();
      int base = 0;
      for (int w = 0; w < (tid >> 6); ++w) base += s_wave[w];
      T = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
      int pos = base + inc - cnt;
#pragma unroll
      for (int j = 0; j < kEP; ++j)
        if ((bits >> j) & 1u) s_list[pos++] = c0 + tid * kEP + j;
      __syncthreads();
    }
    return T;
  };

  ev2 x1[kEPw / 2], y1[kEPw / 2], x2[kEPw / 2], y2[kEPw / 2], w[kEPw / 2];
  auto load_points = [&](int c0, int p0, int T) {
#pragma unroll
    for (int j = 0; j < kEPw; ++j) {
      const int pos = p0 + j * 64 + lane;
      const bool have = pos < T;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (have) v = reinterpret_cast<const float4 *>(mt)[mk ? s_list[pos] : c0 + pos];
      x1[j / 2][j & 1] = v.x; y1[j / 2][j & 1] = v.y; x2[j / 2][j & 1] = v.z; y2[j / 2][j & 1] = v.w;
      w[j / 2][j & 1] = have ? 1.f : 0.f;
    }
  };

  const bool single = N <= kEChunk;          // one chunk: compacted once for all model groups
  int T0 = 0;
  if (single) T0 = compact(0);
  const bool resident = single && T0 <= kEPass;   // ... and one pass: the points stay in registers
  if (resident) load_points(0, 0, T0);

#pragma unroll 1
  for (int g = 0; g < groups; ++g) {
    const int mg = m0 + (wv * groups + g) * kEMW;    // this wave's models mg .. mg + kEMW - 1 (wave-uniform)
    ev2 acc[kEMW][kV];
#pragma unroll
    for (int mi = 0; mi < kEMW; ++mi)
#pragma unroll
      for (int q = 0; q < kV; ++q) acc[mi][q] = esplat(0.f);
    float mcoef[kEMW][9];
    bool mlive[kEMW];
#pragma unroll
    for (int mi = 0; mi < kEMW; ++mi) {
      const int m = mg + mi;
      mlive[mi] = m < M && (!valid || valid[(size_t)p * M + m] != 0);
#pragma unroll
      for (int q = 0; q < 9; ++q) mcoef[mi][q] = mlive[mi] ? models[((size_t)p * M + m) * 9 + q] : 0.f;
    }
    for (int c0 = 0; c0 < N; c0 += kEChunk) {
      const int T = single ? T0 : compact(c0);          // (several chunks: compacted again per group -- long rows only)
      for (int p0 = 0; p0 < T; p0 += kEPass) {
        if (!resident) load_points(c0, p0, T);
#pragma unroll
        for (int mi = 0; mi < kEMW; ++mi) {
          if (!mlive[mi]) continue;   // wave-uniform
          const float(&m)[9] = mcoef[mi];
#pragma unroll
          for (int j = 0; j < kEPw / 2; ++j) {
            if (p0 + 2 * j * 64 >= T) break;   // wave-uniform: the positions of this and all later pairs are empty
            const ev2 a0 = x2[j] * esplat(m[0]) + (y2[j] * esplat(m[3]) + esplat(m[6]));
            const ev2 a1 = x2[j] * esplat(m[1]) + (y2[j] * esplat(m[4]) + esplat(m[7]));
            const ev2 a2 = x2[j] * esplat(m[2]) + (y2[j] * esplat(m[5]) + esplat(m[8]));
            const ev2 b0 = x1[j] * esplat(m[0]) + (y1[j] * esplat(m[1]) + esplat(m[2]));
            const ev2 b1 = x1[j] * esplat(m[3]) + (y1[j] * esplat(m[4]) + esplat(m[5]));
            const ev2 r = x1[j] * a0 + (y1[j] * a1 + a2);
            const ev2 db = b0 * b0 + (b1 * b1 + esplat(1e-15f)), da = a0 * a0 + (a1 * a1 + esplat(1e-15f));
            ev2 ib, ia;
            ib[0] = __builtin_amdgcn_rcpf(db[0]); ib[1] = __builtin_amdgcn_rcpf(db[1]);
            ia[0] = __builtin_amdgcn_rcpf(da[0]); ia[1] = __builtin_amdgcn_rcpf(da[1]);
            const ev2 rr = r * r, s = ib + ia;
            const ev2 ys = rr * s;
            if (kMode != 1) {
              ev2 cl;
              cl[0] = fminf(ys[0], 1.0f); cl[1] = fminf(ys[1], 1.0f);
              acc[mi][kMode == 0 ? 0 : 9] = cl * w[j] + acc[mi][kMode == 0 ? 0 : 9];
            }
            if (kBackward) {
              ev2 live;   // the clamp passes no gradient at or above 1
              live[0] = ys[0] < 1.0f ? w[j][0] : 0.f;
              live[1] = ys[1] < 1.0f ? w[j][1] : 0.f;
              // d ys = c1 dr - cb (b0 db0 + b1 db1) - ca (a0 da0 + a1 da1), dr = x2^T dM x1, db = (dM x1)_{0,1}, da = (dM^T x2)_{0,1}
              const ev2 r2l = (r + r) * live;
              const ev2 c1 = r2l * s, cb = r2l * r * (ib * ib), ca = r2l * r * (ia * ia);
              const ev2 u0 = c1 * x2[j] - cb * b0, u1 = c1 * y2[j] - cb * b1, v0 = ca * a0, v1 = ca * a1;
              acc[mi][0] = acc[mi][0] + (x1[j] * u0 - v0 * x2[j]);
              acc[mi][1] = acc[mi][1] + (y1[j] * u0 - v1 * x2[j]);
              acc[mi][2] = acc[mi][2] + u0;
              acc[mi][3] = acc[mi][3] + (x1[j] * u1 - v0 * y2[j]);
              acc[mi][4] = acc[mi][4] + (y1[j] * u1 - v1 * y2[j]);
              acc[mi][5] = acc[mi][5] + u1;
              acc[mi][6] = acc[mi][6] + (x1[j] * c1 - v0);
              acc[mi][7] = acc[mi][7] + (y1[j] * c1 - v1);
              acc[mi][8] = acc[mi][8] + c1;
            }
          }
        }
      }
    }
#pragma unroll
    for (int mi = 0; mi < kEMW; ++mi) {
      const int m = mg + mi;
      if (m >= M) continue;
      // backward: d loss / d sums[p,m] = grad_sums (per model or per pair) x, for the fused mean, the upstream scalar / P
      const float gs = kMode == 1 ? grad_sums[grad_per_pair ? (size_t)p : (size_t)p * M + m] *
                                        (grad_scalar ? grad_scalar[0] * scalar_scale : 1.f)
                                  : 1.f;
#pragma unroll
      for (int q = 0; q < kV; ++q) {
        const float v = wave_sum_lane63(acc[mi][q][0] + acc[mi][q][1]);
        if (kMode == 2 && q == 9) {
          if (lane == 63) out2[(size_t)p * M + m] = v;
        } else if (lane == 63) {
          out[((size_t)p * M + m) * (kMode == 0 ? 1 : 9) + q] = v * gs;
        }
      }
    }
  }
}

// MatchLoss's reduction over the models of a pair (loss.py:146-153 with the per-pair means written out):
//     per_pair[p] = sum_m sums[p,m] / max(n_in[p] * n_models[p], 1),   coef[p] = 1 / max(n_in[p] * n_models[p], 1)
// n_in = number of selected points (mask; NULL = N), n_models = number of kept models (keep; NULL = M).  One block per
// pair; every reduction is in a fixed order (lane partials, DPP wave sum, waves in order), so the loss is reproducible.
__global__ __launch_bounds__(kET) void match_loss_pair_kernel(const float *__restrict__ sums, const uint8_t *__restrict__ mask,
                                                              const uint8_t *__restrict__ keep, int M, int N,
                                                              float *__restrict__ per_pair, float *__restrict__ coef) {
  __shared__ float s_sum[kET / 64];
  __shared__ int s_in[kET / 64], s_kept[kET / 64];
  const int p = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  float acc = 0.f;
  int n_in = 0, n_kept = 0;
  for (int m = tid; m < M; m += kET) {
    acc += sums[(size_t)p * M + m];
    if (keep) n_kept += keep[(size_t)p * M + m] != 0;
  }
  if (mask)
    for (int n = tid; n < N; n += kET) n_in += mask[(size_t)p * N + n] != 0;
  acc = wave_sum_lane63(acc);
  n_in = wave_sum(n_in);
  n_kept = wave_sum(n_kept);
  if (lane == 63) s_sum[wv] = acc;
  if (lane == 0) { s_in[wv] = n_in; s_kept[wv] = n_kept; }
  __syncthreads();
  if (tid == 0) {
    float total = 0.f;
    int in = 0, kept = 0;
    for (int w = 0; w < kET / 64; ++w) { total += s_sum[w]; in += s_in[w]; kept += s_kept[w]; }
    const float den = fmaxf((float)(mask ? in : N) * (float)(keep ? kept : M), 1.0f);
    per_pair[p] = total / den;
    coef[p] = 1.0f / den;
  }
}

// MatchLoss down to the scalar (loss.py:146-153 incl. the mean over pairs): the per-pair kernel above and the mean of its
// P results in ONE launch -- a 1024-thread block whose sixteen waves take the pairs in turn (a pair is M sums + N mask bytes:
// nothing), fixed reduction order.  Replaces, per training step, a torch mean kernel and the two small kernels of its
// backward; for P <= 64 (a rank's pairs), beyond that the per-pair kernel + torch.mean stay.
__global__ __launch_bounds__(kMeanT) void match_loss_mean_kernel(const float *__restrict__ sums, const uint8_t *__restrict__ mask,
                                                                const uint8_t *__restrict__ keep, int P, int M, int N,
                                                                float *__restrict__ per_pair, float *__restrict__ coef,
                                                                float *__restrict__ mean) {
  __shared__ float s_tot[kMeanT / 64];
  match_loss_reduce(sums, mask, keep, P, M, N, per_pair, coef, mean, s_tot, kMeanT / 64);
}

__global__ __launch_bounds__(kET) void match_loss_scale_kernel(const float *__restrict__ grad_unscaled, const float *__restrict__ coef,
                                                              const float *__restrict__ grad_mean, float inv_pairs, int per_pair,
                                                              size_t total, float *__restrict__ grad_models) {
  // grad_models[p, m, q] = grad_unscaled[p, m, q] * coef[p] * (upstream scalar) / P
  const size_t i = (size_t)blockIdx.x * kET + threadIdx.x;
  if (i < total) grad_models[i] = grad_unscaled[i] * (coef[i / per_pair] * grad_mean[0] * inv_pairs);
}

}  // namespace dr

namespace dr {
// groups per block: the smallest count for which all waves of the launch are resident at once (`per_simd` = waves per SIMD the
// callers pass; 4 for both directions: the backward kernel holds three, but six groups per block measured 61 us against 56 us with four), clamped to [1, 16]
static int episym_groups(int P, int M, int per_simd) {
  static int simds = 0;
  if (!simds) {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    simds = 4 * (cus > 0 ? cus : 256);
  }
  const long wave_models = ((long)P * M + kEMW - 1) / kEMW;       // (wave, group) work items
  const long capacity = (long)simds * per_simd;
  long g = (wave_models + capacity - 1) / capacity;
  return (int)(g < 1 ? 1 : (g > 16 ? 16 : g));
}
}  // namespace dr

extern "C" {

int dr_episym_fwd_f32(const float *matches, const uint8_t *mask, const float *models, const uint8_t *valid, int P, int M,
                      int N, float *sums, void *stream) {
  DR_REQUIRE(matches && models && sums, "null pointer");
  DR_REQUIRE(P > 0 && M > 0 && N > 0 && P <= 65535, "bad sizes");
  const int groups = dr::episym_groups(P, M, 4), per_block = dr::kEMperGroup * groups;
  hipLaunchKernelGGL((dr::episym_kernel<0>), dim3((M + per_block - 1) / per_block, 1, P), dim3(dr::kET), 0,
                     (hipStream_t)stream, matches, mask, models, valid, (const float *)nullptr, 0, M, N, sums, groups,
                     (const float *)nullptr, 1.0f);
  return dr::check_launch("episym_kernel");
}

int dr_episym_bwd_f32(const float *matches, const uint8_t *mask, const float *models, const uint8_t *valid,
                      const float *grad_sums, int P, int M, int N, float *grad_models, void *stream) {
  DR_REQUIRE(matches && models && grad_sums && grad_models, "null pointer");
  DR_REQUIRE(P > 0 && M > 0 && N > 0 && P <= 65535, "bad sizes");
  const int groups = dr::episym_groups(P, M, 4), per_block = dr::kEMperGroup * groups;
  hipLaunchKernelGGL((dr::episym_kernel<1>), dim3((M + per_block - 1) / per_block, 1, P), dim3(dr::kET), 0,
                     (hipStream_t)stream, matches, mask, models, valid, grad_sums, 0, M, N, grad_models, groups,
                     (const float *)nullptr, 1.0f);
  return dr::check_launch("episym_kernel");
}

int dr_episym_bwd_pair_f32(const float *matches, const uint8_t *mask, const float *models, const uint8_t *valid,
                           const float *grad_pair, int P, int M, int N, float *grad_models, void *stream) {
  DR_REQUIRE(matches && models && grad_pair && grad_models, "null pointer");
  DR_REQUIRE(P > 0 && M > 0 && N > 0 && P <= 65535, "bad sizes");
  const int groups = dr::episym_groups(P, M, 4), per_block = dr::kEMperGroup * groups;
  hipLaunchKernelGGL((dr::episym_kernel<1>), dim3((M + per_block - 1) / per_block, 1, P), dim3(dr::kET), 0,
                     (hipStream_t)stream, matches, mask, models, valid, grad_pair, 1, M, N, grad_models, groups,
                     (const float *)nullptr, 1.0f);
  return dr::check_launch("episym_kernel");
}

int dr_episym_bwd_mean_f32(const float *matches, const uint8_t *mask, const float *models, const uint8_t *valid,
                           const float *coef, const float *grad_mean, int P, int M, int N, float *grad_models, void *stream) {
  DR_REQUIRE(matches && models && coef && grad_mean && grad_models, "null pointer");
  DR_REQUIRE(P > 0 && M > 0 && N > 0 && P <= 65535, "bad sizes");
  const int groups = dr::episym_groups(P, M, 4), per_block = dr::kEMperGroup * groups;
  hipLaunchKernelGGL((dr::episym_kernel<1>), dim3((M + per_block - 1) / per_block, 1, P), dim3(dr::kET), 0,
                     (hipStream_t)stream, matches, mask, models, valid, coef, 1, M, N, grad_models, groups, grad_mean,
                     1.0f / (float)P);
  return dr::check_launch("episym_kernel");
}

/* Round 5: MatchLoss (loss.py:107-153) value + gradient in one pass over the (model x point) grid.  dr_match_loss_fused_f32: sums
 * [P,M], per_pair [P], coef [P], mean [1] as dr_episym_fwd + dr_match_loss_mean produce them (the second launch is issued here), plus
 * grad_unscaled [P,M,9] = d sums[p,m] / d model (invalid slots 0).  dr_match_loss_scale_f32: grad_models = grad_unscaled x coef[p] x
 * grad_mean / P -- the whole backward of the loss. */
int dr_match_loss_fused_f32(const float *matches, const uint8_t *mask, const float *models, const uint8_t *valid, int P, int M, int N,
                            float *sums, float *grad_unscaled, float *per_pair, float *coef, float *mean, void *stream) {
  DR_REQUIRE(matches && models && sums && grad_unscaled && per_pair && coef && mean, "null pointer");
  DR_REQUIRE(P > 0 && M > 0 && N > 0 && P <= 65535, "bad sizes");
  const int groups = dr::episym_groups(P, M, 4), per_block = dr::kEMperGroup * groups;
  hipLaunchKernelGGL((dr::episym_kernel<2>), dim3((M + per_block - 1) / per_block, 1, P), dim3(dr::kET), 0, (hipStream_t)stream,
                     matches, mask, models, valid, (const float *)nullptr, 0, M, N, grad_unscaled, groups, (const float *)nullptr,
                     1.0f, sums);
  if (int rc = dr::check_launch("episym_kernel<fused>")) return rc;
  hipLaunchKernelGGL(dr::match_loss_mean_kernel, dim3(1), dim3(dr::kMeanT), 0, (hipStream_t)stream, sums, mask, valid, P, M, N,
                     per_pair, coef, mean);
  return dr::check_launch("match_loss_mean_kernel");
}

int dr_match_loss_scale_f32(const float *grad_unscaled, const float *coef, const float *grad_mean, int P, int M, float *grad_models,
                            void *stream) {
  DR_REQUIRE(grad_unscaled && coef && grad_mean && grad_models, "null pointer");
  DR_REQUIRE(P > 0 && M > 0, "bad sizes");
  const size_t total = (size_t)P * M * 9;
  hipLaunchKernelGGL(dr::match_loss_scale_kernel, dim3((unsigned)((total + dr::kET - 1) / dr::kET)), dim3(dr::kET), 0,
                     (hipStream_t)stream, grad_unscaled, coef, grad_mean, 1.0f / (float)P, M * 9, total, grad_models);
  return dr::check_launch("match_loss_scale_kernel");
}

int dr_match_loss_mean_f32(const float *sums, const uint8_t *mask, const uint8_t *keep, int P, int M, int N,
                           float *per_pair, float *coef, float *mean, void *stream) {
  DR_REQUIRE(sums && per_pair && coef && mean, "null pointer");
  DR_REQUIRE(P > 0 && M > 0 && N > 0, "bad sizes");
  hipLaunchKernelGGL(dr::match_loss_mean_kernel, dim3(1), dim3(dr::kMeanT), 0, (hipStream_t)stream, sums, mask, keep, P, M, N,
                     per_pair, coef, mean);
  return dr::check_launch("match_loss_mean_kernel");
}

int dr_match_loss_pair_f32(const float *sums, const uint8_t *mask, const uint8_t *keep, int P, int M, int N,
                           float *per_pair, float *coef, void *stream) {
  DR_REQUIRE(sums && per_pair && coef, "null pointer");
  DR_REQUIRE(P > 0 && M > 0 && N > 0, "bad sizes");
  hipLaunchKernelGGL(dr::match_loss_pair_kernel, dim3(P), dim3(dr::kET), 0, (hipStream_t)stream, sums, mask, keep, M, N,
                     per_pair, coef);
  return dr::check_launch("match_loss_pair_kernel");
}

}  // extern "C"
