// Micro-benchmark of the K4 filter's inner loop: per 16-point tile two v_mfma_f32_16x16x32_f16 + the candidate test
// (4 fma, max, max3, cmp, select, or), 16 tiles per pass, register-resident operands.  MODE: 0 both, 1 matrix only, 2 vector only.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
#ifndef MODE
#define MODE 0
#endif
#ifndef G
#define G 2
#endif
__global__ __launch_bounds__(512) void probe(const float *in, uint32_t *out, int reps, unsigned long long *cyc) {
  const int lane = threadIdx.x & 63;
  h8 Ar[16], Aj[16];
  for (int t = 0; t < 16; ++t)
    for (int q = 0; q < 8; ++q) { Ar[t][q] = (_Float16)in[(lane * 16 + t + q) & 1023]; Aj[t][q] = (_Float16)in[(lane * 7 + t * 3 + q) & 1023]; }
  h8 Br, Bj;
  for (int q = 0; q < 8; ++q) { Br[q] = (_Float16)in[(lane + q) & 1023]; Bj[q] = (_Float16)in[(lane * 3 + q) & 1023]; }
  f4 Cr = {in[lane], in[lane], in[lane], in[lane]}, Cj = {in[lane + 1], in[lane + 1], in[lane + 1], in[lane + 1]};
  uint32_t acc = 0;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int r = 0; r < reps; ++r) {
    uint32_t cbits = 0;
    constexpr int NG = 16 / G;
    f4 Dr[2][G], Dj[2][G];
    auto issue = [&](int k, int set) {
#pragma unroll
      for (int u = 0; u < G; ++u) {
        if (MODE == 2) { Dr[set][u] = Cr * (float)(r + k); Dj[set][u] = Cj + (float)u; }
        else {
          Dr[set][u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ar[k * G + u], Br, Cr, 0, 0, 0);
          Dj[set][u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Aj[k * G + u], Bj, Cj, 0, 0, 0);
        }
      }
    };
    issue(0, 0);
#pragma unroll
    for (int k = 0; k < NG; ++k) {
      if (k + 1 < NG) issue(k + 1, (k + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < G; ++u) {
        const f4 r_ = Dr[k & 1][u], j_ = Dj[k & 1][u];
        if (MODE == 1) { cbits |= (r_[0] + j_[0] > 1e30f) ? (1u << (k * G + u)) : 0u; continue; }
        const float d0 = fmaf(-r_[0], r_[0], j_[0]), d1 = fmaf(-r_[1], r_[1], j_[1]);
        const float d2 = fmaf(-r_[2], r_[2], j_[2]), d3 = fmaf(-r_[3], r_[3], j_[3]);
        const float mx = fmaxf(fmaxf(fmaxf(d0, d1), d2), d3);
        cbits |= (mx >= 0.f) ? (1u << (k * G + u)) : 0u;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    acc += cbits;
    Cr[0] += 1e-9f * (float)(cbits & 1);   // loop-carried: the compiler cannot hoist the pass
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
extern "C" int run_probe(const float *in, uint32_t *out, int reps, unsigned long long *cyc, int threads, void *stream) {
  hipLaunchKernelGGL(probe, dim3(256), dim3(threads), 0, (hipStream_t)stream, in, out, reps, cyc);
  return (int)hipGetLastError();
}
