// Issue-rate micro-benchmark 2 (gfx950): integer multiplies, transcendentals, f64 ops.  8 independent chains, 1 wave/SIMD.
#include <hip/hip_runtime.h>
#include <stdint.h>
#define REP8(X) X X X X X X X X
template <int MODE>
__global__ __launch_bounds__(256) void rate(float *out, const float *in, int iters) {
  uint32_t u0 = threadIdx.x + 1, u1 = u0 * 3, u2 = u0 * 5, u3 = u0 * 7, u4 = u0 * 11, u5 = u0 * 13, u6 = u0 * 17, u7 = u0 * 19;
  uint32_t k = (uint32_t)in[0] | 0xD2511F53u;
  float f0 = in[1] + threadIdx.x, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3, f4 = f0 + 4, f5 = f0 + 5, f6 = f0 + 6, f7 = f0 + 7;
  double d0 = f0, d1 = f1, d2 = f2, d3 = f3, d4 = f4, d5 = f5, d6 = f6, d7 = f7, dx = in[2], dy = in[3];
  uint64_t w0 = u0, w1 = u1, w2 = u2, w3 = u3, w4 = u4, w5 = u5, w6 = u6, w7 = u7;
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {
#define OP(A) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(A) : "v"(k));
      REP8(OP(u0) OP(u1) OP(u2) OP(u3) OP(u4) OP(u5) OP(u6) OP(u7))
#undef OP
    } else if (MODE == 1) {
#define OP(A) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(A) : "v"(k));
      REP8(OP(u0) OP(u1) OP(u2) OP(u3) OP(u4) OP(u5) OP(u6) OP(u7))
#undef OP
    } else if (MODE == 2) {
#define OP(A) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(A) : "v"(k), "v"(u0) : "vcc");
      REP8(OP(w0) OP(w1) OP(w2) OP(w3) OP(w4) OP(w5) OP(w6) OP(w7))
#undef OP
    } else if (MODE == 3) {
#define OP(A) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(A) : "v"(k));
      REP8(OP(u0) OP(u1) OP(u2) OP(u3) OP(u4) OP(u5) OP(u6) OP(u7))
#undef OP
    } else if (MODE == 4) {
#define OP(A) asm volatile("v_log_f32 %0, %0" : "+v"(A));
      REP8(OP(f0) OP(f1) OP(f2) OP(f3) OP(f4) OP(f5) OP(f6) OP(f7))
#undef OP
    } else if (MODE == 5) {
#define OP(A) asm volatile("v_exp_f32 %0, %0" : "+v"(A));
      REP8(OP(f0) OP(f1) OP(f2) OP(f3) OP(f4) OP(f5) OP(f6) OP(f7))
#undef OP
    } else if (MODE == 6) {
#define OP(A) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(A) : "v"(dx), "v"(dy));
      REP8(OP(d0) OP(d1) OP(d2) OP(d3) OP(d4) OP(d5) OP(d6) OP(d7))
#undef OP
    } else if (MODE == 7) {
#define OP(A) asm volatile("v_rcp_f64 %0, %0" : "+v"(A));
      REP8(OP(d0) OP(d1) OP(d2) OP(d3) OP(d4) OP(d5) OP(d6) OP(d7))
#undef OP
    } else if (MODE == 8) {
#define OP(A) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(A) : "v"(k));
      REP8(OP(u0) OP(u1) OP(u2) OP(u3) OP(u4) OP(u5) OP(u6) OP(u7))
#undef OP
    } else if (MODE == 9) {
#define OP(A) asm volatile("v_alignbit_b32 %0, %0, %0, 13" : "+v"(A));
      REP8(OP(u0) OP(u1) OP(u2) OP(u3) OP(u4) OP(u5) OP(u6) OP(u7))
#undef OP
    } else if (MODE == 10) {  // dependent f64 fma chain (latency): ONE accumulator
#define OP(A) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(A) : "v"(dx), "v"(dy));
      REP8(OP(d0) OP(d0) OP(d0) OP(d0) OP(d0) OP(d0) OP(d0) OP(d0))
#undef OP
    } else if (MODE == 11) {  // dependent pk_fma f32 chain (latency)
#define OP(A) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(A) : "v"(f1), "v"(f2));
      REP8(OP(f0) OP(f0) OP(f0) OP(f0) OP(f0) OP(f0) OP(f0) OP(f0))
#undef OP
    } else if (MODE == 12) {
#define OP(A) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(A) : "v"(dx));
      REP8(OP(d0) OP(d1) OP(d2) OP(d3) OP(d4) OP(d5) OP(d6) OP(d7))
#undef OP
    } else if (MODE == 13) {
#define OP(A) asm volatile("v_add_f64 %0, %0, %1" : "+v"(A) : "v"(dx));
      REP8(OP(d0) OP(d1) OP(d2) OP(d3) OP(d4) OP(d5) OP(d6) OP(d7))
#undef OP
    } else if (MODE == 14) {  // 64-bit select = 2 x v_cndmask_b32
#define OP(A) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(A) : "v"(k) : "vcc");
      REP8(OP(u0) OP(u1) OP(u2) OP(u3) OP(u4) OP(u5) OP(u6) OP(u7))
#undef OP
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = (float)(u0 + u1 + u2 + u3 + u4 + u5 + u6 + u7) + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7 +
      (float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7) + (float)(w0 + w1 + w2 + w3 + w4 + w5 + w6 + w7);
}
extern "C" int run(int mode, float *out, const float *in, int iters, int blocks, int threads, void *st) {
  hipStream_t s = (hipStream_t)st;
#define L(M) case M: hipLaunchKernelGGL(rate<M>, dim3(blocks), dim3(threads), 0, s, out, in, iters); break;
  switch (mode) { L(0) L(1) L(2) L(3) L(4) L(5) L(6) L(7) L(8) L(9) L(10) L(11) L(12) L(13) L(14) }
  return (int)hipGetLastError();
}
