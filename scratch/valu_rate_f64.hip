// f64 VALU issue rate / dependent-issue latency on gfx950: cycles (s_memtime) per wave-instruction with N independent chains.
#include <hip/hip_runtime.h>
#include <stdint.h>
#define REP8(X) X X X X X X X X
template <int MODE>
__global__ __launch_bounds__(64) void rate(double *out, const double *in, int iters, long long *cyc) {
  double a0 = in[0] + threadIdx.x, a1 = a0 * 1.1, a2 = a0 * 1.2, a3 = a0 * 1.3, a4 = a0 * 1.4, a5 = a0 * 1.5, a6 = a0 * 1.6, a7 = a0 * 1.7;
  double x = in[1], y = in[2];
  unsigned u0 = threadIdx.x, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3;
  long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {  // 8 independent fma chains
#define OP(A) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(A) : "v"(x), "v"(y));
      REP8(OP(a0) OP(a1) OP(a2) OP(a3) OP(a4) OP(a5) OP(a6) OP(a7))
#undef OP
    } else if (MODE == 1) {  // 1 chain (Horner: acc = acc * x + y)
#define OP(A) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(A) : "v"(x), "v"(y));
      REP8(OP(a0) OP(a0) OP(a0) OP(a0) OP(a0) OP(a0) OP(a0) OP(a0))
#undef OP
    } else if (MODE == 2) {  // 2 chains
#define OP(A) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(A) : "v"(x), "v"(y));
      REP8(OP(a0) OP(a1) OP(a0) OP(a1) OP(a0) OP(a1) OP(a0) OP(a1))
#undef OP
    } else if (MODE == 3) {  // 4 chains
#define OP(A) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(A) : "v"(x), "v"(y));
      REP8(OP(a0) OP(a1) OP(a2) OP(a3) OP(a0) OP(a1) OP(a2) OP(a3))
#undef OP
    } else if (MODE == 4) {  // v_add_f64 independent
#define OP(A) asm volatile("v_add_f64 %0, %0, %1" : "+v"(A) : "v"(x));
      REP8(OP(a0) OP(a1) OP(a2) OP(a3) OP(a4) OP(a5) OP(a6) OP(a7))
#undef OP
    } else if (MODE == 5) {  // v_mul_f64 independent
#define OP(A) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(A) : "v"(x));
      REP8(OP(a0) OP(a1) OP(a2) OP(a3) OP(a4) OP(a5) OP(a6) OP(a7))
#undef OP
    } else if (MODE == 6) {  // v_rcp_f64 independent
#define OP(A) asm volatile("v_rcp_f64 %0, %0" : "+v"(A));
      REP8(OP(a0) OP(a1) OP(a2) OP(a3) OP(a4) OP(a5) OP(a6) OP(a7))
#undef OP
    } else if (MODE == 7) {  // v_cndmask_b32 independent
#define OP(A) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(A) : "v"(u3) : );
      REP8(OP(u0) OP(u1) OP(u2) OP(u0) OP(u1) OP(u2) OP(u0) OP(u1))
#undef OP
    } else if (MODE == 8) {  // v_cmp_lt_f64 -> vcc
#define OP(A) asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(A), "v"(x) : "vcc");
      REP8(OP(a0) OP(a1) OP(a2) OP(a3) OP(a4) OP(a5) OP(a6) OP(a7))
#undef OP
    } else if (MODE == 9) {  // f32 fma, 1 chain
      float f = (float)a0, fx = (float)x, fy = (float)y;
#define OP(A) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(A) : "v"(fx), "v"(fy));
      REP8(OP(f) OP(f) OP(f) OP(f) OP(f) OP(f) OP(f) OP(f))
#undef OP
      a0 += f;
    } else if (MODE == 10) {  // f32 fma, 8 chains
      float f0 = (float)a0, f1 = (float)a1, f2 = (float)a2, f3 = (float)a3, f4 = (float)a4, f5 = (float)a5, f6 = (float)a6, f7 = (float)a7, fx = (float)x, fy = (float)y;
#define OP(A) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(A) : "v"(fx), "v"(fy));
      REP8(OP(f0) OP(f1) OP(f2) OP(f3) OP(f4) OP(f5) OP(f6) OP(f7))
#undef OP
      a0 += f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7;
    } else if (MODE == 11) {  // bisection step as compiled: add, mul, 10 fma chain, cmp, 4 cndmask  (one task)
      double lo = a0, hi = a1;
      REP8({ double m = 0.5 * (lo + hi); double fx = a2; _Pragma("unroll") for (int k = 0; k < 10; ++k) fx = fx * m + y; bool l = fx < 0; lo = l ? m : lo; hi = l ? hi : m; })
      a0 = lo; a1 = hi;
    }
  }
  long long t1 = clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + u0 + u1 + u2;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
extern "C" int run(int mode, double *out, const double *in, int iters, long long *cyc, int blocks, void *st) {
  hipStream_t s = (hipStream_t)st;
#define L(M) case M: hipLaunchKernelGGL(rate<M>, dim3(blocks), dim3(64), 0, s, out, in, iters, cyc); break;
  switch (mode) { L(0) L(1) L(2) L(3) L(4) L(5) L(6) L(7) L(8) L(9) L(10) L(11) }
  return (int)hipGetLastError();
}
