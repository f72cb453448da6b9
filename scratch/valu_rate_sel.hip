// select-class VALU instructions on gfx950, one wave per SIMD: cycles per wave-instruction
#include <hip/hip_runtime.h>
#include <stdint.h>
#define REP8(X) X X X X X X X X
template <int MODE>
__global__ __launch_bounds__(64) void rate(unsigned *out, const unsigned *in, int iters, long long *cyc) {
  unsigned u0 = in[0] + threadIdx.x, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3, u4 = u0 + 4, u5 = u0 + 5, u6 = u0 + 6, u7 = u0 + 7, x = in[1], m = in[2];
  double d0 = u0, d1 = u1;
  unsigned long long sm = __ballot(threadIdx.x & 1);
  long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {
#define OP(A) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(A) : "v"(x));
      REP8(OP(u0) OP(u1) OP(u2) OP(u3) OP(u4) OP(u5) OP(u6) OP(u7))
#undef OP
    } else if (MODE == 1) {
#define OP(A) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(A) : "v"(x), "s"(sm));
      REP8(OP(u0) OP(u1) OP(u2) OP(u3) OP(u4) OP(u5) OP(u6) OP(u7))
#undef OP
    } else if (MODE == 2) {
#define OP(A) asm volatile("v_bfi_b32 %0, %1, %2, %0" : "+v"(A) : "v"(m), "v"(x));
      REP8(OP(u0) OP(u1) OP(u2) OP(u3) OP(u4) OP(u5) OP(u6) OP(u7))
#undef OP
    } else if (MODE == 3) {
#define OP(A) asm volatile("v_mov_b32 %0, %1" : "=v"(A) : "v"(x));
      REP8(OP(u0) OP(u1) OP(u2) OP(u3) OP(u4) OP(u5) OP(u6) OP(u7))
#undef OP
    } else if (MODE == 4) {  // cmp writes vcc, cndmask reads it
#define OP(A) asm volatile("v_cmp_lt_f64 vcc, %1, %2\n v_cndmask_b32 %0, %0, %3, vcc" : "+v"(A) : "v"(d0), "v"(d1), "v"(x) : "vcc");
      REP8(OP(u0) OP(u1) OP(u2) OP(u3))
#undef OP
    } else if (MODE == 5) {  // cmp writes an SGPR pair, four cndmasks read it (the bisection update)
#define OP(A, B, C, D) asm volatile("v_cmp_lt_f64 s[20:21], %4, %5\n v_cndmask_b32 %0, %0, %6, s[20:21]\n v_cndmask_b32 %1, %1, %6, s[20:21]\n v_cndmask_b32 %2, %6, %2, s[20:21]\n v_cndmask_b32 %3, %6, %3, s[20:21]" : "+v"(A), "+v"(B), "+v"(C), "+v"(D) : "v"(d0), "v"(d1), "v"(x) : "s20", "s21");
      REP8(OP(u0, u1, u2, u3) OP(u4, u5, u6, u7))
#undef OP
    } else if (MODE == 6) {
#define OP(A) asm volatile("v_and_b32 %0, %0, %1" : "+v"(A) : "v"(x));
      REP8(OP(u0) OP(u1) OP(u2) OP(u3) OP(u4) OP(u5) OP(u6) OP(u7))
#undef OP
    } else if (MODE == 7) {  // v_max_f64 / v_min_f64 class
#define OP(A) asm volatile("v_max_f64 %0, %0, %1" : "+v"(A) : "v"(d1));
      REP8(OP(d0) OP(d0) OP(d0) OP(d0) OP(d0) OP(d0) OP(d0) OP(d0))
#undef OP
    }
  }
  long long t1 = clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = u0 + u1 + u2 + u3 + u4 + u5 + u6 + u7 + (unsigned)d0;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
extern "C" int run(int mode, unsigned *out, const unsigned *in, int iters, long long *cyc, int blocks, void *st) {
  hipStream_t s = (hipStream_t)st;
#define L(M) case M: hipLaunchKernelGGL(rate<M>, dim3(blocks), dim3(64), 0, s, out, in, iters, cyc); break;
  switch (mode) { L(0) L(1) L(2) L(3) L(4) L(5) L(6) L(7) }
  return (int)hipGetLastError();
}
