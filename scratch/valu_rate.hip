// VALU issue-rate micro-benchmark (gfx950): cycles per wave-instruction for the op mix of K4.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float v2f __attribute__((ext_vector_type(2)));
#define REP8(X) X X X X X X X X
template <int MODE>
__global__ __launch_bounds__(256) void rate(float *out, const float *in, int iters, long long *cyc) {
  v2f a0 = {in[0] + threadIdx.x, 1.f}, a1 = a0 * 1.1f, a2 = a0 * 1.2f, a3 = a0 * 1.3f, a4 = a0 * 1.4f, a5 = a0 * 1.5f,
      a6 = a0 * 1.6f, a7 = a0 * 1.7f;
  v2f x = {in[1], in[2]}, y = {in[3], in[4]};
  v2f sx;  // wave-uniform pair in SGPRs
  sx[0] = __builtin_amdgcn_readfirstlane(in[5]); sx[1] = __builtin_amdgcn_readfirstlane(in[6]);
  float f0 = a0[0], f1 = a1[0], f2 = a2[0], f3 = a3[0], f4 = a4[0], f5 = a5[0], f6 = a6[0], f7 = a7[0];
  long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {  // pk_fma, three VGPR pairs
#define OP(A) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(A) : "v"(x), "v"(y));
      REP8(OP(a0) OP(a1) OP(a2) OP(a3) OP(a4) OP(a5) OP(a6) OP(a7))
#undef OP
    } else if (MODE == 1) {  // pk_fma, one SGPR pair source
#define OP(A) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(A) : "v"(x), "s"(sx));
      REP8(OP(a0) OP(a1) OP(a2) OP(a3) OP(a4) OP(a5) OP(a6) OP(a7))
#undef OP
    } else if (MODE == 2) {  // scalar fma, VGPRs (same 64 ops = half the flops)
#define OP(A) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(A) : "v"(x[0]), "v"(y[0]));
      REP8(OP(f0) OP(f1) OP(f2) OP(f3) OP(f4) OP(f5) OP(f6) OP(f7))
#undef OP
    } else if (MODE == 3) {  // pk_fma with op_sel broadcast of a VGPR half (what splat(m) compiles to)
#define OP(A) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(A) : "v"(x), "v"(y));
      REP8(OP(a0) OP(a1) OP(a2) OP(a3) OP(a4) OP(a5) OP(a6) OP(a7))
#undef OP
    } else if (MODE == 4) {  // pk_mul
#define OP(A) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(A) : "v"(x));
      REP8(OP(a0) OP(a1) OP(a2) OP(a3) OP(a4) OP(a5) OP(a6) OP(a7))
#undef OP
    } else if (MODE == 5) {  // rcp only
#define OP(A) asm volatile("v_rcp_f32 %0, %0" : "+v"(A));
      REP8(OP(f0) OP(f1) OP(f2) OP(f3) OP(f4) OP(f5) OP(f6) OP(f7))
#undef OP
    } else if (MODE == 6) {  // 1 rcp + 7 pk_fma per group of 8: does the transcendental overlap the FMA pipe?
#define OPF(A) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(A) : "v"(x), "v"(y));
#define OPR(A) asm volatile("v_rcp_f32 %0, %0" : "+v"(A));
      REP8(OPR(f0) OPF(a1) OPF(a2) OPF(a3) OPF(a4) OPF(a5) OPF(a6) OPF(a7))
#undef OPF
#undef OPR
    } else if (MODE == 7) {  // same group with the rcp replaced by a scalar fma
#define OPF(A) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(A) : "v"(x), "v"(y));
#define OPR(A) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(A) : "v"(x[0]), "v"(y[0]));
      REP8(OPR(f0) OPF(a1) OPF(a2) OPF(a3) OPF(a4) OPF(a5) OPF(a6) OPF(a7))
#undef OPF
#undef OPR
    } else if (MODE == 8) {  // v_perm_b32 / v_min_i32 class (plain 32-bit integer VALU)
#define OP(A) asm volatile("v_min_i32 %0, %0, %1" : "+v"(A) : "v"(x[0]));
      REP8(OP(f0) OP(f1) OP(f2) OP(f3) OP(f4) OP(f5) OP(f6) OP(f7))
#undef OP
    } else if (MODE == 9) {  // pk_fma with all three sources distinct VGPR pairs + separate dst
#define OP(A, B) asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(A) : "v"(x), "v"(y), "v"(B));
      REP8(OP(a0, a1) OP(a2, a3) OP(a4, a5) OP(a6, a7) OP(a1, a0) OP(a3, a2) OP(a5, a4) OP(a7, a6))
#undef OP
    }
  }
  long long t1 = clock64();
  v2f s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
extern "C" int run(int mode, float *out, const float *in, int iters, long long *cyc, int blocks, int threads, void *st) {
  hipStream_t s = (hipStream_t)st;
#define L(M) case M: hipLaunchKernelGGL(rate<M>, dim3(blocks), dim3(threads), 0, s, out, in, iters, cyc); break;
  switch (mode) { L(0) L(1) L(2) L(3) L(4) L(5) L(6) L(7) L(8) L(9) }
  return (int)hipGetLastError();
}
