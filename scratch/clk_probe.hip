// What clock does the chip sustain under (a) f32 MFMA load, (b) packed-f32 VALU load, (c) both?  s_memtime ticks (shader
// cycles, MI355X_MICROARCH.md) per wall nanosecond, measured by one wave while every SIMD runs the same loop.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void load(float *out, const float *in, int iters, unsigned long long *ticks) {
  f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  float a = in[0] + threadIdx.x, b = in[1];
  v2f p0 = {a, b}, p1 = p0 * 1.1f, p2 = p0 * 1.2f, p3 = p0 * 1.3f, x = {in[2], in[3]}, y = {in[4], in[5]};
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0 || MODE == 2) {
      c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c3, 0, 0, 0);
    }
    if (MODE == 3 || MODE == 4) {   // scalar f32 FMAs: 3 = alone, 4 = interleaved with the MFMAs in program order
#define OPS(A) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(A) : "v"(x[0]), "v"(y[0]));
#define GRP OPS(p0[0]) OPS(p1[0]) OPS(p2[0]) OPS(p3[0]) OPS(p0[1]) OPS(p1[1])
      if (MODE == 4) { c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0); }
      GRP
      if (MODE == 4) { c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0); }
      GRP
      if (MODE == 4) { c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c2, 0, 0, 0); }
      GRP
      if (MODE == 4) { c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c3, 0, 0, 0); }
      GRP
#undef GRP
#undef OPS
    }
    if (MODE == 1 || MODE == 2) {
#define OP(A) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(A) : "v"(x), "v"(y));
      OP(p0) OP(p1) OP(p2) OP(p3) OP(p0) OP(p1) OP(p2) OP(p3) OP(p0) OP(p1) OP(p2) OP(p3) OP(p0) OP(p1) OP(p2) OP(p3)
#undef OP
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + p0[0] + p1[0] + p2[1] + p3[1];
  if (blockIdx.x == 0 && threadIdx.x == 0) *ticks = t1 - t0;
}
extern "C" int run(int mode, float *out, const float *in, int iters, unsigned long long *ticks, int blocks, void *st) {
  if (mode == 0) hipLaunchKernelGGL(load<0>, dim3(blocks), dim3(256), 0, (hipStream_t)st, out, in, iters, ticks);
  if (mode == 1) hipLaunchKernelGGL(load<1>, dim3(blocks), dim3(256), 0, (hipStream_t)st, out, in, iters, ticks);
  if (mode == 3) hipLaunchKernelGGL(load<3>, dim3(blocks), dim3(256), 0, (hipStream_t)st, out, in, iters, ticks);
  if (mode == 4) hipLaunchKernelGGL(load<4>, dim3(blocks), dim3(256), 0, (hipStream_t)st, out, in, iters, ticks);
  if (mode == 2) hipLaunchKernelGGL(load<2>, dim3(blocks), dim3(256), 0, (hipStream_t)st, out, in, iters, ticks);
  return (int)hipGetLastError();
}
