// Does straight-line f64 code that exceeds the 64 KiB instruction cache (shared by two CUs) run slower per instruction than a loop
// that fits?  One wave per SIMD (the five-point kernels' situation), 4096 blocks, every variant executes the SAME number of FMAs:
// a body of kBlocks x 512 v_fma_f64 (8 independent chains, 8 bytes each) x (24 / kBlocks) iterations.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define F(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x##i) : "v"(a), "v"(b));
#define F8 F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7)
#define R8(X) X X X X X X X X
#define F512 R8(R8(F8))
template <int kBlocks>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) void probe(double *out, int iters, double a, double b) {
  extern __shared__ double lds[];
  double x0 = a + threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
    F512
    if constexpr (kBlocks >= 2) { F512 }
    if constexpr (kBlocks >= 3) { F512 }
    if constexpr (kBlocks >= 4) { F512 }
    if constexpr (kBlocks >= 6) { F512 F512 }
    if constexpr (kBlocks >= 8) { F512 F512 }
    if constexpr (kBlocks >= 12) { F512 F512 F512 F512 }
    if constexpr (kBlocks >= 24) { F512 F512 F512 F512 F512 F512 F512 F512 F512 F512 F512 F512 }
  }
  const double s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
  if (s == 123.456) out[blockIdx.x * 64 + threadIdx.x] = s + lds[threadIdx.x];
}
template <int kBlocks>
float run(double *out, int waves_limit_lds) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 24 / kBlocks;
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(probe<kBlocks>, dim3(4096), dim3(64), waves_limit_lds, 0, out, iters, 1.0000001, 1e-9);
  (void)hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(probe<kBlocks>, dim3(4096), dim3(64), waves_limit_lds, 0, out, iters, 1.0000001, 1e-9);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms / 5 * 1e3f;
}
int main() {
  double *out; (void)hipMalloc(&out, 4096 * 64 * 8);
  printf("straight-line body KiB   us per launch   (4096 blocks x 12288 v_fma_f64 per lane, 1 wave/SIMD, 4 blocks per CU)\n");
  printf("%6d %10.1f\n", 4 * 1, run<1>(out, 38912));
  printf("%6d %10.1f\n", 4 * 2, run<2>(out, 38912));
  printf("%6d %10.1f\n", 4 * 4, run<4>(out, 38912));
  printf("%6d %10.1f\n", 4 * 8, run<8>(out, 38912));
  printf("%6d %10.1f\n", 4 * 12, run<12>(out, 38912));
  printf("%6d %10.1f\n", 4 * 24, run<24>(out, 38912));
  return 0;
}
