// Store-pattern micro-benchmark for the K4 mask stream: zero-fills a [R, N] byte matrix (R = P*M rows) in different orders.
#include <hip/hip_runtime.h>
#include <stdint.h>
extern "C" {
__global__ void pat_flat(uint4 *dst, size_t n16) {   // memset-like
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n16; i += stride) dst[i] = make_uint4(0, 0, 0, 0);
}
// block = 64 rows x (blockDim.x * 16) bytes; thread t writes 16 B at row*N + t*16, row after row (K4's order)
__global__ void pat_rows(uint8_t *dst, int N, int rows_per_block, size_t stride) {
  const size_t r0 = (size_t)blockIdx.x * rows_per_block;
  const int off = threadIdx.x * 16;
  if (off >= N) return;
  for (int r = 0; r < rows_per_block; ++r) *reinterpret_cast<uint4 *>(dst + (r0 + r) * stride + off) = make_uint4(0, 0, 0, 0);
}
// same rows, but the block treats its rows_per_block*N bytes as ONE contiguous range (only valid when stride == N)
__global__ void pat_block_contig(uint8_t *dst, int N, int rows_per_block) {
  uint8_t *base = dst + (size_t)blockIdx.x * rows_per_block * N;
  const int total = rows_per_block * N;
  for (int o = threadIdx.x * 16; o < total; o += blockDim.x * 16) *reinterpret_cast<uint4 *>(base + o) = make_uint4(0, 0, 0, 0);
}
int run(int which, void *dst, int R, int N, int rows_per_block, int threads, size_t stride, void *stream) {
  hipStream_t st = (hipStream_t)stream;
  if (which == 0) hipLaunchKernelGGL(pat_flat, dim3(256 * 8), dim3(256), 0, st, (uint4 *)dst, (size_t)R * N / 16);
  if (which == 1) hipLaunchKernelGGL(pat_rows, dim3(R / rows_per_block), dim3(threads), 0, st, (uint8_t *)dst, N, rows_per_block, stride);
  if (which == 2) hipLaunchKernelGGL(pat_block_contig, dim3(R / rows_per_block), dim3(threads), 0, st, (uint8_t *)dst, N, rows_per_block);
  return (int)hipGetLastError();
}
}
