// Store-pattern micro-benchmark 2: which property of K4's mask stream costs bandwidth?
#include <hip/hip_runtime.h>
#include <stdint.h>
extern "C" {
// reference: K4's order (block = rows_per_block rows, thread t writes 16 B at row*stride + 16 t)
__global__ void pat_rows(uint8_t *dst, int N, int rows_per_block, size_t stride) {
  const size_t r0 = (size_t)blockIdx.x * rows_per_block;
  const int off = threadIdx.x * 16;
  if (off >= N) return;
  for (int r = 0; r < rows_per_block; ++r) *reinterpret_cast<uint4 *>(dst + (r0 + r) * stride + off) = make_uint4(0, 0, 0, 0);
}
__global__ void pat_block_contig(uint8_t *dst, int N, int rows_per_block) {
  uint8_t *base = dst + (size_t)blockIdx.x * rows_per_block * N;
  const int total = rows_per_block * N;
  for (int o = threadIdx.x * 16; o < total; o += blockDim.x * 16) *reinterpret_cast<uint4 *>(base + o) = make_uint4(0, 0, 0, 0);
}
// proposed: each WAVE owns rows_per_wave whole rows and flushes them in windows of W rows as one contiguous range
__global__ void pat_wave_window(uint8_t *dst, int N, int rows_per_wave, int W) {
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  uint8_t *base = dst + (size_t)wave * rows_per_wave * N;
  const int win = W * N;
  for (int w0 = 0; w0 < rows_per_wave; w0 += W) {
    uint8_t *b = base + (size_t)w0 * N;
    for (int o = lane * 16; o < win; o += 1024) *reinterpret_cast<uint4 *>(b + o) = make_uint4(0, 0, 0, 0);
  }
}
// same but the data really comes out of LDS (ds_read_b128 -> global_store_dwordx4)
__global__ void pat_wave_window_lds(uint8_t *dst, int N, int rows_per_wave, int W) {
  extern __shared__ uint4 lds[];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wave = blockIdx.x * (blockDim.x >> 6) + wv;
  uint4 *mine = lds + wv * (W * N / 16);
  uint8_t *base = dst + (size_t)wave * rows_per_wave * N;
  const int win = W * N;
  for (int w0 = 0; w0 < rows_per_wave; w0 += W) {
    for (int r = 0; r < W; ++r) {   // "evaluation": two ds_write_b128 per row
      if (lane * 16 < N) mine[(r * N + lane * 16) / 16] = make_uint4(r, w0, lane, 0);
      if (1024 + lane * 16 < N) mine[(r * N + 1024 + lane * 16) / 16] = make_uint4(r, w0, lane, 1);
    }
    uint8_t *b = base + (size_t)w0 * N;
    for (int o = lane * 16; o < win; o += 1024) *reinterpret_cast<uint4 *>(b + o) = mine[o / 16];
  }
}
int run(int which, void *dst, int R, int N, int rpb, int threads, size_t stride, int W, void *stream) {
  hipStream_t st = (hipStream_t)stream;
  if (which == 1) hipLaunchKernelGGL(pat_rows, dim3(R / rpb), dim3(threads), 0, st, (uint8_t *)dst, N, rpb, stride);
  if (which == 2) hipLaunchKernelGGL(pat_block_contig, dim3(R / rpb), dim3(threads), 0, st, (uint8_t *)dst, N, rpb);
  if (which == 3) hipLaunchKernelGGL(pat_wave_window, dim3(R / rpb / (threads / 64)), dim3(threads), 0, st, (uint8_t *)dst, N, rpb, W);
  if (which == 4) hipLaunchKernelGGL(pat_wave_window_lds, dim3(R / rpb / (threads / 64)), dim3(threads), (threads / 64) * W * N, st, (uint8_t *)dst, N, rpb, W);
  return (int)hipGetLastError();
}
}
