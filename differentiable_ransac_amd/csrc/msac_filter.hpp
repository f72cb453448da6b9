// K4 benchmark-shape path (msac_filter.hip): declarations shared with msac_score.hip, which chooses the kernel.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dr {

// hard constraint of the kernel: N % 16 == 0, 16 <= N <= 2048 (f32 only)
bool msac_filter_supported(int N);
// enough (pair x 16-slot chunk) work for a persistent chip-filling grid to amortise its prologue
bool msac_filter_profitable(int P, int M, int N);

int msac_filter_launch(const float *matches, const float *models, const uint8_t *valid, const float *thr, int P, int M,
                       int N, float *scores, uint8_t *masks, hipStream_t st);

}  // namespace dr
