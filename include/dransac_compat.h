/* dransac_compat.h -- the entry-point names of rounds 2-5 that round 6 folded into their base entries (flag / pointer arguments
 * instead of `_dseed`, `_gated`, `_path`, `_hp`, `_keep`, `_acc`, `_w` twins: the kernels were the same).  Each is a static inline
 * wrapper of the entry that replaced it; libdransac.so no longer exports these symbols.  Kept for ONE round: bind the names of
 * dransac.h. */
#ifndef DRANSAC_COMPAT_H
#define DRANSAC_COMPAT_H

#include "dransac.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- the Philox key in device memory: `seed_dev` is an argument of every sampler entry now ---- */
static inline int dr_gumbel_topk_fwd_f32_dseed(const float *logits, const uint64_t *seed_dev, float tau, int P, int B, int N, int k,
                                               int32_t *idx, float *y_sel, float *lse, void *stream) {
  return dr_gumbel_topk_fwd_f32(logits, 0, 0, seed_dev, tau, P, B, N, k, idx, y_sel, lse, 0, 0, 0, stream);
}
static inline int dr_gumbel_topk_fwd_f64_dseed(const double *logits, const uint64_t *seed_dev, double tau, int P, int B, int N, int k,
                                               int32_t *idx, double *y_sel, double *lse, void *stream) {
  return dr_gumbel_topk_fwd_f64(logits, 0, 0, seed_dev, tau, P, B, N, k, idx, y_sel, lse, 0, 0, 0, stream);
}
static inline int dr_gumbel_topk_bwd_f32_dseed(const float *logits, const uint64_t *seed_dev, float tau, int P, int B, int N, int k,
                                               const int32_t *idx, const float *lse, const float *a_sel, float *grad_logits,
                                               void *stream) {
  return dr_gumbel_topk_bwd_f32(logits, 0, 0, seed_dev, tau, P, B, N, k, idx, lse, a_sel, grad_logits, stream);
}
static inline int dr_topdown_sample_f32_dseed(const float *logits, const uint64_t *seed_dev, int P, int B, int N, int k,
                                              double *cdf_ws, int32_t *idx, void *stream) {
  return dr_topdown_sample_f32(logits, 0, seed_dev, P, B, N, k, cdf_ws, idx, stream);
}
static inline int dr_uniform_sample_dseed(const uint64_t *seed_dev, int P, int B, int k, int N, int32_t *idx, void *stream) {
  return dr_uniform_sample(0, seed_dev, P, B, k, N, idx, stream);
}
static inline int dr_seed_next(uint64_t *state, uint64_t *seed_out, void *stream) { return dr_seed_next_n(state, seed_out, 1, stream); }

/* ---- sampler + gather of a gated round: dr_gumbel_topk_gather_f32 carries the options ---- */
static inline int dr_gumbel_topk_gather_gated_f32(const float *logits, const float *matches, uint64_t seed, const uint64_t *seed_dev,
                                                  float tau, int P, int B, int N, int k, int32_t *idx, float *samples,
                                                  uint32_t *screen_ws, const int32_t *gate_iters, const double *gate_max_iters,
                                                  void *stream) {
  return dr_gumbel_topk_gather_f32(logits, matches, seed, seed_dev, tau, P, B, N, k, idx, samples, screen_ws, gate_iters,
                                   gate_max_iters, 0, 0, 0, stream);
}

/* ---- five-point solvers ---- */
static inline int dr_solve_nister5_f32_hp(const float *samples, const float *weights, int Bt, float *models, double *models_f64,
                                          uint8_t *valid, void *stream) {
  return dr_solve_nister5_f32(samples, weights, Bt, 5, models, models_f64, valid, 0, 0, 0, 0, stream);
}
static inline int dr_solve_nister5_path_f32(const float *samples, const float *weights, int Bt, float *models, double *models_f64,
                                            uint8_t *valid, int path, void *stream) {
  return dr_solve_nister5_f32(samples, weights, Bt, 5, models, models_f64, valid, path, 0, 0, 0, stream);
}
static inline int dr_solve_nister5_gated_f32(const float *samples, const float *weights, int Bt, float *models, uint8_t *valid,
                                             int per_pair, const int32_t *gate_iters, const double *gate_max_iters, void *stream) {
  return dr_solve_nister5_f32(samples, weights, Bt, 5, models, 0, valid, 0, per_pair, gate_iters, gate_max_iters, stream);
}
static inline int dr_solve_stewenius5_path_f32(const float *samples, int Bt, float *models, uint8_t *valid, int path, void *stream) {
  return dr_solve_stewenius5_f32(samples, Bt, models, valid, path, 0, 0, 0, stream);
}
static inline int dr_solve_stewenius5_gated_f32(const float *samples, int Bt, float *models, uint8_t *valid, int per_pair,
                                                const int32_t *gate_iters, const double *gate_max_iters, void *stream) {
  return dr_solve_stewenius5_f32(samples, Bt, models, valid, 0, per_pair, gate_iters, gate_max_iters, stream);
}

/* ---- scoring, selection, residuals, refit ---- */
static inline int dr_msac_score_gated_f32(const float *matches, const float *models, const uint8_t *valid, const float *thr, int P,
                                          int M, int N, float *scores, uint8_t *masks, const int32_t *gate_iters,
                                          const double *gate_max_iters, void *stream) {
  return dr_msac_score_f32(matches, models, valid, thr, P, M, N, scores, masks, gate_iters, gate_max_iters, stream);
}
static inline int dr_msac_score_path_f32(const float *matches, const float *models, const uint8_t *valid, const float *thr, int P,
                                         int M, int N, float *scores, uint8_t *masks, int path, void *stream) {
  if (path != 0 && path != 1) return -1;   /* path 2 (the matrix-core filter of round 2) left the library in round 4 */
  return dr_msac_score_f32(matches, models, valid, thr, P, M, N, scores, masks, 0, 0, stream);
}
static inline int dr_select_closest_keep_f32(const float *models, const uint8_t *valid, const float *gt, int P, int B, int S,
                                             float *chosen, int32_t *which, uint8_t *keep, void *stream) {
  return dr_select_closest_f32(models, valid, gt, P, B, S, chosen, which, keep, stream);
}
static inline int dr_select_closest_keep_f64(const double *models, const uint8_t *valid, const double *gt, int P, int B, int S,
                                             double *chosen, int32_t *which, uint8_t *keep, void *stream) {
  return dr_select_closest_f64(models, valid, gt, P, B, S, chosen, which, keep, stream);
}
static inline int dr_rigid_residual_acc_f32(const float *pts, const float *models, float threshold, int P, int M, int N,
                                            float *res_sum, uint8_t *masks, void *stream) {
  return dr_rigid_residual_f32(pts, models, threshold, P, M, N, res_sum, masks, 1, stream);
}
static inline int dr_refit_fundamental_w_f32(const float *matches, const uint8_t *mask, const float *weights, int P, int N,
                                             float *models, uint8_t *valid, void *stream) {
  return dr_refit_fundamental_f32(matches, mask, weights, P, N, models, valid, stream);
}
static inline int dr_refit_fundamental_w_f64(const double *matches, const uint8_t *mask, const double *weights, int P, int N,
                                             double *models, uint8_t *valid, void *stream) {
  return dr_refit_fundamental_f64(matches, mask, weights, P, N, models, valid, stream);
}

#ifdef __cplusplus
}
#endif
#endif /* DRANSAC_COMPAT_H */
