/*
 * dransac.h -- C ABI of libdransac.so, the MI355X (gfx950) implementation of the
 * differentiable-RANSAC hot path (sampler -> minimal solver -> soft-inlier scoring).
 *
 * The reference (weitong8591/differentiable_ransac) has NO native boundary: its hot path is
 * duck-typed Python objects consumed by RANSAC.__init__ (ransac.py:8-39).  Each entry point
 * below therefore cites the reference *Python* interface it replaces; the Python plugin
 * classes in differentiable_ransac_amd/ bind these symbols through ctypes.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to a contiguous row-major array, unless marked [host];
 *   - P = image pairs, N = points per pair, B = hypotheses (minimal samples) per pair,
 *     k = sample size, S = model slots per sample, M = models per pair (= B*S);
 *   - `stream` is a hipStream_t passed as void*; work is enqueued, never synchronised;
 *   - return value: DR_OK (0) or a negative DR_E* code; dr_last_error() gives the message of
 *     the last failure on the calling thread.  No entry point throws, aborts or allocates
 *     device memory; scratch space is passed in by the caller where needed;
 *   - *_f32 entry points take float I/O, *_f64 double I/O.  Minimal solvers always compute
 *     in f64 internally (f64 FMA runs at the scalar-f32 rate on CDNA4) so that the f32 entry
 *     points meet the 1e-4 model tolerance that the reference's own f32 path does not.
 *   - numerically failed hypotheses never produce NaN: their slot is eye(3) and valid = 0
 *     (the reference drops them, nister.py:154-157,365-366,400-405).
 */
#ifndef DRANSAC_H_
#define DRANSAC_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DR_OK 0
#define DR_EINVAL (-1)  /* bad argument (null pointer, non-positive size, unsupported k) */
#define DR_ELAUNCH (-2) /* HIP launch / runtime failure */
#define DR_ENOTIMPL (-3)

#define DR_ABI_VERSION 1

int dr_version(void);
const char *dr_last_error(void);

/* ------------------------------------------------------------------------------------------
 * K1  Gumbel-softmax top-k sampler     GumbelSoftmaxSampler.sample, samplers/gumbel_sampler.py:25-42
 *
 *   g[p,b,n] = (logits[p,n] + gumbel[p,b,n]) / tau ; y = softmax_n(g) ; idx = top-k_n(g).
 *   gumbel == NULL  -> noise generated in-kernel: Philox4x32-7(key = seed, counter = (n/4, b, p, 0)),
 *                      word n%4 of the call = w; u = fl(float(w) * 2^-32 (1-eps-tiny) + tiny) (the u32 -> f32 conversion rounds
 *                      w to 24 significant bits: torch's tiny + rand * (1-eps-tiny) with rand on the 2^-24 grid),
 *                      gumbel = -ln2 * log2(-log2 u) - ln ln 2 = -log(-log u), rounded to an f32 VALUE before it meets the logit;
 *   gumbel != NULL  -> explicit noise [P,B,N] (parity mode: index sets are bit-exact w.r.t. the reference).
 *   Outputs: idx [P,B,k] int32, ASCENDING point index (= the order `points[samples != 0]` yields,
 *   ransac.py:65); y_sel [P,B,k] = y at idx; lse [P,B] = log-sum-exp of g (so y = exp(g - lse));
 *   optional dense outputs (API-faithful mode) y_soft [P,B,N], ret [P,B,N] = y_hard - y + y
 *   (gumbel_sampler.py:38) and gumbel_out [P,B,N] (the noise that was used); NULL to skip.
 *   logits == NULL  -> all-ones logits (gumbel_sampler.py:27-28).
 *   y_sel == NULL and lse == NULL (both, and then no dense outputs) -> index sets only: what test mode consumes
 *   (`points[samples != 0]`, ransac.py:65); the same idx, without the soft-max statistics of the rows.
 * ------------------------------------------------------------------------------------------ */
/* seed_dev (every sampler entry; round 6: ONE entry per sampler, the `_dseed` twins of rounds 2-5 are inline wrappers in
 * dransac_compat.h): NULL = the Philox key is `seed`; != NULL = the key is read from DEVICE memory (`*seed_dev`) when the kernel
 * starts and `seed` is ignored -- for steps captured in a HIP graph, where a by-value seed would be frozen at capture time.  Same
 * kernels, same random numbers for equal keys; a device key serves the in-kernel noise of given logits only (no explicit noise,
 * no dense outputs).  dr_seed_next_n advances such a key on the device. */
int dr_gumbel_topk_fwd_f32(const float *logits, const float *gumbel, uint64_t seed, const uint64_t *seed_dev, float tau, int P, int B,
                           int N, int k, int32_t *idx, float *y_sel, float *lse, float *y_soft, float *ret,
                           float *gumbel_out, void *stream);
int dr_gumbel_topk_fwd_f64(const double *logits, const double *gumbel, uint64_t seed, const uint64_t *seed_dev, double tau, int P,
                           int B, int N, int k, int32_t *idx, double *y_sel, double *lse, double *y_soft, double *ret,
                           double *gumbel_out, void *stream);

/* Backward of sampler + gather (K1+K2).  a_sel [P,B,k] = dL/d(straight-through value at idx)
 * (= <grad_sample, matches[idx]> + grad_weight, computed by dr_gather_bwd).  grad_logits [P,N] is
 * OVERWRITTEN with (1/tau) * sum_b y*(a - sum_m y_m a_m).  Needs the forward's noise: pass the same
 * gumbel pointer or the same seed. */
int dr_gumbel_topk_bwd_f32(const float *logits, const float *gumbel, uint64_t seed, const uint64_t *seed_dev, float tau, int P, int B,
                           int N, int k, const int32_t *idx, const float *lse, const float *a_sel, float *grad_logits,
                           void *stream);
/* the same in double: `-pr 2 -tr 1` (model_cl.py:164-169; f64 works end to end upstream) */
int dr_gumbel_topk_bwd_f64(const double *logits, const double *gumbel, uint64_t seed, double tau, int P, int B, int N,
                           int k, const int32_t *idx, const double *lse, const double *a_sel, double *grad_logits,
                           void *stream);

/* K1, inference variant: the index SET of the Gumbel top-k sampler by top-down (Plackett-Luce) sampling -- k sequential
 * draws without replacement from softmax(logits), which is the distribution of the top-k of logits + iid Gumbel noise
 * (any tau > 0).  For callers that only consume `samples != 0` (RANSAC test mode, ransac.py:65); y_sel / lse need the
 * whole noise row and come from dr_gumbel_topk_fwd.  logits [P,N] or NULL (uniform); cdf_ws [P,N] f64 workspace (the
 * per-pair cumulative soft-max weights, overwritten); idx [P,B,k] ascending.
 * Philox4x32-7(key = seed, counter = (draw / 2, b, p, 2)). */
int dr_topdown_sample_f32(const float *logits, uint64_t seed, const uint64_t *seed_dev, int P, int B, int N, int k, double *cdf_ws,
                          int32_t *idx, void *stream);
int dr_topdown_sample_f64(const double *logits, uint64_t seed, int P, int B, int N, int k, double *cdf_ws, int32_t *idx,
                          void *stream);

/* K1u  UniformSampler.batch_generate, samplers/uniform_sampler.py:15-19: idx ~ U{0..N-2}, with replacement.
 * Philox4x32-7(key = seed, counter = (j, b, p, 1)). */
int dr_uniform_sample(uint64_t seed, const uint64_t *seed_dev, int P, int B, int k, int N, int32_t *idx, void *stream);

/* The per-call key of the batched drivers, advanced on the device (every replay of a captured step then draws fresh hypotheses):
 * state[0] = base, state[1] = number of calls so far -> seeds_out[i] = base * 0x9E3779B97F4A7C15 + calls + i (mod 2^64) for
 * i < n (<= 65 536), calls += n: CONSECUTIVE integers, one per call (ransac.py of this package: `_next_seed`).  n = 1 is the
 * `dr_seed_next` of rounds 2-5 (dransac_compat.h). */
int dr_seed_next_n(uint64_t *state, uint64_t *seeds_out, int n, void *stream);

/* K1 in index-only mode + K2 in one call (test mode: `points[samples != 0]`, ransac.py:58-65, with in-kernel noise):
 * idx [P,B,k] ascending and samples [P,B,k,4] = matches[p, idx] (c = 4; 16-byte aligned buffers).  One launch when the
 * register-resident sampler kernel serves the shape (N % 4 == 0, N <= 2048, tau == 1), sampler + gather launches otherwise.
 * Every argument from screen_ws on is optional (NULL / 0): */
/* gate_iters / gate_max_iters (a round > 1 of a multi-round test-mode call): pairs with gate_iters[p] >= gate_max_iters[p] are
 * skipped, their rows keep their contents (see dr_ransac_update and the solver entries below) */
/* screen_ws (optional; (N + 32) * P words, 16-byte aligned): rows of <= 2048 points then take the screened register kernel -- only
 * the points whose Philox word can lift them to logsumexp(logits) - ln(11 + k) are evaluated (same index sets, bit for bit). */
/* sub (round 6, super-rounds): > 0 = the B rows are ceil(B / sub) consecutive SUB-BATCHES of `sub` rows, the batches the loop of
 * ransac.py:55-144 draws one call after the other: row b gets the noise of row b % sub of a call keyed (seed | *seed_dev) + b / sub
 * (the drivers' per-call seeds are consecutive integers), so ONE launch samples what ceil(B / sub) calls of that loop sample and
 * dr_ransac_update(sub_models = sub * S) walks them in order.  0 = one batch. */
/* race_ws (round 6; optional, (N + 32) * P floats, 16-byte aligned, not together with screen_ws): rows of <= 2048 points (N % 4 == 0,
 * tau == 1) then rank key_n = exp(lmax - logit_n) * log2 u_n instead of logit_n - ln(-ln u_n) -- the exponential-race form of the
 * same top-k (gumbel_sampler.py:30-36), ONE logarithm per element; the per-pair weights are written into the workspace by a
 * prologue launch.  Same index sets up to the rounding of near-ties (measured: tests/test_gpu_round6.py); pairs whose logits are
 * not all finite or span more than 80 keep the two-logarithm form.  race_ready != 0: the workspace already holds the weights of
 * these logits (dr_ransac_init wrote them, once for all rounds of the call): no prologue launch.  The workspace holds w [P,N], then
 * a flag word and the key density -1 / (ln 2 sum_n 1 / w_n) per pair: this form selects its k winners on wave compare masks, from
 * a threshold searched on that density (same winners, same order as the candidate list of the other forms). */
int dr_gumbel_topk_gather_f32(const float *logits, const float *matches, uint64_t seed, const uint64_t *seed_dev, float tau,
                              int P, int B, int N, int k, int32_t *idx, float *samples, uint32_t *screen_ws,
                              const int32_t *gate_iters, const double *gate_max_iters, int sub, float *race_ws, int race_ready,
                              void *stream);

/* Train mode (round 5): K1 WITH the soft-max statistics + K2 in one call (GumbelSoftmaxSampler.sample, samplers/gumbel_sampler.py:25-42,
 * followed by `matches * ret` + the mask gather of ransac.py:58-65): idx, y_sel [P,B,k], lse [P,B] as dr_gumbel_topk_fwd_f32 and
 * samples [P,B,k,4] = matches[p, idx] * ((1 - y_sel) + y_sel) as dr_gather_fwd_f32 (c = 4; 16-byte aligned buffers).  One launch
 * when the register-resident kernel serves the shape, sampler + gather launches otherwise.
 * race_ws (round 6; optional, (N + 32) * P floats, 16-byte aligned): the one-logarithm form in train mode -- keys, winners AND the
 * soft-max statistics (y_n = (1 / -key_n) / sum_m (1 / -key_m), lse = lmax + ln sum - ln ln 2) from one logarithm and one reciprocal
 * per element; y_sel / lse agree with the two-logarithm form to rounding, the index sets up to the rounding of near-ties
 * (tests/test_gpu_round6.py).  NULL = the two-logarithm form. */
int dr_gumbel_topk_gather_soft_f32(const float *logits, const float *matches, uint64_t seed, const uint64_t *seed_dev, float tau,
                                   int P, int B, int N, int k, int32_t *idx, float *y_sel, float *lse, float *samples, float *race_ws,
                                   void *stream);
/* ... and the backward of the pair in one launch: grad_logits [P,N] from grad_samples [P,B,k,4] and grad_w [P,B,k]
 * (gradient of the y_sel output; NULL = none) -- dr_gather_bwd_f32's a_sel is formed inside dr_gumbel_topk_bwd_f32's row
 * prologue.  The correspondences receive no gradient from this entry. */
int dr_gumbel_topk_gather_bwd_f32(const float *logits, const float *matches, uint64_t seed, const uint64_t *seed_dev, float tau,
                                  int P, int B, int N, int k, const int32_t *idx, const float *lse, const float *grad_samples,
                                  const float *grad_w, float *grad_logits, void *stream);

/* K1, index sets only, in-kernel noise, with an optional screening workspace (round 4; GumbelSoftmaxSampler.sample,
 * samplers/gumbel_sampler.py:25-42, as test mode consumes it: `points[samples != 0]`, ransac.py:65).
 * screen_ws: (N + 32) * P 32-bit words of device memory, 16-byte aligned, or NULL (then = dr_gumbel_topk_fwd_f32 with
 * y_sel = lse = NULL).  With a workspace, rows longer than the register kernel holds (N > 2048, N % 4 == 0, tau == 1, k <= 5)
 * are SCREENED: one pass over the pair's logits writes, per point, the smallest Philox word that can still lift the point to the
 * score T = logsumexp(logits) - ln(20 + k) (rounded down); a step of a wave (256 points) in which no word reaches its
 * threshold is skipped after Philox + four integer compares -- no Gumbel transform, no logits, no list update.  Steps with a
 * candidate are evaluated in full, so every listed score is exact; a row in which fewer than k evaluated scores reach T
 * (probability < 1e-7) repeats itself unscreened.  The index sets are those of dr_gumbel_topk_fwd_f32 for the same seed, bit
 * for bit (tests/test_gpu_round4.py).  seed_dev != NULL: the Philox key is read from device memory (captured graphs). */
int dr_gumbel_topk_index_f32(const float *logits, uint64_t seed, const uint64_t *seed_dev, float tau, int P, int B, int N, int k,
                             int32_t *idx, uint32_t *screen_ws, void *stream);

/* ------------------------------------------------------------------------------------------
 * K2  straight-through gather          RANSAC.__call__, ransac.py:58-65 (+ :73 weighted)
 *   samples[p,b,j,:] = matches[p, idx[p,b,j], :] * st[p,b,j],  st = (1 - y_sel) + y_sel  (f32 rounding
 *   of y_hard - y_soft + y_soft); y_sel == NULL -> st = 1 (uniform sampler, ransac.py:60).
 *   c = 4 (two-view) or 6 (3-D).
 * ------------------------------------------------------------------------------------------ */
int dr_gather_fwd_f32(const float *matches, const int32_t *idx, const float *y_sel, int P, int N, int B, int k,
                      int c, float *samples, void *stream);
int dr_gather_fwd_f64(const double *matches, const int32_t *idx, const double *y_sel, int P, int N, int B, int k,
                      int c, double *samples, void *stream);
/* a_sel[p,b,j] = <grad_samples[p,b,j,:], matches[p,idx,:]> (+ grad_w[p,b,j] if non-NULL);
 * grad_matches [P,N,c] (may be NULL) += grad_samples * st  (atomic; caller zeroes it). */
int dr_gather_bwd_f32(const float *matches, const int32_t *idx, const float *y_sel, const float *grad_samples,
                      const float *grad_w, int P, int N, int B, int k, int c, float *a_sel, float *grad_matches,
                      void *stream);
int dr_gather_bwd_f64(const double *matches, const int32_t *idx, const double *y_sel, const double *grad_samples,
                      const double *grad_w, int P, int N, int B, int k, int c, double *a_sel, double *grad_matches,
                      void *stream);

/* ------------------------------------------------------------------------------------------
 * K3  minimal solvers.  samples [Bt,k,c] (Bt = total hypotheses = P*B, flattened), optional
 *     per-point weights [Bt,k] (NULL = unweighted).  models [Bt,S,9] row-major 3x3 (rigid: [Bt,16]),
 *     valid [Bt,S] uint8.
 *
 *   dr_solve_nister5     EssentialMatrixEstimatorNister.estimate_minimal_model, nister.py:69-408; S = 10;
 *                        n >= 5 points per sample (n > 5 = the non-minimal fallback of nister.py:64-65);
 *                        real roots only, ascending in z; unit Frobenius norm; unused slots = eye(3).
 *   dr_solve_stewenius5  EssentialMatrixEstimator.estimate_minimal_model, stewenius.py:20-80; S = 10;
 *                        unit Frobenius norm (the reference leaves LAPACK's eigenvector scale).
 *   dr_solve_f8          FundamentalMatrixEstimatorNew normalize + estimate_non_minimal_model,
 *                        fundamental_matrix_estimator.py:177-260; S = 1; n >= 8.
 *   dr_solve_f7          7-point with the correct maths (reference degenerate, Q7/Q8); S = 4.
 *   dr_solve_rigid       RigidTransformationSVDBasedSolver.estimate_model, rigid…:11-74; n >= 3, c = 6;
 *                        models [Bt,16] = 4x4, also R [Bt,9], t [Bt,3], scale [Bt] (any may be NULL);
 *                        flag != 0 reproduces the reference default (svd of cov^T cov, R ~ I, Q9).
 * ------------------------------------------------------------------------------------------ */
/* The f32 five-point entries carry every option of the path (round 6: the `_hp`, `_path_`, `_gated_` twins of rounds 3-5 are
 * inline wrappers in dransac_compat.h); everything after `valid` is optional (NULL / 0):
 *   models_f64 (Nister, n = 5): train mode -- the models written BOTH as f32 (what the scoring reads) and as f64 polished to the
 *     f64 tolerance (what dr_solve_nister5_bwd_f32 wants), from one launch: the models of dr_solve_nister5_f64 on the widened
 *     samples without the two conversion passes;
 *   path (n = 5): 0 = automatic, 1 = two lanes per sample from the first instruction (the only kernels until round 4), 2 = the
 *     two-phase kernels: ONE lane per sample for the part the two lanes of a sample otherwise compute twice (null space, the ten
 *     constraints, QR, reduced rows, det B(z) of nister.py:117-348 / the action matrix' characteristic polynomial of
 *     stewenius.py:44-74), handed over in registers to the two-lanes-per-sample root search and final stage (nister.py:355-402,
 *     stewenius.py:74-78); automatic = two-phase when the grid is at least two rounds of lane-pair blocks.  Same solutions either
 *     way (Stewenius: bit for bit; Nister: to the rounding of det B(z));
 *   gate_iters / gate_max_iters + per_pair (n = 5; Bt = pairs x per_pair): device-side termination (ransac.py:135-144:
 *     `max_iters = min(max_iterations, adaptive_iteration_number(...))` decides per pair when the loop of ransac.py:55 ends).  The
 *     per-pair counters live on the device (dr_ransac_init / dr_ransac_update: iters [P] int32, max_iters [P] f64); a block all of
 *     whose samples belong to pairs with iters >= max_iters returns at once, its outputs keep their contents, and dr_ransac_update
 *     leaves such a pair's state alone.  The driver can therefore ISSUE every round of a call without reading anything back -- one
 *     HIP graph per call, whatever the data decide. */
int dr_solve_nister5_f32(const float *samples, const float *weights, int Bt, int n, float *models, double *models_f64, uint8_t *valid,
                         int path, int per_pair, const int32_t *gate_iters, const double *gate_max_iters, void *stream);
int dr_solve_nister5_f64(const double *samples, const double *weights, int Bt, int n, double *models,
                         uint8_t *valid, void *stream);
int dr_solve_stewenius5_f32(const float *samples, int Bt, float *models, uint8_t *valid, int path, int per_pair,
                            const int32_t *gate_iters, const double *gate_max_iters, void *stream);
int dr_solve_stewenius5_f64(const double *samples, int Bt, double *models, uint8_t *valid, void *stream);
int dr_solve_f8_f32(const float *samples, const float *weights, int Bt, int n, float *models, uint8_t *valid,
                    void *stream);
int dr_solve_f8_f64(const double *samples, const double *weights, int Bt, int n, double *models, uint8_t *valid,
                    void *stream);
/* Test hook: the real-root search of the five-point kernels (replaces torch.linalg.eigvals of the companion matrix,
 * nister.py:361-370, and eig, stewenius.py:74) on given polynomials.  coef [n,11] ascending; per polynomial two searches,
 * exactly as the solver kernels run them: roots [n,2,10] -- [.,0,.] the roots with |z| <= 1 ascending, [.,1,.] those with
 * |z| > 1 (found as roots w of the reversed polynomial, ascending in w, returned as 1/w) -- and counts [n,2].
 * method 0 = derivative chain (rounds 1-2), 1 = Sturm-sequence isolation (round 3, what the kernels run). */
int dr_debug_real_roots10(const double *coef, int n, int method, double *roots, int32_t *counts, void *stream);
/* K1u + K2 + K3f8 in one launch (round 4; UniformSampler.batch_generate, uniform_sampler.py:15-19 -> `matches[idx]`, ransac.py:60
 * -> FundamentalMatrixEstimatorNew.estimate_model on the minimal sample): matches [P,N,4]; every one of the P x B samples draws
 * its eight indices in [0, N - 2] itself -- the index sets dr_uniform_sample(seed, P, B, 8, N) draws -- and is solved by the
 * 8-point kernel: idx [P,B,8] (may be NULL), models [P*B,9], valid [P*B].  seed_dev != NULL: the key is read from device memory. */
int dr_solve_f8_uniform_f32(const float *matches, uint64_t seed, const uint64_t *seed_dev, int P, int B, int N, int32_t *idx,
                            float *models, uint8_t *valid, void *stream);
int dr_solve_f7_f32(const float *samples, int Bt, float *models, uint8_t *valid, void *stream);
int dr_solve_f7_f64(const double *samples, int Bt, double *models, uint8_t *valid, void *stream);
int dr_solve_rigid_f32(const float *samples, const float *weights, int Bt, int n, int flag, float *models, float *R,
                       float *t, float *scale, uint8_t *valid, void *stream);
int dr_solve_rigid_f64(const double *samples, const double *weights, int Bt, int n, int flag, double *models,
                       double *R, double *t, double *scale, uint8_t *valid, void *stream);

/* Backward of the minimal solvers by implicit differentiation of the defining constraints at the
 * returned model.  grad_models has the forward's model shape; grad_samples [Bt,k,c]
 * is overwritten.  Invalid slots contribute nothing. */
/* models_f64 (optional, preferred): the same models computed by dr_solve_nister5_f64 on the same samples -- with
 * f32-rounded models the epipolar residual (6e-8) is amplified by the conditioning of the tangent-space system. */
int dr_solve_nister5_bwd_f32(const float *samples, const float *models, const double *models_f64,
                             const uint8_t *valid, const float *grad_models, int Bt, float *grad_samples,
                             void *stream);
/* The same with the gradient in the form K5 leaves it (ransac.py:87-96 picks ONE of a sample's ten models): grad_chosen [Bt,9]
 * = gradient of the model in slot which[s] (which [Bt] int32 from dr_select_closest; < 0 = no model picked), all other slots
 * carry none.  Replaces dr_select_closest_bwd + dr_solve_nister5_bwd on the training path (no dense [Bt,10,9] gradient). */
int dr_solve_nister5_bwd_sel_f32(const float *samples, const float *models, const double *models_f64,
                                 const uint8_t *valid, const float *grad_chosen, const int32_t *which, int Bt,
                                 float *grad_samples, void *stream);
/* Round 5: the same backward kernels with f64 samples / models / gradients in memory (`-pr 2 -tr 1`, model_cl.py:164-169, Q17):
 * the arithmetic was f64 already; until round 4 the f64 training path rounded their inputs and outputs to f32. */
int dr_solve_nister5_bwd_f64(const double *samples, const double *models, const uint8_t *valid, const double *grad_models, int Bt,
                             double *grad_samples, void *stream);
int dr_solve_f8_bwd_f64(const double *samples, const double *weights, const double *models, const double *grad_models, int Bt, int n,
                        double *grad_samples, double *grad_weights, void *stream);
/* Non-minimal samples (n > 5 rows per sample, optional row weights: ransac.py:82-83 `num_samples == 8` feeding the five-point
 * estimator, which runs its minimal code on all rows -- nister.py:64-65, weighted rows :88-93).  The returned models lie in the
 * span of the four smallest eigenvectors of sum_r w_r^2 rho_r rho_r^T; the backward differentiates that invariant subspace and
 * the essential-manifold constraints implicitly.  grad_models [Bt,10,9] dense, grad_samples [Bt,n,4] and grad_weights [Bt,n]
 * (optional) are overwritten. */
int dr_solve_nister5_nm_bwd_f32(const float *samples, const float *weights, const float *models, const double *models_f64,
                                const uint8_t *valid, const float *grad_models, int Bt, int n, float *grad_samples,
                                float *grad_weights, void *stream);
/* the same with samples, weights, models and gradients f64 in memory (`-sam 3 -fmat 0 -tr 1 -pr 2`, model_cl.py:164-169; round 6) */
int dr_solve_nister5_nm_bwd_f64(const double *samples, const double *weights, const double *models, const uint8_t *valid,
                                const double *grad_models, int Bt, int n, double *grad_samples, double *grad_weights, void *stream);
int dr_solve_f8_bwd_f32(const float *samples, const float *weights, const float *models, const float *grad_models,
                        int Bt, int n, float *grad_samples, float *grad_weights, void *stream);
int dr_solve_rigid_bwd_f32(const float *samples, const float *models, const float *grad_models, int Bt, int n,
                           int flag, float *grad_samples, void *stream);

/* ------------------------------------------------------------------------------------------
 * K4  MSAC soft-inlier scoring on the Sampson distance      MSACScore.score, scorings/msac_score.py:12-55
 *   matches [P,N,4], models [P,M,9], thr [P] (the `threshold` argument per pair; the kernel applies
 *   the (3/2 thr)^2 of msac_score.py:21).  scores [P,M]; masks [P,M,N] uint8 (torch.bool layout) or NULL.
 *   Models with a non-finite coefficient get score = NaN and an all-false mask (what the reference's
 *   arithmetic yields for them).  valid [P,M] uint8 (optional, NULL = score everything): slots the solver marked
 *   invalid (eye(3) fillers of non-real roots) get score 0 and an all-false mask row without being evaluated.
 * ------------------------------------------------------------------------------------------ */
/*   gate_iters / gate_max_iters (f32; optional, NULL = none; round 6: the `_gated` twin of round 5 folded in): a round > 1 of a
 *   multi-round test-mode call -- the blocks of a pair with gate_iters[p] >= gate_max_iters[p] return at once, its scores / masks
 *   keep their contents (dr_ransac_update ignores such pairs). */
int dr_msac_score_f32(const float *matches, const float *models, const uint8_t *valid, const float *thr, int P,
                      int M, int N, float *scores, uint8_t *masks, const int32_t *gate_iters, const double *gate_max_iters,
                      void *stream);
int dr_msac_score_f64(const double *matches, const double *models, const uint8_t *valid, const double *thr, int P,
                      int M, int N, double *scores, uint8_t *masks, void *stream);
/* (The explicit-path entry of rounds 2-5, dr_msac_score_path_f32, is gone: path 0 = path 1 = this kernel family -- every (model,
 * point) through the f32 fma chain on the vector units; rows of <= 256 points: a wave per model with 1 / 2 / 4 points per lane,
 * longer rows: a lane owns 8 / 16 points -- and path 2, the matrix-core candidate filter of round 2, measured slower, left the
 * library in round 4: scratch/k4_filter_kernel.patch.) */
/* dL/dmodels [P,M,9] from dL/dscores [P,M] (flows only through points with d2 < thr2). */
int dr_msac_score_bwd_f32(const float *matches, const float *models, const float *thr, const float *grad_scores,
                          int P, int M, int N, float *grad_models, void *stream);

/* K4r squared residual of rigid models   RigidTransformationSVDBasedSolver.squared_residual, rigid…:76-89
 *   pts [P,N,6] = (p,q); models [P,M,16] (4x4: q_hat = R p + t with the reference's row-vector descriptor
 *   D = model[:3,:]^T, ransac.py:380); res_sum [P,M] = sum_n d2; masks [P,M,N] = d2 < threshold or NULL.
 *   The reference's scalar mean is sum(res_sum)/(M*N), left to the caller. */
/*   accumulate (f32): != 0 = the sums are ADDED to res_sum, which must hold zeros (dr_solve_rigid_gather_f32 with zero_sums =
 *   res_sum, M = B, leaves it so): no memset launch (round 6: the `_acc` twin of round 4 folded in) */
int dr_rigid_residual_f32(const float *pts, const float *models, float threshold, int P, int M, int N,
                          float *res_sum, uint8_t *masks, int accumulate, void *stream);
int dr_rigid_residual_f64(const double *pts, const double *models, double threshold, int P, int M, int N,
                          double *res_sum, uint8_t *masks, void *stream);
/* Round 4, test mode of the 3-D driver (RANSAC3D.__call__, ransac.py:355-367,380): K2 + K3r in one launch and K4r without its
 * memset launch.
 *   dr_solve_rigid_gather_f32: samples are read straight through the index sets -- matches [P,N,6], idx [P,B,k] ->
 *     models [P*B,16], R / t / scale (may be NULL), valid [P*B]; zero_sums [P*B] (may be NULL) is cleared on the way;
 *   dr_rigid_residual_f32(accumulate = 1) then adds the sums to it. */
int dr_solve_rigid_gather_f32(const float *matches, const int32_t *idx, int P, int B, int N, int k, int flag, float *models,
                              float *R, float *t, float *scale, uint8_t *valid, float *zero_sums, void *stream);
int dr_rigid_residual_bwd_f32(const float *pts, const float *models, const float *grad_res, int P, int M, int N,
                              float *grad_models, void *stream);

/* K6 of the 3-D path (RANSAC3D.__call__ test branch, ransac.py:383-406 -- dead code upstream; selection rule =
 * the valid model with the smallest residual sum, as BatchedRANSAC3D documents): per pair the arg-min of res [P,M] over the
 * valid slots (valid [P,M] or NULL; NaN sums never win; ties -> lowest index), compared STRICTLY with best_res_in [P].
 * Where it is better: best_res_out / best_model_out [P,16] take the winner and best_mask [P,N] (or NULL) is rewritten with
 * `d2 < threshold` of the winner; elsewhere the state is copied through and the mask left alone.  The small state is
 * ping-ponged (in != out) because several blocks serve one pair.  best_idx [P] (or NULL): the round's winner, -1 if kept.
 * First round of a call: best_res_in = best_model_in = NULL stands for "no state yet" (residual +inf, model = identity, empty
 * mask -- the mask is then written in full even where no model is selected). */
int dr_ransac3d_update_f32(const float *pts, const float *models, const uint8_t *valid, const float *res, float threshold,
                           int P, int M, int N, const float *best_res_in, const float *best_model_in, float *best_res_out,
                           float *best_model_out, uint8_t *best_mask, int32_t *best_idx, void *stream);
int dr_ransac3d_update_f64(const double *pts, const double *models, const uint8_t *valid, const double *res, double threshold,
                           int P, int M, int N, const double *best_res_in, const double *best_model_in, double *best_res_out,
                           double *best_model_out, uint8_t *best_mask, int32_t *best_idx, void *stream);

/* ------------------------------------------------------------------------------------------
 * K5  train-mode best-of-S selection      RANSAC.__call__, ransac.py:87-96
 *   chosen[p,b] = models[p,b,argmin_s ||models[p,b,s] - gt[p]||_F]; invalid slots (valid == 0) are
 *   skipped; which [P*B] int32 (-1 when no slot is valid; chosen = eye(3) then).
 * ------------------------------------------------------------------------------------------ */
/* keep [P*B] uint8 (optional, NULL = not wanted) = (which >= 0): the `nan_filter` of ransac.py:104-106 as a flag the loss consumes
 * (round 6: the `_keep` twin of round 3 folded in). */
int dr_select_closest_f32(const float *models, const uint8_t *valid, const float *gt, int P, int B, int S,
                          float *chosen, int32_t *which, uint8_t *keep, void *stream);
int dr_select_closest_f64(const double *models, const uint8_t *valid, const double *gt, int P, int B, int S,
                          double *chosen, int32_t *which, uint8_t *keep, void *stream);
/* backward: grad_models [P,B,S,9] = grad_chosen [P,B,9] at slot which[p,b], 0 elsewhere (all of it is written). */
int dr_select_closest_bwd_f32(const float *grad_chosen, const int32_t *which, int P, int B, int S, float *grad_models,
                              void *stream);
int dr_select_closest_bwd_f64(const double *grad_chosen, const int32_t *which, int P, int B, int S, double *grad_models,
                              void *stream);

/* ------------------------------------------------------------------------------------------
 * K6  test-mode selection      RANSAC.__call__, ransac.py:111-120
 *   Per pair: best_idx = first arg-max of scores over valid models (NaN scores never win),
 *   best_score, best_mask [P,N] uint8 recomputed for the winning model, inlier count.
 *   valid may be NULL (all valid).
 * ------------------------------------------------------------------------------------------ */
int dr_select_best_f32(const float *matches, const float *models, const uint8_t *valid, const float *scores,
                       const float *thr, int P, int M, int N, int32_t *best_idx, float *best_score,
                       float *best_model, uint8_t *best_mask, int32_t *inliers, void *stream);
int dr_select_best_f64(const double *matches, const double *models, const uint8_t *valid, const double *scores,
                       const double *thr, int P, int M, int N, int32_t *best_idx, double *best_score,
                       double *best_model, uint8_t *best_mask, int32_t *inliers, void *stream);

/* K6 set-up: threshold normalisation of ransac.py:49-53 and the initial per-pair state, one launch.
 *   K1, K2: calibration matrices, [3,3] shared by all pairs (k_stride 0) or [P,3,3] (k_stride 9), or both NULL
 *   (fundamental-matrix mode / threshold already normalised: thr[p] = threshold).  Otherwise
 *   thr[p] = threshold / ((K1[0,0] + K1[1,1] + K1[0,0] + K2[1,1]) / 4)  -- sic, ransac.py:52.
 *   State: best_score = 0, best_model = eye(3), best_mask = 0, best_inliers = 0, iters = 0, max_iters = max_iterations.
 *   seed_state / seeds_out / n_seeds (optional, NULL / 0 = none; round 6): the call's sampler keys from the same launch --
 *   dr_seed_next_n(seed_state, seeds_out, n_seeds), one node fewer in a replayed call.
 *   race_logits [P,N] f32 / race_ws ((N + 32) * P floats; optional, both or neither): the per-pair weights of the one-logarithm
 *   sampler (dr_gumbel_topk_gather_f32's race_ws) from the same launch, once per call: pass the workspace to the sampler with
 *   race_ready = 1. */
int dr_ransac_init_f32(const float *K1, const float *K2, int k_stride, double threshold, int P, int N,
                       int max_iterations, float *thr, float *best_score, float *best_model, uint8_t *best_mask,
                       int32_t *best_inliers, int32_t *iters, double *max_iters, uint64_t *seed_state, uint64_t *seeds_out,
                       int n_seeds, const float *race_logits, float *race_ws, void *stream);
int dr_ransac_init_f64(const double *K1, const double *K2, int k_stride, double threshold, int P, int N,
                       int max_iterations, double *thr, double *best_score, double *best_model, uint8_t *best_mask,
                       int32_t *best_inliers, int32_t *iters, double *max_iters, uint64_t *seed_state, uint64_t *seeds_out,
                       int n_seeds, const float *race_logits, float *race_ws, void *stream);

/* K6 (batched, state on the device)  RANSAC.__call__ ransac.py:109-144 + adaptive_iteration_number :202-215.
 *   For every pair p with iters[p] < max_iters[p] (the others have terminated and are left untouched):
 *     b = first arg-max of scores[p] over valid, non-NaN models;
 *     if scores[p,b] > best_score[p] or iters[p] == 0:  best_score / best_model [P,9] / best_mask [P,N] /
 *         best_inliers <- model b (mask recomputed), max_iters[p] = min(max_iterations,
 *         log10(1-confidence) / log10(1 - (inliers/N)^k + eps))   (computed in f64);
 *     iters[p] += B.
 *   The caller initialises best_score = 0, iters = 0, max_iters = max_iterations (dr_ransac_init does).
 *   sub_models (round 6): 0 (or >= M) = the M models are one batch of B hypotheses.  0 < sub_models < M: they are ceil(M / sub_models)
 *   consecutive sub-batches of B hypotheses each (at most 512), and the steps above are applied to one sub-batch after the other, IN
 *   ORDER, stopping as the loop of ransac.py:55 does when iters[p] >= max_iters[p]: the state after the launch is the state that loop
 *   reaches on the same hypotheses batch by batch -- the reference's `-rbs 64` costs the launches of `-rbs 1024`. */
int dr_ransac_update_f32(const float *matches, const float *models, const uint8_t *valid, const float *scores,
                         const float *thr, int P, int M, int N, int B, int k, double confidence, double eps,
                         int max_iterations, float *best_score, float *best_model, uint8_t *best_mask,
                         int32_t *best_inliers, int32_t *iters, double *max_iters, int sub_models, void *stream);
int dr_ransac_update_f64(const double *matches, const double *models, const uint8_t *valid, const double *scores,
                         const double *thr, int P, int M, int N, int B, int k, double confidence, double eps,
                         int max_iterations, double *best_score, double *best_model, uint8_t *best_mask,
                         int32_t *best_inliers, int32_t *iters, double *max_iters, int sub_models, void *stream);

/* ------------------------------------------------------------------------------------------
 * K7  final refit, RANSAC.__call__ ransac.py:148-195, batched over pairs (one cooperative block per pair, the ragged
 *     inlier sets stay on the device).  mask [P,N] uint8 selects the points (NULL = all).
 *   dr_refit_essential    five-point solver on all selected points as ONE sample (nister.py:64-65, the path taken
 *                         when pymagsac is absent; ransac.py:157-165 passes ALL points => mask = NULL), computed in
 *                         f64; models [P,10,9], valid [P,10].
 *   dr_refit_fundamental  Hartley-normalised LSQ 8-point on the selected points (ransac.py:150-155 passes the inliers
 *                         of the best mask); models [P,9], valid [P] (0 when fewer than 8 points are selected).
 *                         weights [P,N] (NULL = unweighted; round 6: the `_w` twin of round 3 folded in): RANSAC.__call__ with
 *                         `weighted=1`, ransac.py:151-153 -- `estimate_model(inlier_points, soft_weights[0, inlier_indices[0]])`,
 *                         the weights multiply the epipolar rows (fundamental_matrix_estimator.py:243-244); the Hartley
 *                         normalisation of the selected points stays unweighted (:177-228).
 * ------------------------------------------------------------------------------------------ */
int dr_refit_essential_f32(const float *matches, const uint8_t *mask, int P, int N, float *models, uint8_t *valid,
                           void *stream);
int dr_refit_essential_f64(const double *matches, const uint8_t *mask, int P, int N, double *models, uint8_t *valid,
                           void *stream);
int dr_refit_fundamental_f32(const float *matches, const uint8_t *mask, const float *weights, int P, int N, float *models,
                             uint8_t *valid, void *stream);
int dr_refit_fundamental_f64(const double *matches, const uint8_t *mask, const double *weights, int P, int N, double *models,
                             uint8_t *valid, void *stream);

/* K7 acceptance (ransac.py:173-185): MSAC scores of the S refit candidates of every pair (cand [P,S,9], cand_valid [P,S]
 * or NULL); where the best candidate scores strictly higher than best_score[p], best_score[p] and best_model[p] ([P,9])
 * are replaced in place (the best mask is not touched, as in the reference). */
int dr_refit_accept_f32(const float *matches, const float *cand, const uint8_t *cand_valid, const float *thr, int P, int S,
                        int N, float *best_score, float *best_model, void *stream);
int dr_refit_accept_f64(const double *matches, const double *cand, const uint8_t *cand_valid, const double *thr, int P,
                        int S, int N, double *best_score, double *best_model, void *stream);

/* ------------------------------------------------------------------------------------------
 * the training loss right after the path -- MatchLoss (loss.py:107-153) on batch_episym
 * (cv_utils.py:680-695).  sums [P,M] = sum over the points with mask[p,n] != 0 (NULL = all points) of
 * min(ys, 1), ys = (x2^T M x1)^2 (1/((Mx1)_0^2+(Mx1)_1^2+1e-15) + 1/((M^T x2)_0^2+(M^T x2)_1^2+1e-15)).
 * Slots with valid == 0 get 0 (NULL = all valid).  The mean over (models x masked points) and over pairs is the
 * caller's.  Backward: grad_models [P,M,9] from grad_sums [P,M] (the clamp passes no gradient at ys >= 1;
 * invalid slots get 0: every entry of grad_models is written).
 * ------------------------------------------------------------------------------------------ */
int dr_episym_fwd_f32(const float *matches, const uint8_t *mask, const float *models, const uint8_t *valid, int P, int M,
                      int N, float *sums, void *stream);
int dr_episym_bwd_f32(const float *matches, const uint8_t *mask, const float *models, const uint8_t *valid,
                      const float *grad_sums, int P, int M, int N, float *grad_models, void *stream);
/* The rest of MatchLoss.forward (loss.py:146-153: mean over the GT-inlier points, mean over the models, per pair):
 *   per_pair[p] = sum_m sums[p,m] / max(n_in[p] * n_models[p], 1),  coef[p] = 1 / max(n_in[p] * n_models[p], 1)
 * with n_in = number of points with mask != 0 (NULL = N) and n_models = number of slots with keep != 0 (NULL = M); the
 * mean over pairs stays with the caller.  dr_episym_bwd_pair is dr_episym_bwd with ONE gradient per pair
 * (grad_pair[p] = d loss / d sums[p, m] for every m, i.e. upstream gradient x coef[p]). */
int dr_match_loss_pair_f32(const float *sums, const uint8_t *mask, const uint8_t *keep, int P, int M, int N,
                           float *per_pair, float *coef, void *stream);
int dr_episym_bwd_pair_f32(const float *matches, const uint8_t *mask, const float *models, const uint8_t *valid,
                           const float *grad_pair, int P, int M, int N, float *grad_models, void *stream);
/* MatchLoss down to the scalar (loss.py:146-153 including `.mean()` over the pairs of the batch), one launch:
 *   per_pair / coef as above and mean[0] = sum_p per_pair[p] / P.  dr_episym_bwd_mean is the matching backward:
 *   d loss / d sums[p,m] = grad_mean[0] * coef[p] / P (grad_mean = the upstream gradient of the scalar, one float in device
 *   memory), so no per-pair gradient tensor is formed on the host side. */
/* Round 5: MatchLoss value + gradient in ONE pass over the (model x point) grid (the loss is a scalar mean: its gradient w.r.t. a
 * model is one number per pair times a quantity the forward can write while it holds the residuals in registers).
 * dr_match_loss_fused_f32 = dr_episym_fwd + dr_match_loss_mean + the unscaled gradient grad_unscaled [P,M,9] = d sums[p,m] / d model.
 * dr_match_loss_scale_f32: grad_models = grad_unscaled x coef[p] x grad_mean[0] / P, the whole backward of the loss. */
int dr_match_loss_fused_f32(const float *matches, const uint8_t *mask, const float *models, const uint8_t *valid, int P, int M, int N,
                            float *sums, float *grad_unscaled, float *per_pair, float *coef, float *mean, void *stream);
int dr_match_loss_scale_f32(const float *grad_unscaled, const float *coef, const float *grad_mean, int P, int M, float *grad_models,
                            void *stream);
int dr_match_loss_mean_f32(const float *sums, const uint8_t *mask, const uint8_t *keep, int P, int M, int N,
                           float *per_pair, float *coef, float *mean, void *stream);
int dr_episym_bwd_mean_f32(const float *matches, const uint8_t *mask, const float *models, const uint8_t *valid,
                           const float *coef, const float *grad_mean, int P, int M, int N, float *grad_models, void *stream);

/* ------------------------------------------------------------------------------------------
 * pose error of essential matrices -- the body of PoseLoss.forward_average (loss.py:11-68) =
 * eval_essential_matrix(svd=False) (cv_utils.py:503-525) for every model of every pair:
 *   Horn decomposition (new_decompose_E, cv_utils.py:118-161) -> candidates (R1,t) (R2,t) (R1,-t) (R2,-t);
 *   votes[c] = number of points triangulated in front of both cameras and closer than distance_threshold
 *   (recoverPose / cheirality_check, cv_utils.py:48-80,177-189; the reference passes 50); which = first arg-max;
 *   err_R, err_t in degrees (evaluate_R_t_tensor, cv_utils.py:361-380).
 *   matches [P,N,4] (normalised coordinates), models [P,M,9], gt_R [P,9], gt_t [P,3];
 *   err_R, err_t [P,M]; which [P,M] int32; votes [P,M,4] int32 or NULL.
 * Backward: grad_models [P,M,9] from grad_err_R, grad_err_t [P,M] at the recorded candidate `which`
 * (the skew matrix of Horn's formula is a constant, as in the reference; non-finite derivatives -> 0).
 * ------------------------------------------------------------------------------------------ */
int dr_pose_error_fwd_f32(const float *matches, const float *models, const float *gt_R, const float *gt_t, int P, int M,
                          int N, double distance_threshold, float *err_R, float *err_t, int32_t *which, int32_t *votes,
                          void *stream);
int dr_pose_error_fwd_f64(const double *matches, const double *models, const double *gt_R, const double *gt_t, int P,
                          int M, int N, double distance_threshold, double *err_R, double *err_t, int32_t *which,
                          int32_t *votes, void *stream);
/* The same with the SVD decomposition of the reference's `svd=True` branch (decompose_E, cv_utils.py:83-116;
 * eval_essential_matrix's default, cv_utils.py:503): R1,2 = U W^(+-T) V^T, t = +-u3.  The four candidate poses are the
 * same set for every SVD sign convention; their order -- `which`, and the winner of an exact tie of votes -- is not.
 * Forward only: the reference's gradient goes through torch.linalg.svd of a matrix with sigma_1 = sigma_2, where it is
 * undefined. */
int dr_pose_error_svd_fwd_f32(const float *matches, const float *models, const float *gt_R, const float *gt_t, int P, int M,
                              int N, double distance_threshold, float *err_R, float *err_t, int32_t *which, int32_t *votes,
                              void *stream);
int dr_pose_error_svd_fwd_f64(const double *matches, const double *models, const double *gt_R, const double *gt_t, int P,
                              int M, int N, double distance_threshold, double *err_R, double *err_t, int32_t *which,
                              int32_t *votes, void *stream);
int dr_pose_error_bwd_f32(const float *models, const float *gt_R, const float *gt_t, const int32_t *which,
                          const float *grad_err_R, const float *grad_err_t, int P, int M, float *grad_models,
                          void *stream);
int dr_pose_error_bwd_f64(const double *models, const double *gt_R, const double *gt_t, const int32_t *which,
                          const double *grad_err_R, const double *grad_err_t, int P, int M, double *grad_models,
                          void *stream);

/* The inlier mask of cv2.recoverPose (loss.py:99,134: the ground-truth inlier mask of ClassificationLoss / MatchLoss):
 * per (pair, model) the points that pass the cheirality test of the winning one of the four candidate poses.
 * models [P,M,9] (normally the ground-truth E, M = 1); which [P,M] int32 or NULL; mask [P,M,N] uint8. */
int dr_recover_pose_mask_f32(const float *matches, const float *models, int P, int M, int N, double distance_threshold,
                             int32_t *which, uint8_t *mask, void *stream);
int dr_recover_pose_mask_f64(const double *matches, const double *models, int P, int M, int N, double distance_threshold,
                             int32_t *which, uint8_t *mask, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* DRANSAC_H_ */
