/* tramp_hip.h - C ABI of the MI355X (gfx950) message-passing hot path of Tree-AMP.
 *
 * Drop-in boundary.  The reference, sphinxteam/tramp, is pure
 * Python: its "plugin interface" for this path is the factor-module protocol
 * (compute_forward_message / compute_backward_message / compute_*_posterior /
 * compute_*_error).  Each entry point below replaces the NumPy body of one such
 * method; the reference file:line it stands for is cited on every declaration
 * (paths relative to the reference repository root).  The Python mirror of the
 * protocol lives in tramp_amd/ and binds these symbols with ctypes
 * (INTEGRATION.md shows the stub a tramp maintainer would add).
 *
 * Conventions
 *  - extern "C", plain pointers and sizes, no C++ / torch types.
 *  - every function returns a tramp_status; the message is retrievable with
 *    tramp_last_error(ctx).  Nothing aborts.
 *  - dtype selects the element type of the N-vectors ("b", "r", "y"): f32 or
 *    f64.  Scalar precisions "a" and variances "v" are ALWAYS double.
 *  - ensemble axis: every vector argument is a [batch, n] row-major array of
 *    independent problem instances; every scalar argument is a double[batch]
 *    array IN DEVICE MEMORY (instances never interact).  batch = 1 is the
 *    reference's single-instance call.
 *  - all pointers named *_d / documented "device" are device pointers obtained
 *    from tramp_buf_alloc (or any hipMalloc'd memory); kernels never allocate;
 *    inputs are const; launches are asynchronous on the context's stream, in
 *    program order.  tramp_sync / tramp_d2h synchronise.
 *  - message update (tramp/base.py:250-255, fused into every kernel):
 *        a_new = clip(1/max(v,1e-20) - a, 1e-11, 1e11);  b_new = r (a + a_new) - b
 *    optional constant damping (tramp/algos/message_passing.py:119-127):
 *        out = d * old + (1 - d) * new   on both a and b,
 *    selected by passing damping != 0 together with a_old / b_old (which may
 *    alias the outputs for an in-place update).
 *  - a non-finite posterior mean poisons the reduced variance (NaN), so the
 *    reference's NaN guard (message_passing.py:187-209) only needs a_new.
 */
#ifndef TRAMP_HIP_H
#define TRAMP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct tramp_ctx tramp_ctx;
typedef struct tramp_linear tramp_linear;
typedef struct tramp_event tramp_event;

typedef enum {
    TRAMP_OK = 0,
    TRAMP_EINVAL = 1,   /* bad argument (null pointer, negative size, unknown enum) */
    TRAMP_EHIP = 2,     /* a HIP runtime call failed; see tramp_last_error */
    TRAMP_ENOMEM = 3,
    TRAMP_ENOTSUP = 4,  /* valid request this build does not implement */
    TRAMP_ENODEV = 5    /* no usable gfx950 device */
} tramp_status;

typedef enum { TRAMP_F32 = 0, TRAMP_F64 = 1 } tramp_dtype;
typedef enum { TRAMP_SGN = 0, TRAMP_ABS = 1, TRAMP_RELU = 2 } tramp_pwl_kind;
typedef enum { TRAMP_FWD = 0, TRAMP_BWD = 1 } tramp_dir;

/* ---- library / context ------------------------------------------------- */
const char* tramp_version(void);
/* Number of visible HIP devices (0 without a GPU; never fails hard). */
int tramp_device_count(int* n);
/* One context per GPU per host thread: owns a HIP stream and a reduction
 * workspace; holds no globals, re-entrant across contexts. */
int tramp_ctx_create(int device, tramp_ctx** ctx);
int tramp_ctx_destroy(tramp_ctx* ctx);
/* HIP's current device is per host thread: when one thread drives contexts on several GPUs it
 * makes a context current before using it (the allocating entry points do so themselves). */
int tramp_ctx_make_current(tramp_ctx* ctx);
const char* tramp_last_error(const tramp_ctx* ctx);
int tramp_sync(tramp_ctx* ctx);
/* Raw hipStream_t of the context (for interop with other HIP users). */
int tramp_ctx_stream(tramp_ctx* ctx, void** stream);
int tramp_device_name(tramp_ctx* ctx, char* out, size_t cap);
int tramp_mem_info(tramp_ctx* ctx, size_t* free_bytes, size_t* total_bytes);

/* ---- device memory ------------------------------------------------------ */
int tramp_buf_alloc(tramp_ctx* ctx, size_t bytes, void** dptr);
int tramp_buf_free(tramp_ctx* ctx, void* dptr);
int tramp_h2d(tramp_ctx* ctx, void* dst_d, const void* src_h, size_t bytes);
int tramp_d2h(tramp_ctx* ctx, void* dst_h, const void* src_d, size_t bytes); /* synchronises */
int tramp_d2d(tramp_ctx* ctx, void* dst_d, const void* src_d, size_t bytes);
int tramp_memset(tramp_ctx* ctx, void* dst_d, int value, size_t bytes);
/* out[i] = value for a typed vector / a double scalar array. */
int tramp_fill(tramp_ctx* ctx, int dtype, void* dst_d, size_t count, double value);
int tramp_fill_f64(tramp_ctx* ctx, double* dst_d, size_t count, double value);
/* dtype conversion of a vector (f64 <-> f32), count elements. */
int tramp_convert(tramp_ctx* ctx, int dst_dtype, void* dst_d, int src_dtype,
                  const void* src_d, size_t count);

/* ---- timing (HIP events on the context's stream; bench.py roofline) ------ */
int tramp_event_create(tramp_ctx* ctx, tramp_event** ev);
int tramp_event_destroy(tramp_ctx* ctx, tramp_event* ev);
int tramp_event_record(tramp_ctx* ctx, tramp_event* ev);
/* Synchronises on `stop`; milliseconds between the two recorded events. */
int tramp_event_elapsed_ms(tramp_ctx* ctx, tramp_event* start, tramp_event* stop, float* ms);

/* ---- hipGraph capture of a launch sequence (launch-bound inner loops) --------
 * Between tramp_graph_begin and tramp_graph_end every launch on the context's stream is
 * recorded instead of executed; the resulting executable graph replays the whole sequence
 * (e.g. one EP iteration: all factor updates + variable posteriors) with one host call.
 * Everything the sequence needs must have been allocated by an eager run before (capture
 * forbids allocation and synchronisation); buffers and scalar arguments are baked in. */
typedef struct tramp_graph tramp_graph;
int tramp_graph_begin(tramp_ctx* ctx);
int tramp_graph_end(tramp_ctx* ctx, tramp_graph** graph);
int tramp_graph_launch(tramp_ctx* ctx, tramp_graph* graph);
int tramp_graph_destroy(tramp_ctx* ctx, tramp_graph* graph);

/* Per-kernel timing for the roofline report: while enabled, every launch group is
 * bracketed by HIP events on the context's stream.  tramp_profile_report synchronises
 * and returns, per tag (TRAMP_PROF_*), the number of launch groups, their summed
 * duration in ms and the summed ALGORITHMIC bytes / flops the library attributed to
 * them (matrix passes: R*C*sizeof(T) bytes, 2*R*C*batch flops; elementwise updates:
 * 2T or 3T bytes per component).  Arrays hold TRAMP_PROF_NTAGS entries. */
#define TRAMP_PROF_EW 0
#define TRAMP_PROF_MATVEC_NT 1
#define TRAMP_PROF_MATVEC_NN 2
#define TRAMP_PROF_LIN_SCALARS 3
#define TRAMP_PROF_VAR_POST 4
#define TRAMP_PROF_OTHER 5
#define TRAMP_PROF_FFT 6          /* rocFFT transforms of the convolution channel */
#define TRAMP_PROF_NTAGS 7
int tramp_profile_enable(tramp_ctx* ctx, int on);
int tramp_profile_report(tramp_ctx* ctx, int* count, double* ms, double* bytes, double* flops);

/* ---- priors (root factors) ---------------------------------------------- */
/* K1. GaussBernoulliPrior.compute_forward_posterior + Prior.compute_forward_message
 *     tramp/priors/gauss_bernoulli_prior.py:70-77, tramp/priors/base_prior.py:13-16,
 *     tramp/beliefs/sparse.py:9-22, tramp/beliefs/normal.py:3-4.
 * in : ax_d[batch], bx_d[batch,n]; prior (rho, mean, var)
 * out: a_new_d[batch], b_new_d[batch,n]; optional r_d[batch,n], v_d[batch] (posterior). */
int tramp_gb_prior_fwd(tramp_ctx* ctx, int dtype, size_t n, size_t batch,
                       const double* ax_d, const void* bx_d,
                       double rho, double mean, double var,
                       double damping, const double* a_old_d, const void* b_old_d,
                       double* a_new_d, void* b_new_d, void* r_d, double* v_d);

/* K2. MAP_L1NormPrior.compute_forward_posterior (soft threshold)
 *     tramp/priors/map_L1_norm_prior.py:11-17,66-71.
 * count_d[batch] (optional) receives the exact support size #{|bx| > gamma};
 * mask_d[batch,n] (optional, uint8) the support mask itself. */
int tramp_l1_prior_fwd(tramp_ctx* ctx, int dtype, size_t n, size_t batch,
                       const double* ax_d, const void* bx_d, double gamma,
                       double damping, const double* a_old_d, const void* b_old_d,
                       double* a_new_d, void* b_new_d, void* r_d, double* v_d,
                       unsigned long long* count_d, uint8_t* mask_d);

/* K3. GaussianPrior.compute_forward_message: constant message, bypasses ab_new
 *     tramp/priors/gaussian_prior.py:86-89 (posterior :63-68). */
int tramp_gauss_prior_fwd(tramp_ctx* ctx, int dtype, size_t n, size_t batch,
                          double mean, double var, double* a_new_d, void* b_new_d);

/* ---- likelihoods (leaf factors, y observed) ------------------------------- */
/* K6. GaussianLikelihood.compute_backward_message: constant (1/var, y/var)
 *     tramp/likelihoods/gaussian_likelihood.py:68-71,:17. */
int tramp_gauss_lik_bwd(tramp_ctx* ctx, int dtype, size_t n, size_t batch,
                        const void* y_d, double var, double* a_new_d, void* b_new_d);

/* K4. SgnLikelihood.compute_backward_posterior + Likelihood.compute_backward_message
 *     tramp/likelihoods/sgn_likelihood.py:32-37, tramp/likelihoods/base_likelihood.py:25-28,
 *     tramp/beliefs/positive.py:12-17, tramp/utils/truncated_normal.py:92-94,147-149,224-266.
 * y_d[batch,n] in {-1,+1}. */
int tramp_sgn_lik_bwd(tramp_ctx* ctx, int dtype, size_t n, size_t batch,
                      const double* az_d, const void* bz_d, const void* y_d,
                      double damping, const double* a_old_d, const void* b_old_d,
                      double* a_new_d, void* b_new_d, void* r_d, double* v_d);

/* Sign observations packed one bit per component for tramp_sgn_lik_bwd_bits: y_d[batch][n] in {-1, +1} ->
 * bits_d[batch][8 ceil(n / 64)] bytes, bit j of byte q of a row = (y[8 q + j] == +1).  *bad_h = entries that are
 * neither -1 nor +1 (then the packed form must not be used).  Synchronises the stream (done once, at construction:
 * SgnLikelihood holds a static observation, tramp/likelihoods/sgn_likelihood.py:14-19). */
int tramp_sgn_pack(tramp_ctx* ctx, int dtype, size_t n, size_t batch, const void* y_d, void* bits_d,
                   unsigned long long* bad_h);
/* tramp_sgn_lik_bwd (sgn_likelihood.py:32-37 + base_likelihood.py:25-28) with the packed observation beside y_d:
 * batches of large instances stream bz and one BIT of y per component through the resident pipeline (2T + n/8 bytes
 * moved instead of 3T); any other shape, a wanted posterior (r_d) or an armed producer hint takes tramp_sgn_lik_bwd on
 * y_d.  bits_d may be NULL. */
int tramp_sgn_lik_bwd_bits(tramp_ctx* ctx, int dtype, size_t n, size_t batch, const double* az_d, const void* bz_d,
                           const void* y_d, const void* bits_d, double damping, const double* a_old_d,
                           const void* b_old_d, double* a_new_d, void* b_new_d, void* r_d, double* v_d);
/* K5. AbsLikelihood.compute_backward_posterior  (y >= 0)
 *     tramp/likelihoods/abs_likelihood.py:31-36, tramp/beliefs/binary.py:8-13. */
int tramp_abs_lik_bwd(tramp_ctx* ctx, int dtype, size_t n, size_t batch,
                      const double* az_d, const void* bz_d, const void* y_d,
                      double damping, const double* a_old_d, const void* b_old_d,
                      double* a_new_d, void* b_new_d, void* r_d, double* v_d);

/* ReluLikelihood (slope_neg = 0) / LeakyReluLikelihood (slope_neg = s > 0): SURVEY section 8f rank 2
 *     tramp/likelihoods/piecewise_linear_likelihood.py:9-63,201-215,279-291.
 * y > 0 pins z = y (variance 0); Relu with y == 0: truncated normal on (-inf, 0];
 * Leaky with y < 0: z = y / s.  An observation no region contains yields NaN, as the reference. */
int tramp_relu_lik_bwd(tramp_ctx* ctx, int dtype, size_t n, size_t batch, double slope_neg,
                       const double* az_d, const void* bz_d, const void* y_d,
                       double damping, const double* a_old_d, const void* b_old_d,
                       double* a_new_d, void* b_new_d, void* r_d, double* v_d);

/* ---- componentwise channels ---------------------------------------------- */
/* K7. PiecewiseLinearChannel.compute_forward/backward_posterior + Channel message
 *     tramp/channels/activation/piecewise_linear_channel.py:27-53 (regions :80-106),
 *     tramp/utils/linear_region.py:32-64, tramp/utils/truncated_normal.py:21-29,218-298,
 *     tramp/channels/base_channel.py:9-17.
 * kind: SGN/ABS/RELU; dir: FWD updates the x-side message (a = ax, b = bx), BWD
 * the z-side.  mask_d[batch,n] (optional, uint8): bit0/bit1 = the log_Phi
 * "x < 30" branch mask (truncated_normal.py:27) of the pos/neg region. */
int tramp_pwl_channel(tramp_ctx* ctx, int dtype, int kind, int dir, size_t n, size_t batch,
                      const double* az_d, const void* bz_d,
                      const double* ax_d, const void* bx_d,
                      double damping, const double* a_old_d, const void* b_old_d,
                      double* a_new_d, void* b_new_d, void* r_d, double* v_d,
                      uint8_t* mask_d);

/* K7, generic form: any channel made of the two half-line regions z in [zcut, inf) with
 * x = x0_pos + slope_pos z and z in (-inf, zcut] with x = x0_neg + slope_neg z:
 * LeakyReluChannel (slopes 1 / s; piecewise_linear_channel.py:72-77) and AsymmetricAbsChannel
 * (slopes +1 / -1, zcut = shift; :94-99).  Same arguments otherwise as tramp_pwl_channel. */
int tramp_pwl2_channel(tramp_ctx* ctx, int dtype, int dir, size_t n, size_t batch,
                       double slope_pos, double slope_neg, double x0_pos, double x0_neg, double zcut,
                       const double* az_d, const void* bz_d, const double* ax_d, const void* bx_d,
                       double damping, const double* a_old_d, const void* b_old_d,
                       double* a_new_d, void* b_new_d, void* r_d, double* v_d, uint8_t* mask_d);

/* K7, three regions [z_hi, inf), [z_lo, z_hi], (-inf, z_lo] (order pos, mid, neg in slopes3 / x03):
 * HardTanhChannel, HardSigmoidChannel, SymmetricDoorChannel
 * (tramp/channels/activation/piecewise_linear_channel.py:109-132); the mid region uses the
 * finite-interval truncated normal with its close/neg/pos/other branches
 * (tramp/utils/truncated_normal.py:32-200).  mask_d bits: 0/1 log_Phi mask of pos/neg,
 * 2-3 branch taken by the mid region (0 close, 1 neg, 2 pos, 3 other). */
int tramp_pwl3_channel(tramp_ctx* ctx, int dtype, int dir, size_t n, size_t batch,
                       double z_lo, double z_hi, const double* slopes3, const double* x03,
                       const double* az_d, const void* bz_d, const double* ax_d, const void* bx_d,
                       double damping, const double* a_old_d, const void* b_old_d,
                       double* a_new_d, void* b_new_d, void* r_d, double* v_d, uint8_t* mask_d);

/* Three-region LIKELIHOODS with the same region table as tramp_pwl3_channel (y_d observed):
 * HardTanhLikelihood, HardSigmoidLikelihood, SymmetricDoorLikelihood
 * (tramp/likelihoods/piecewise_linear_likelihood.py:9-63,156-205,293-320): what
 * Model.to_observed turns an observed three-region output channel into.  z_lo == z_hi with a sloped middle region
 * leaves two half lines: AsymmetricAbsLikelihood (:260-265, registry key "a-abs") is slopes (1, any, -1), x0 = 0,
 * z_lo = z_hi = shift.  A sloped region contributes -az z^2/2 + bz z - log|slope| (the reference subtracts
 * log(slope): NaN for the descending branch, so its AsymmetricAbsLikelihood is NaN everywhere - DESIGN 10.3). */
int tramp_pwl3_lik_bwd(tramp_ctx* ctx, int dtype, size_t n, size_t batch,
                       double z_lo, double z_hi, const double* slopes3, const double* x03,
                       const double* az_d, const void* bz_d, const void* y_d,
                       double damping, const double* a_old_d, const void* b_old_d,
                       double* a_new_d, void* b_new_d, void* r_d, double* v_d);

/* ---- diagonal-belief variants (isotropic=False): a is an n-vector too ------
 * Same files as above with the `.mean()` line skipped
 * (gauss_bernoulli_prior.py:75-76, sgn_likelihood.py:35-36, map_L1_norm_prior.py:69-70)
 * and compute_ab_new clipping element-wise (tramp/base.py:252).  `count` flat
 * elements; a vectors have the same dtype as b.  which: 0 = GaussBernoulli
 * (p0=rho,p1=mean,p2=var), 1 = L1 (p0=gamma), 2 = SgnLikelihood, 3 = AbsLikelihood
 * (in2_d = y). Outputs r, v are the posterior, a_new/b_new the message. */
int tramp_diag_factor(tramp_ctx* ctx, int dtype, int which, size_t count,
                      const void* a_d, const void* b_d, const void* in2_d,
                      double p0, double p1, double p2,
                      void* a_new_d, void* b_new_d, void* r_d, void* v_d);
/* The same for every separable factor built here, parameters as an array (the inputs of the reference's own
 * factor tests: tramp/tests/test_priors.py:17-51, tramp/tests/test_likelihoods.py:22-83).  which / params:
 *   TRAMP_DIAG_GB {rho, mean, var}      gauss_bernoulli_prior.py:70-77     TRAMP_DIAG_L1 {gamma}   map_L1_norm_prior.py:66-71
 *   TRAMP_DIAG_SGN_LIK {}  (in2 = y)    sgn_likelihood.py:32-37            TRAMP_DIAG_ABS_LIK {}   abs_likelihood.py:31-36
 *   TRAMP_DIAG_GAUSS_PRIOR {mean, var}  gaussian_prior.py:63-68            TRAMP_DIAG_GAUSS_LIK {var}  gaussian_likelihood.py:43-49
 *   TRAMP_DIAG_BINARY {b0 = 0.5 log(p+/p-)}  binary_prior.py:27,52-58
 *   TRAMP_DIAG_RELU_LIK {slope of the negative side; 0 = Relu}  piecewise_linear_likelihood.py:9-63,279-291
 *   TRAMP_DIAG_PWL3_LIK {z_lo, z_hi, slope pos/mid/neg, x0 pos/mid/neg}  piecewise_linear_likelihood.py:156-218,293-320
 *   TRAMP_DIAG_HALFLINE {a0, b0}  exponential_prior.py:54-60 (a0 = 0, b0 = -1/mean), positive_prior.py:56-63 (a0 = 1, b0 = 0)
 *   TRAMP_DIAG_GM {K, probs[K], means[K], vars[K]} (up to 32 parameters)  gaussian_mixture_prior.py:88-97
 * (the Gaussian factors' MESSAGES are constants in the reference, gaussian_prior.py:86-89: the host mirror
 * returns those; a_new / b_new here are the generic compute_ab_new of the posterior) */
enum { TRAMP_DIAG_GB = 0, TRAMP_DIAG_L1 = 1, TRAMP_DIAG_SGN_LIK = 2, TRAMP_DIAG_ABS_LIK = 3, TRAMP_DIAG_GAUSS_PRIOR = 4,
       TRAMP_DIAG_GAUSS_LIK = 5, TRAMP_DIAG_BINARY = 6, TRAMP_DIAG_RELU_LIK = 7, TRAMP_DIAG_PWL3_LIK = 8,
       TRAMP_DIAG_HALFLINE = 9, TRAMP_DIAG_GM = 10 };
int tramp_diag_factor_p(tramp_ctx* ctx, int dtype, int which, size_t count,
                        const void* a_d, const void* b_d, const void* in2_d,
                        const double* params, int nparams,
                        void* a_new_d, void* b_new_d, void* r_d, void* v_d);

/* Producer-fused moments.  An isotropic prior update needs v = mean_i v_i over the message it
 * receives before it can write anything (tramp/base.py:250-255) - a whole extra read of b.  The
 * kernel that WRITES that message already holds every b_i: announce the consumer before the
 * producing call and, if that kernel supports it (dense channel back-projection / rotated-basis
 * epilogues, the inverse FFT's fused message update), it leaves the partial sums of v_i; the next
 * tramp_gb_prior_fwd / tramp_l1_prior_fwd on exactly (a_d, b_d) then reads b ONCE (2T bytes per
 * component).  kind 1: GaussBernoulli (p0, p1, p2) = (rho, mean, var);
 * kind 2: L1 (p0 = gamma); kind 0 clears.  No producer, no match or TRAMP_PRIOR_FUSE=0: the prior
 * simply runs its own first pass - results agree to reduction-order rounding. */
int tramp_prior_hint(tramp_ctx* ctx, int kind, double p0, double p1, double p2, size_t n, size_t batch,
                     const void* b_d, const double* a_d);
/* The same at the other end of the chain: the message (a_d, b_d) about to be written (the dense channel's forward
 * update, the inverse FFT's fused update) feeds tramp_sgn_lik_bwd (kind 3, sgn_likelihood.py:32-37) or
 * tramp_abs_lik_bwd (kind 4, abs_likelihood.py:31-36) with the static observation y_d: the producer reads y_i once
 * more and leaves sum v_i; the likelihood update is then finalize + ONE pass (read bz and y, write b_new: the 3T
 * algorithmic bytes of SURVEY section 8d) instead of two (5T / 6T moved). */
int tramp_lik_hint(tramp_ctx* ctx, int kind, size_t n, size_t batch, const void* b_d, const double* a_d,
                   const void* y_d);
/* how many producers left moments / how many prior updates consumed them, since the context was made */
int tramp_prior_hint_stats(tramp_ctx* ctx, unsigned long long* armed, unsigned long long* used);
/* Resident single-pass updates (csrc/ew_resident.h): large batches of large instances of the separable factors
 * (tramp_gb_prior_fwd, tramp_l1_prior_fwd, ...) are streamed ONCE through a grid of resident workgroups whose
 * registers hold the window that waits for its instance mean (tramp/base.py:245-255), instead of twice through the
 * tile kernels.  Same entry points, same results to reduction-order rounding, run-to-run reproducible; chosen by shape
 * (TRAMP_EW_RESIDENT=0 turns it off).  How many updates took that form since the context was made: */
int tramp_resident_stats(tramp_ctx* ctx, unsigned long long* runs);
/* A FEW instances too large for one workgroup (BASELINE configs[1] read literally: ONE instance of 10^6 components) are
 * updated in ONE launch: batch x G workgroups, all resident, share the instances, exchange their block sums through
 * self-validating granules and write the update from their registers (ew_kernels.hip: ew_coop; TRAMP_EW_COOP=0 turns it
 * off).  How many updates took that form: */
int tramp_coop_stats(tramp_ctx* ctx, unsigned long long* runs);
/* The pipeline's workgroups hand tile sums to each other and must all be resident at once (checked by an occupancy query
 * before every launch).  Should a hand-off still not arrive within the kernel's deadline (2 s; a CU mask or another
 * process's kernels on the device can do that), the waiting workgroups give up: the update of the affected instances
 * comes back NaN - never a trap, never a wrong number - and every later call of this context takes the tile kernels.
 * *gave_up = 1 once that has happened. */
int tramp_resident_errors(tramp_ctx* ctx, unsigned long long* gave_up);
/* *yes = 1 when updates of `batch` instances of n components in `dtype` take the resident form (a host-side shape
 * test: the scheduler asks before it arms a producer for a consumer that would not use the moments) */
int tramp_resident_covers(tramp_ctx* ctx, int dtype, size_t n, size_t batch, int* yes);

/* ---- dense linear channel x = W z --------------------------------------- */
/* LinearChannel.__init__ (tramp/channels/linear/linear_channel.py:30-46): the
 * factorisation is supplied by the caller (host LAPACK or tramp_linear_factorize)
 * in THIN form, k = min(M, N):
 *    Uk_d  [M, k] row-major  left singular vectors
 *    Vt_d  [k, N] row-major  right singular vectors (rows)
 *    sing_d[k]     singular values (descending), always double
 * The handle keeps the pointers (no copy); the caller owns the buffers. */
int tramp_linear_create(tramp_ctx* ctx, int dtype, size_t M, size_t N, size_t rank,
                        const void* Uk_d, const void* Vt_d, const double* sing_d,
                        tramp_linear** lin);
int tramp_linear_destroy(tramp_ctx* ctx, tramp_linear* lin);

/* Device factorisation (setup path; replaces the two host SVDs of
 * LinearChannel.__init__, tramp/channels/linear/linear_channel.py:8-15,30-46): Gram matrix on the
 * smaller side (rocBLAS), symmetric eigensolve (tramp_eigh), thin factors; singular values below
 * 1e-5 sigma_max are re-resolved from the Gram matrix of their own rows, and the rank uses
 * np.linalg.matrix_rank's tolerance sigma_max * max(M,N) * eps.
 * W_d [M,N] row-major of `dtype`; outputs as tramp_linear_create expects
 * (caller-allocated).  Computation is always carried out in f64; an f32 W is widened panel by panel, never as a whole. */
int tramp_linear_factorize(tramp_ctx* ctx, int dtype, size_t M, size_t N, const void* W_d,
                           void* Uk_d, void* Vt_d, double* sing_d, size_t* rank);
/* Device work space the last tramp_linear_factorize of this context held at most, in bytes (the caller's W, Uk, Vt not
 * included).  f64 input: ~4 k^2 doubles (the Gram matrix and three k x k work matrices of the spectral split); f32 input:
 * the same plus a few GiB of widened panels - W, Uk and Vt are never copied to f64 (round 5). */
int tramp_factorize_peak_bytes(tramp_ctx* ctx, size_t* bytes);
/* The symmetric eigensolver of tramp_linear_factorize on its own (np.linalg.eigh on the Gram
 * matrix): rocSOLVER syevd up to n = TRAMP_EIG_CAP (default 46336, its 32-bit limit), above that a
 * spectral divide and conquer (matrix sign function by Cholesky-form weighted Halley iterations,
 * rocBLAS level 3 only) halves the problem until the halves fit - this is what lets BASELINE
 * configs[3] (k = 50000) be set up on the device.  A_d [n, n] symmetric f64 (upper triangle read),
 * overwritten by the eigenvectors (A_d[i + j n] = component i of eigenvector j); w_d[n] ascending. */
int tramp_eigh(tramp_ctx* ctx, size_t n, double* A_d, double* w_d);

/* K9. LinearChannel.compute_n_eff / compute_forward_variance / compute_backward_variance
 *     tramp/channels/linear/linear_channel.py:58-67,91-105 (same scalars serve SE:
 *     compute_forward_error / compute_backward_error :119-125) + compute_a_new
 *     (tramp/base.py:245-248).  dir FWD: v = vx, a_new from ax; BWD: v = vz, from az. */
int tramp_linear_variance(tramp_ctx* ctx, const tramp_linear* lin, int dir, size_t batch,
                          const double* az_d, const double* ax_d,
                          double* v_d, double* a_new_d, double* n_eff_d);

/* K8. LinearChannel.compute_forward_message / compute_backward_message
 *     tramp/channels/linear/linear_channel.py:69-117 + tramp/channels/base_channel.py:9-17.
 * With t = Vt bz, u = Uk^T bx, D = az + ax sing^2:
 *    FWD  rx = Uk [ sing (t + sing u) / D ]
 *    BWD  rz = bz/az + Vt^T [ (1/D - 1/az) t + sing u / D ]     (k < N)
 *         rz =         Vt^T [ (t + sing u) / D ]                 (k = N)
 * then the fused message update.  bz_d [batch,N], bx_d [batch,M].
 * Batches of 1-2 use HBM-streaming VALU GEMV kernels, larger batches the MFMA kernels.  r_d optional (posterior mean), v_d optional. */
int tramp_linear_message(tramp_ctx* ctx, tramp_linear* lin, int dir, size_t batch,
                         const double* az_d, const void* bz_d,
                         const double* ax_d, const void* bx_d,
                         double damping, const double* a_old_d, const void* b_old_d,
                         double* a_new_d, void* b_new_d, void* r_d, double* v_d, int reuse);
/* `reuse` (bit mask): the handle keeps the projections t = Vt bz and u = Uk^T bx of its
 * previous call.  TRAMP_REUSE_T / TRAMP_REUSE_U tell this call that bz / bx are unchanged
 * since then (the forward and the backward update of one EP iteration share bz), so the
 * corresponding matrix pass is skipped.  The caller is responsible for the claim. */
#define TRAMP_REUSE_T 1
#define TRAMP_REUSE_U 2

/* Express the channel's OUTPUT side in the left singular basis (x' = Uk^T x; needs M <= N so that Uk is
 * square): legitimate when the factor behind the output is rotation invariant - an isotropic
 * GaussianLikelihood, whose message becomes (1/var, Uk^T y / var).  While on, every b / r vector of the
 * output side passed to or returned by tramp_linear_message / _log_partition is the rotated one, and the
 * two passes over Uk leave the iteration (u = bx', rx' = w).  The caller rotates y once
 * (tramp_matvec_nn on Uk) and rotates results back when read (tramp_matvec_nt on Uk). */
/* ---- LinearChannel(precompute_svd=False) ---------------------------------------
 * tramp/channels/linear/linear_channel.py:43-46, 79-82: C = W^T W is kept and (az I + ax C) rz = bz + W^T bx is
 * solved on every call (np.linalg.solve); spectrum = np.linalg.eigvalsh(C) is ASCENDING and singular = spectrum[:rank]
 * keeps its smallest `rank` entries (a quirk of the reference that n_eff and the variances inherit; reproduced).
 * W_d [M, N] row-major in `dtype` is borrowed and must outlive the handle.  create forms C and its spectrum
 * (read back with tramp_lsolve_spectrum into N host doubles); tramp_lsolve_set_rank passes the rank the caller found
 * (np.linalg.matrix_rank(W)) and must precede the first message.  tramp_lsolve_message has the argument meaning of
 * tramp_linear_message (no projection cache).  A matrix that is not positive definite (az = 0 with singular C)
 * returns TRAMP_EINVAL "singular matrix" (numpy raises LinAlgError there).  Not capturable in a graph; N <= 46336. */
typedef struct tramp_lsolve tramp_lsolve;
int tramp_lsolve_create(tramp_ctx* ctx, int dtype, size_t M, size_t N, const void* W_d, tramp_lsolve** out);
int tramp_lsolve_destroy(tramp_ctx* ctx, tramp_lsolve* ls);
int tramp_lsolve_spectrum(tramp_ctx* ctx, const tramp_lsolve* ls, double* spectrum_host);
int tramp_lsolve_set_rank(tramp_ctx* ctx, tramp_lsolve* ls, size_t rank);
int tramp_lsolve_message(tramp_ctx* ctx, tramp_lsolve* ls, int dir, size_t batch, const double* az_d,
                         const void* bz_d, const double* ax_d, const void* bx_d, double damping,
                         const double* a_old_d, const void* b_old_d, double* a_new_d, void* b_new_d, void* r_d,
                         double* v_d);
/* log-partition of the channel (linear_channel.py:127-132) with rz from the solve: out_d [batch] */
int tramp_lsolve_log_partition(tramp_ctx* ctx, tramp_lsolve* ls, size_t batch, const double* az_d, const void* bz_d,
                               const double* ax_d, const void* bx_d, double* out_d);

/* `on` is a bit mask: 1 = the output side as above; 2 = the INPUT side in the right singular basis
 * (z' = Vt z, k numbers per instance): legitimate behind an isotropic zero-mean GaussianPrior with zero
 * initial messages - the component of z orthogonal to the rows of Vt then stays exactly zero - and removes
 * the two passes over Vt (t = bz', rz' = bz' / az + w); z = Vt^T z' when read (tramp_matvec_nn on Vt). */
int tramp_linear_set_rotated(tramp_ctx* ctx, tramp_linear* lin, int on);
/* dst [C, R] = src [R, C]^T (32 x 32 LDS tiles) */
int tramp_transpose(tramp_ctx* ctx, int dtype, size_t R, size_t C, const void* src_d, void* dst_d);
/* Constant damping of a CONSTANT factor's message (tramp/algos/message_passing.py:119-127
 * applied to e.g. GaussianLikelihood's backward message) replaces the message on one side of
 * the channel by  b <- damping * b + (1 - damping) * c  with the same c every iteration.  The
 * projections are linear, so the cached one follows with O(k) work instead of a matrix pass:
 *    p <- damping * p + (1 - damping) * P c ,   P c computed once (again when c_changed != 0).
 * `which` is TRAMP_REUSE_T (the message is bz) or TRAMP_REUSE_U (bx); the handle must hold the
 * projection of the message as it was BEFORE the update (TRAMP_EINVAL otherwise).  Afterwards
 * the caller may claim the corresponding TRAMP_REUSE_* bit in the next tramp_linear_message. */
int tramp_linear_project_affine(tramp_ctx* ctx, tramp_linear* lin, int which, size_t batch,
                                const void* c_d, double damping, int c_changed);

/* ---- EP bookkeeping on the device ------------------------------ */
/* Per problem instance: the NaN guard of MessagePassing.check_message
 * (tramp/algos/message_passing.py:187-209) and the stopping rule of EarlyStoppingEP
 * (tramp/algos/callbacks.py:250-286) as launches behind the sweeps - captured with them in a
 * hipGraph; the host polls the header every k iterations.
 *   A_d [n_edges, batch]  precisions of the factor->variable messages
 *   S_d [3 n_vars, batch] per variable: v, sum (r - r_old)^2, sum r^2   (tramp_variable_posterior)
 *   state_d int32 [4 + 2 batch]: [0] iterations completed (the kernel increments it), [1] instances
 *        still running, [2] NaN code (0, or 1 + edge * batch + instance), [3] instances diverged;
 *        status[batch] (0 running, 1 converged, 2 diverged); n_iter[batch] (stopping iteration)
 *   v_prev_d, v_frozen_d [n_vars, batch], a_frozen_d [n_edges, batch]: the previous iterate's
 *        variances, and what a stopped instance reports (a diverged instance reports the PREVIOUS
 *        iterate's mean and variance, as the reference restores its previous message dag).
 * tol < 0 never converges, max_increase = +inf never gives up (NaN guard only).
 * tramp_ep_freeze copies, for the instances that stopped in the iteration the monitor just closed,
 * the current (converged) or previous (diverged) posterior mean into frozen_d [batch, n]. */
int tramp_ep_monitor(tramp_ctx* ctx, int n_vars, int n_edges, size_t batch, const double* A_d,
                     const double* S_d, double tol, double max_increase, int wait_increase,
                     int* state_d, double* v_prev_d, double* v_frozen_d, double* a_frozen_d,
                     const double* A_prev_d /* precisions before this iteration, or NULL */);
/* "restoring old message dag" per instance (tramp/algos/callbacks.py:277-283, message_passing.py:187-209):
 * the scheduler snapshots the messages before an iteration, tramp_ep_freeze keeps - per message buffer, with
 * (cur, prev, frozen) = (live message, snapshot, frozen message) - what an instance that stops in this iteration
 * would have left behind (the live message if it converged, the snapshot if it gave up), and at the end of the run
 * tramp_ep_restore / tramp_ep_restore_a put those back into the live buffers / the precision table for every
 * stopped instance (all != 0: for every instance - a NaN was found, everybody goes back one iteration). */
int tramp_ep_restore(tramp_ctx* ctx, int dtype, size_t n, size_t batch, const int* state_d,
                     const void* frozen_d, void* live_d, int all);
int tramp_ep_restore_a(tramp_ctx* ctx, int n_edges, size_t batch, const int* state_d, const double* a_frozen_d,
                       double* A_d, int all);
int tramp_ep_freeze(tramp_ctx* ctx, int dtype, size_t n, size_t batch, const int* state_d,
                    const void* r_cur_d, const void* r_prev_d, void* frozen_d);

/* ---- multi-GPU: ensemble shard over ranks, RCCL over xGMI -------- */
/* Replaces the sequential loop over problem instances of run_experiments
 * (tramp/experiments/multiple_experiments.py:30-49): one process per GPU, instances dealt
 * to the ranks, NO per-iteration collective.  librccl is loaded on first use.
 *   tramp_comm_unique_id : rank 0 draws the id (ncclGetUniqueId); the host side carries the
 *                          TRAMP_COMM_ID_BYTES to the other processes (any rendezvous)
 *   tramp_comm_init      : ncclCommInitRank on the context's device (collective call)
 *   tramp_bcast_matrix   : in-place ncclBroadcast of `bytes` from `root` on the context's
 *                          stream - the shared sensing matrix W or its thin factors
 *   tramp_gather_scalars : ncclAllGather of `count` doubles per rank (per-instance n_iter,
 *                          v, mse at the end of a run); recv_d holds world*count
 *   tramp_comm_max       : all-reduce MAX of one host double (elapsed time over ranks),
 *                          synchronises the stream
 *   tramp_comm_barrier   : every rank has finished its queued work */
#define TRAMP_COMM_ID_BYTES 128
typedef struct tramp_comm tramp_comm;
int tramp_comm_available(tramp_ctx* ctx);   /* TRAMP_OK if librccl maps and resolves; no collective, no device call */
int tramp_comm_unique_id(tramp_ctx* ctx, unsigned char* id_out);
int tramp_comm_init(tramp_ctx* ctx, int rank, int world, const unsigned char* id, tramp_comm** comm);
int tramp_comm_destroy(tramp_ctx* ctx, tramp_comm* comm);
int tramp_comm_rank(const tramp_comm* comm, int* rank, int* world);
int tramp_bcast_matrix(tramp_ctx* ctx, tramp_comm* comm, void* buf_d, size_t bytes, int root);
int tramp_gather_scalars(tramp_ctx* ctx, tramp_comm* comm, const double* send_d, size_t count,
                         double* recv_d);
int tramp_comm_max(tramp_ctx* ctx, tramp_comm* comm, double* value);
int tramp_comm_barrier(tramp_ctx* ctx, tramp_comm* comm);

/* Building blocks of K8, exported for tests, profiling and the roofline bench:
 *   NT: out[b, r] = sum_c mat[r, c] * x[b, c]     mat [R, C] row-major
 *   NN: out[b, c] = sum_r mat[r, c] * x[b, r]
 * algo: 0 = auto (the digit copy of mat_d when one is attached, else MFMA from 3 instances on, else the GEMV),
 *       1 = VALU GEMV, 2 = f64 / f32 MFMA (never the digit copy), 3 = the digit copy (error without one). */
int tramp_matvec_nt(tramp_ctx* ctx, int dtype, int algo, size_t R, size_t C, size_t batch,
                    const void* mat_d, const void* x_d, void* out_d);
int tramp_matvec_nn(tramp_ctx* ctx, int dtype, int algo, size_t R, size_t C, size_t batch,
                    const void* mat_d, const void* x_d, void* out_d);
/* Launch shape the last MFMA matrix pass of this context chose: split-K factor, number of 64-instance chunks,
 * and whether sibling chunks of a matrix tile were interleaved in a one-dimensional grid (ensembles > 64).
 * Lets the tests assert that they reach the launch configurations the large runs use. */
int tramp_matvec_last_launch(tramp_ctx* ctx, int* nsplit, int* nchunk, int* flat_chunks);

/* Opt-in engine for the same two passes (csrc/gemm_sliced.h): a resident matrix [R, C] gets a copy in signed radix-256
 * digits (`digits` per element against a per-row power-of-two scale; 0 = by dtype: 7 for f64, 4 for f32; 6 / 7 / 8
 * allowed) laid out in int8 matrix-core operand order, and every pass over `mat_d` then
 * runs on v_mfma_i32_16x16x64_i8 with exact int32 accumulation (x is split the same way per pass); results agree with
 * the exact product to ~2^(2 - 8 digits) max|row| sum|x| - 7 digits: the error of an f64 FMA chain, 8: twenty times
 * below it.  which: bit 0 = the row-dot form (tramp_matvec_nt, (rows/16) x (C/64) x digits KiB), bit 1 = the column
 * form (tramp_matvec_nn: a digit copy of the transpose).  The copy is keyed by the device pointer and must be
 * detached (or re-attached) when the matrix changes or is freed; tramp_ctx_destroy drops all copies.
 * Replaces the same reference products as tramp_matvec_nt / _nn (tramp/channels/linear/linear_channel.py:85-118). */
int tramp_matrix_attach_digits(tramp_ctx* ctx, int dtype, size_t R, size_t C, const void* mat_d, int digits, int which);
/* mat_d == NULL: every copy of the context */
int tramp_matrix_detach_digits(tramp_ctx* ctx, const void* mat_d);
int tramp_matrix_digits_info(tramp_ctx* ctx, const void* mat_d, int* digits, int* which, size_t* bytes);
/* dense passes of this context that ran on a digit copy */
int tramp_sliced_stats(tramp_ctx* ctx, unsigned long long* runs);

/* ---- convolution channel x = w * z ------------------ */
/* ConvChannel (tramp/channels/linear/conv_channel.py:13-149), real signals of 1-3 axes
 * with circular boundary: the operator is diagonal in the Fourier basis, so the update is
 * O(n log n).  filter_d holds the FILTER f (always double, row-major `dims`); the conv
 * weights are w[u] = f[-u], i.e. fft(w) = conj(fft(f)) (conv_channel.py:37-45).  The
 * transforms run through rocFFT (batched real-to-complex plans on the context's stream),
 * the spectral solve, scalar reductions and message update are this library's kernels. */
typedef struct tramp_conv tramp_conv;
int tramp_conv_create(tramp_ctx* ctx, int dtype, int ndim, const size_t* dims, const double* filter_d,
                      tramp_conv** conv);
/* ConvChannel(filter, real=False) (conv_channel.py:25-32): x, w, z complex.  filter_d: planar [2, n] doubles (real parts,
 * then imaginary parts); every message / mean vector of the handle is the planar array [batch, 2, n] of `dtype`, so the
 * vector length per instance is 2 n.  Same entry points afterwards (tramp_conv_variance / apply / message; the spectrum
 * returned by tramp_conv_spectrum is the full one, n bins, in the engine's internal order); tramp_conv_project_affine
 * returns ENOTSUP. */
int tramp_conv_create_complex(tramp_ctx* ctx, int dtype, int ndim, const size_t* dims, const double* filter_d,
                              tramp_conv** conv);
int tramp_conv_destroy(tramp_ctx* ctx, tramp_conv* conv);
/* Half spectrum |fft(f)|^2 (last axis n_last/2+1 bins; *count entries, device copy optional)
 * and the mean of the FULL spectrum (ConvChannel.second_moment, conv_channel.py:66-67). */
int tramp_conv_spectrum(tramp_ctx* ctx, const tramp_conv* conv, double* s2_d, size_t* count, double* mean);
/* ConvChannel.compute_n_eff / compute_backward_variance / compute_forward_variance
 * (conv_channel.py:69-79,111-123) + compute_a_new; argument meaning as tramp_linear_variance. */
int tramp_conv_variance(tramp_ctx* ctx, tramp_conv* conv, int dir, size_t batch,
                        const double* az_d, const double* ax_d,
                        double* v_d, double* a_new_d, double* n_eff_d);
/* ConvChannel.convolve / sample (conv_channel.py:47-63): x = w * z for [batch, n] rows. */
int tramp_conv_apply(tramp_ctx* ctx, tramp_conv* conv, size_t batch, const void* z_d, void* x_d);
/* ConvChannel.compute_backward_mean / compute_forward_mean (conv_channel.py:81-109) with the
 * fused message update; arguments and `reuse` exactly as tramp_linear_message (the cached
 * projections are rfft(bz) and rfft(bx)).  bz_d, bx_d, b_new_d, r_d: [batch, n]. */
/* As tramp_linear_project_affine, for the cached transforms rfft(bz) / rfft(bx). */
int tramp_conv_project_affine(tramp_ctx* ctx, tramp_conv* conv, int which, size_t batch,
                              const void* c_d, double damping, int c_changed);
int tramp_conv_message(tramp_ctx* ctx, tramp_conv* conv, int dir, size_t batch,
                       const double* az_d, const void* bz_d,
                       const double* ax_d, const void* bx_d,
                       double damping, const double* a_old_d, const void* b_old_d,
                       double* a_new_d, void* b_new_d, void* r_d, double* v_d, int reuse);

/* ---- gradient channel x = grad z --------------------- */
/* GradientChannel (tramp/channels/linear/gradient_channel.py:10-118, real=True): one circular
 * forward-difference filter per axis (d = ndim), x = [batch, d, n] for z = [batch, n]; diagonal in
 * the Fourier basis with the summed spectrum sum_i |fft(f_i)|^2.  filters_d: [d][n] doubles
 * (tramp/utils/conv_filters.py:102-119).  Variances: vz = (1 - n_eff) / az, vx = n_eff / (ax d)
 * (gradient_channel.py:85-97); messages as tramp_conv_message without the projection cache:
 * dir BWD writes [batch, n] (message to z), dir FWD [batch, d n] (message to x). */
typedef struct tramp_grad tramp_grad;
int tramp_grad_create(tramp_ctx* ctx, int dtype, int ndim, const size_t* dims, const double* filters_d,
                      tramp_grad** grad);
int tramp_grad_destroy(tramp_ctx* ctx, tramp_grad* grad);
int tramp_grad_spectrum(tramp_ctx* ctx, const tramp_grad* grad, double* s2_d, size_t* count, double* mean);
int tramp_grad_variance(tramp_ctx* ctx, tramp_grad* grad, int dir, size_t batch,
                        const double* az_d, const double* ax_d,
                        double* v_d, double* a_new_d, double* n_eff_d);
/* GradientChannel.convolve / sample (gradient_channel.py:26-37): x[b, i, :] = w_i * z[b, :]. */
int tramp_grad_apply(tramp_ctx* ctx, tramp_grad* grad, size_t batch, const void* z_d, void* x_d);
/* `reuse` as in tramp_conv_message: TRAMP_REUSE_T / _U promise that bz / bx hold what the previous message call on this
 * handle transformed (same batch), so the forward and the backward update of one iteration share their 1 + d transforms. */
int tramp_grad_message(tramp_ctx* ctx, tramp_grad* grad, int dir, size_t batch,
                       const double* az_d, const void* bz_d,
                       const double* ax_d, const void* bx_d,
                       double damping, const double* a_old_d, const void* b_old_d,
                       double* a_new_d, void* b_new_d, void* r_d, double* v_d, int reuse);

/* Two-pass mixed-radix complex FFT of length h = 2^a 5^b (1024 <= h <= ~1.3 10^6) used by the convolution
 * channel for 1-D sizes n = 2 h (csrc/fft_pow2.hip).  Exported for tests and tools: forward maps natural
 * order to the PERMUTED spectrum (position p holds frequency kfreq[p], tramp_fft_order), inverse maps a
 * permuted spectrum (overwritten) back to natural order, unnormalised.  Interleaved complex. */
int tramp_fft_len(tramp_ctx* ctx, int dtype, size_t h, size_t batch, int inverse, void* in_d, void* out_d);
int tramp_fft_order(tramp_ctx* ctx, size_t h, size_t count, unsigned long long* kfreq_host);
/* average milliseconds of `reps` forward and `reps` inverse transforms (tools/bench_fft.py) */
int tramp_fft_bench(tramp_ctx* ctx, int dtype, size_t h, size_t batch, int reps, void* in_d, void* out_d,
                    double* ms_fwd, double* ms_inv);
/* The same engine in 1-3 dimensions (what the 2-D / 3-D convolution and gradient channels run:
 * conv_channel.py:81-109, gradient_channel.py:56-76 call np.fft.fftn there).  real_dims[ndim]: shape of the REAL
 * signal (last axis even); data: `batch` packed signals, i.e. complex arrays [d0][d1][n_last / 2] with
 * z[..][j] = x[..][2j] + i x[..][2j+1].  Forward: every axis transformed, position p of axis a holding frequency
 * kfreq[p] (tramp_fftn_order); inverse (input overwritten): natural order, unnormalised.  ENOTSUP for shapes outside
 * the engine (an axis not 2^a 5^b, n_last / 2 not a multiple of the 128-byte line); tramp_fftn_supported asks without
 * a device call (returns 1 / 0). */
int tramp_fftn(tramp_ctx* ctx, int dtype, int ndim, const size_t* real_dims, size_t batch, int inverse,
               void* in_d, void* out_d);
int tramp_fftn_order(tramp_ctx* ctx, int dtype, int ndim, const size_t* real_dims, int axis,
                     unsigned long long* kfreq_host);
int tramp_fftn_supported(int dtype, int ndim, const size_t* real_dims);

/* ---- unitary channels: DFT and dense rotation ------------------------------- */
/* DFTChannel (tramp/channels/linear/dft_channel.py:7-77, real=True): x = fftn(z, norm="ortho").
 * The channel is unitary, so messages pass through: dir FWD  a_new = az, b_new = F bz as the
 * planar array [batch, 2, n] (real parts, then imaginary parts); dir BWD  a_new = ax,
 * b_new = Re F^-1 bx [batch, n] from the planar [batch, 2, n].  Constant damping fused.
 * (RotationChannel, rotation_channel.py:15-44, is two calls of tramp_matvec_nt / _nn.) */
typedef struct tramp_dft tramp_dft;
int tramp_dft_create(tramp_ctx* ctx, int dtype, int ndim, const size_t* dims, tramp_dft** dft);
int tramp_dft_destroy(tramp_ctx* ctx, tramp_dft* dft);
/* real=False (dft_channel.py:31-45): z is complex as well - b_in of FWD and b_new of BWD are planar [batch, 2, n] */
int tramp_dft_set_complex_input(tramp_ctx* ctx, tramp_dft* dft, int on);
int tramp_dft_message(tramp_ctx* ctx, tramp_dft* dft, int dir, size_t batch,
                      const double* a_in_d, const void* b_in_d,
                      double damping, const double* a_old_d, const void* b_old_d,
                      double* a_new_d, void* b_new_d);

/* ---- variable nodes / scheduler helpers ------------------------------------ */
/* K11. Variable.posterior_rv (tramp/base.py:157-161) for a SISO variable:
 *      r = (b1 + b2) / (a1 + a2), v = 1 / (a1 + a2), plus the two sums the
 *      EarlyStoppingEP rule needs (tramp/algos/callbacks.py:246-247,265-268):
 *      sq_diff[b] = sum (r - r_old)^2, sq_norm[b] = sum r^2.
 *      r_old_d may be NULL (first iteration; sq_diff = sq_norm). r_d may alias r_old_d. */
int tramp_variable_posterior(tramp_ctx* ctx, int dtype, size_t n, size_t batch,
                             const double* a1_d, const void* b1_d,
                             const double* a2_d, const void* b2_d,
                             const void* r_old_d, void* r_d, double* v_d,
                             double* sq_diff_d, double* sq_norm_d);

/* K12. constant damping of an (a, b) message in place
 *      (tramp/algos/message_passing.py:119-127): x = d*old + (1-d)*x. */
int tramp_damp(tramp_ctx* ctx, int dtype, size_t n, size_t batch, double damping,
               const double* a_old_d, const void* b_old_d, double* a_d, void* b_d);
/* three operands: (a, b) = d * (a_old, b_old) + (1 - d) * (a_src, b_src); the output may alias either input (what a
 * scheduler that alternates two sets of message buffers needs for a damped constant message) */
int tramp_damp_to(tramp_ctx* ctx, int dtype, size_t n, size_t batch, double damping, const double* a_old_d,
                  const void* b_old_d, const double* a_src_d, const void* b_src_d, double* a_d, void* b_d);
/* BinaryPrior p(x) = p+ delta(x - 1) + p- delta(x + 1) (tramp/priors/binary_prior.py:8-62, beliefs/binary.py):
 * r = tanh(bx + b0), v = mean(1 - r^2), b0 = 0.5 log(p+ / p-); outputs optional as in tramp_gb_prior_fwd. */
int tramp_binary_prior_fwd(tramp_ctx* ctx, int dtype, size_t n, size_t batch,
                           const double* ax_d, const void* bx_d, double b0,
                           double damping, const double* a_old_d, const void* b_old_d,
                           double* a_new_d, void* b_new_d, void* r_d, double* v_d);
/* ExponentialPrior p(x) = 1[x >= 0] exp(-x / mean) / mean (tramp/priors/exponential_prior.py:8-80: a0 = 0, b0 = -1/mean)
 * and PositivePrior p(x) = 2 1[x >= 0] N(x | 0, 1) (tramp/priors/positive_prior.py:8-82: a0 = 1, b0 = 0), one kernel:
 * the posterior is the half-line truncated normal positive.r / positive.v (tramp/beliefs/positive.py:12-17) at
 * (ax + a0, bx + b0), then compute_ab_new; outputs optional as in tramp_gb_prior_fwd. */
int tramp_halfline_prior_fwd(tramp_ctx* ctx, int dtype, size_t n, size_t batch,
                             const double* ax_d, const void* bx_d, double a0, double b0,
                             double damping, const double* a_old_d, const void* b_old_d,
                             double* a_new_d, void* b_new_d, void* r_d, double* v_d);
/* GaussianMixturePrior p(x) = sum_k p_k N(x | m_k, s_k), K <= 8 components given as HOST arrays
 * (tramp/priors/gaussian_mixture_prior.py:8-139, tramp/beliefs/mixture.py:18-47): softmax over the components'
 * log-partitions, r = sum s_k r_k, v = mean_i [sum s_k / a_k + sum_{k<l} s_k s_l (r_k - r_l)^2]. */
int tramp_gm_prior_fwd(tramp_ctx* ctx, int dtype, size_t n, size_t batch,
                       const double* ax_d, const void* bx_d, int K, const double* probs_h, const double* means_h,
                       const double* vars_h, double damping, const double* a_old_d, const void* b_old_d,
                       double* a_new_d, void* b_new_d, void* r_d, double* v_d);
/* ModulusLikelihood y = |z|, z complex (tramp/likelihoods/modulus_likelihood.py:36-131): the output factor of phase
 * retrieval.  Complex vectors are PLANAR real arrays (tramp/utils/misc.py:13-27): bz_d, b_old_d, b_new_d, r_d are
 * [batch, 2, n] (plane 0 = Re, plane 1 = Im), y_d is [batch, n].  rz = (bz / |bz|) y I1(|bz| y) / I0(|bz| y),
 * v = mean_i y_i^2 (1 - I^2) / 2 (:82-93), then compute_ab_new on the planar array.  Outputs optional as above. */
int tramp_modulus_lik_bwd(tramp_ctx* ctx, int dtype, size_t n, size_t batch,
                          const double* az_d, const void* bz_d, const void* y_d,
                          double damping, const double* a_old_d, const void* b_old_d,
                          double* a_new_d, void* b_new_d, void* r_d, double* v_d);
/* MAP_L21NormPrior.compute_forward_posterior + compute_ab_new (tramp/priors/map_L21_norm_prior.py:7-20,69-74),
 * axis = 0: bx is [batch, d, n], groups are the d components of a site (the total-variation prior on a
 * gradient field).  count_d[batch]: sites with ||b[:, j]|| > gamma; mask_d[batch, n]: that support, bit-exact
 * in f64.  Outputs optional as in tramp_l1_prior_fwd. */
int tramp_l21_prior_fwd(tramp_ctx* ctx, int dtype, int d, size_t n, size_t batch, double gamma,
                        const double* ax_d, const void* bx_d, double damping,
                        const double* a_old_d, const void* b_old_d,
                        double* a_new_d, void* b_new_d, void* r_d, double* v_d,
                        unsigned long long* count_d, unsigned char* mask_d);
/* GaussianChannel x = z + N(0, var) as a hidden factor (tramp/channels/noise/gaussian_channel.py:23-33):
 * (a_new, b_new) = k (a_in, b_in), k = (1/var) / (1/var + a_in); the forward message takes (az, bz), the
 * backward one (ax, bx).  Constant damping fused (old may alias new). */
int tramp_gauss_channel(tramp_ctx* ctx, int dtype, size_t n, size_t batch, double var,
                        const double* a_in_d, const void* b_in_d, double damping,
                        const double* a_old_d, const void* b_old_d, double* a_new_d, void* b_new_d);
/* BiasChannel x = z + bias (tramp/channels/linear/bias_channel.py:21-29): dir FWD takes (az, bz) and returns
 * (az, bz + az bias); dir BWD takes (ax, bx) and returns (ax, bx - ax bias).  bias_d: [n], shared by the
 * ensemble.  Constant damping fused (old may alias new). */
int tramp_bias_channel(tramp_ctx* ctx, int dtype, int dir, size_t n, size_t batch, const void* bias_d,
                       const double* a_in_d, const void* b_in_d, double damping,
                       const double* a_old_d, const void* b_old_d, double* a_new_d, void* b_new_d);
/* SumChannel x = z_1 + ... + z_K (tramp/channels/linear/sum_channel.py:26-43), K <= 8: az_d / bz_d are host
 * arrays of K device pointers (the messages z_k -> factor).  target = -1: forward message to x
 * (1 / v_bar, r_bar / v_bar); target = k: backward message to z_k, which also needs the message (ax, bx)
 * from x.  Constant damping fused (old may alias new). */
int tramp_sum_channel(tramp_ctx* ctx, int dtype, size_t n, size_t batch, int k,
                      const double* const* az_d, const void* const* bz_d,
                      const double* ax_d, const void* bx_d, int target, double damping,
                      const double* a_old_d, const void* b_old_d, double* a_new_d, void* b_new_d);
/* DuplicateChannel x_1 = ... = x_K = z (tramp/channels/shape/duplicate_channel.py:24-31), K <= 8: ax_d / bx_d are host
 * arrays of K device pointers (the messages x_k -> factor), (az, bz) the message z -> factor.  target = -1: backward
 * message to z; target = k: forward message to x_k.  Constant damping fused (old may alias new). */
int tramp_duplicate_channel(tramp_ctx* ctx, int dtype, size_t n, size_t batch, int k, const double* az_d,
                            const void* bz_d, const double* const* ax_d, const void* const* bx_d, int target,
                            double damping, const double* a_old_d, const void* b_old_d, double* a_new_d, void* b_new_d);
/* Variable.forward_message / backward_message on a tree (tramp/base.py:186-211): out = sum of the k
 * incoming messages (host arrays of k device pointers, k <= 8) minus message `exclude` (the target's own;
 * -1: plain sum = Variable.posterior_ab, base.py:152-155). */
int tramp_message_sum(tramp_ctx* ctx, int dtype, size_t n, size_t batch, int k,
                      const double* const* a_in_d, const void* const* b_in_d, int exclude,
                      double* a_out_d, void* b_out_d);

/* sum_d[batch] = sum_i (x[b,i] - y[b,i])^2  (y may be NULL): mse / norms for callbacks
 * (tramp/algos/metrics.py:5-8). */
int tramp_sq_dist(tramp_ctx* ctx, int dtype, size_t n, size_t batch,
                  const void* x_d, const void* y_d, double* sum_d);

/* ---- state evolution (scalar recursions; host scalars in, host scalars out) --
 * Each call evaluates `count` independent parameter points in one launch.
 * K13. Prior.compute_forward_error for the Gauss-Bernoulli prior
 *      tramp/priors/base_prior.py:66-69, gauss_bernoulli_prior.py:59-62,116-122,
 *      tramp/utils/integration.py:13-28.  All arrays are HOST arrays[count]. */
/* integrand evaluations spent by the last state-evolution quadrature call (K13: graded composite
 * Gauss-Legendre, fixed count; K14: adaptive panel refinement, data dependent) - for bench.py */
int tramp_se_last_evals(tramp_ctx* ctx, unsigned long long* evals);
int tramp_se_gb_prior(tramp_ctx* ctx, size_t count, const double* ax,
                      const double* rho, const double* mean, const double* var, double* v);
/* K15. AnalyticalLinearChannel / MarchenkoPasturChannel closed forms
 * (tramp/channels/linear/analytical_linear_channel.py:28-53, tramp/ensembles/marchenko_pastur_ensemble.py:39-45):
 * dir BWD v = (1 - n_eff) / max(az, 1e-11), dir FWD v = n_eff / (alpha ax) (mean_spectrum / (alpha az) at ax = 0),
 * n_eff = 1 - eta(ax / az).  Host arrays of `count` points; n_eff optional. */
int tramp_se_mp_channel(tramp_ctx* ctx, int dir, size_t count, const double* alpha, const double* mean_spectrum,
                        const double* az, const double* ax, double* v, double* n_eff);
/* BinaryPrior.compute_forward_error (tramp/priors/binary_prior.py:82-86, base_prior.py:66-69):
 * v(ax) = p+ E_{N(ax, ax)}[1 - tanh^2(b + b0)] + p- E_{N(-ax, ax)}[.], host arrays of `count` points. */
int tramp_se_binary_prior(tramp_ctx* ctx, size_t count, const double* ax, const double* p_pos, double* v);
/* GaussianMixturePrior.compute_forward_error (tramp/priors/gaussian_mixture_prior.py:132-137, base_prior.py:66-69):
 * v(ax) = sum_k p_k E_{N(ax m_k, ax + ax^2 s_k)}[v(ax, b)], host arrays (`count` points ax; K <= 8 components). */
int tramp_se_gm_prior(tramp_ctx* ctx, size_t count, const double* ax, int K, const double* probs,
                      const double* means, const double* vars, double* v);
/* K13. Likelihood.compute_backward_error for SgnLikelihood (1-D) and AbsLikelihood (2-D)
 *      tramp/likelihoods/base_likelihood.py:78-81, sgn_likelihood.py:79-92,
 *      abs_likelihood.py:60-69.  Requires az > 1/tau_z (TRAMP_EINVAL otherwise,
 *      mirroring the reference's AssertionError). */
int tramp_se_sgn_lik(tramp_ctx* ctx, size_t count, const double* az, const double* tau_z, double* v);
int tramp_se_abs_lik(tramp_ctx* ctx, size_t count, const double* az, const double* tau_z, double* v);
/* ModulusLikelihood.compute_backward_error (tramp/likelihoods/modulus_likelihood.py:101-120, base_likelihood.py:78-81):
 * the 2-D average of (2 pi / sqrt(u)) relu(b) relu(y) ive(0, b y) y^2 (1 - (I1/I0)(b y)^2) / 2, u = az tau_z - 1 > 0,
 * by the adaptive panel refinement of K14; u <= 0 is the closed form 1 / az of the reference's 1-D branch. */
int tramp_se_modulus_lik(tramp_ctx* ctx, size_t count, const double* az, const double* tau_z, double* v);
/* ReluLikelihood SE error (only the flat region has a variance; 1-D quadrature). */
int tramp_se_relu_lik(tramp_ctx* ctx, size_t count, const double* az, const double* tau_z, double* v);
/* K14. Channel.compute_forward_error / compute_backward_error for Sgn/Abs/Relu channels
 *      tramp/channels/base_channel.py:29-41, piecewise_linear_channel.py:60-64,
 *      tramp/utils/linear_region.py:78-101, tramp/utils/integration.py:31-73. */
int tramp_se_pwl_channel(tramp_ctx* ctx, int kind, int dir, size_t count,
                         const double* az, const double* ax, const double* tau_z, double* v);

int tramp_se_pwl2_channel(tramp_ctx* ctx, int dir, size_t count, double slope_pos, double slope_neg,
                          double x0_pos, double x0_neg, double zcut,
                          const double* az, const double* ax, const double* tau_z, double* v);

int tramp_se_pwl3_channel(tramp_ctx* ctx, int dir, size_t count, double z_lo, double z_hi,
                          const double* slopes3, const double* x03,
                          const double* az, const double* ax, const double* tau_z, double* v);

/* SE error of the three-region likelihoods (piecewise_linear_likelihood.py:120-146,210-214). */
int tramp_se_pwl3_lik(tramp_ctx* ctx, size_t count, double z_lo, double z_hi,
                      const double* slopes3, const double* x03,
                      const double* az, const double* tau_z, double* v);

/* ---- log-partitions -------------------------------- */
/* compute_log_partition of the round-6 priors, out[b] = SUM over the n components (the host takes the mean):
 *   which 0, params_h = {a0, b0, A0}: positive.A(ax + a0, bx + b0) - A0 - ExponentialPrior (exponential_prior.py:62-65,
 *     A0 = exponential.A(b0) = -log(-b0)) and PositivePrior (positive_prior.py:65-69, A0 = positive.A(1, 0));
 *   which 1, params_h = {K, probs[K], means[K], vars[K]}: GaussianMixturePrior (gaussian_mixture_prior.py:99-104);
 *   which 2, params_h = {b0}: BinaryPrior (binary_prior.py:64-67): binary.A(bx + b0) - binary.A(b0) - ax / 2. */
int tramp_prior2_log_partition(tramp_ctx* ctx, int dtype, int which, size_t n, size_t batch, const double* params_h,
                               int nparams, const double* ax_d, const void* bx_d, double* out_d);
/* ModulusLikelihood.compute_log_partition AS FOUND (modulus_likelihood.py:125-129: np.absolute acts on each real
 * coordinate of the planar array): out[b] = sum over both planes of -az y^2 / 2 + log(2 pi y) + log I0(|b| y);
 * the host divides by 4 n (A.mean() / 2). */
int tramp_modulus_log_partition(tramp_ctx* ctx, int dtype, size_t n, size_t batch, const double* az_d,
                                const void* bz_d, const void* y_d, double* out_d);
/* Log-partition of a factor given as a TABLE OF LINEAR REGIONS - the rest of the piecewise-linear family (LeakyRelu,
 * AsymmetricAbs, HardTanh, HardSigmoid, SymmetricDoor and their observed twins), restated as the reference loops:
 * regions_h = nreg (2 or 3) host rows (zmin, zmax, slope, x0) in the reference's order, +-INFINITY for open ends.
 * mode 0, a channel: A_k = truncated_normal_logZ(b/a, 1/a, zmin, zmax) - ax x0^2 / 2 + bx x0 with a = az + slope^2 ax,
 *   b = bz + slope (bx - ax x0) (tramp/utils/linear_region.py:55-64); out[b] = sum_i logsumexp_k A_k
 *   (tramp/channels/activation/piecewise_linear_channel.py:55-58); v1_d = bx.
 * mode 1, a likelihood: a flat region holds truncated.A(az, bz, zmin, zmax) where y == x0, a sloped one
 *   -az z^2 / 2 + bz z - log|slope| where z = (y - x0)/slope lies strictly inside, -inf otherwise
 *   (tramp/likelihoods/piecewise_linear_likelihood.py:52-61); out[b] = sum_i logsumexp_k (the host takes the mean,
 *   :216-219); v1_d = y, ax_d unused.  (The reference writes log(slope): NaN on a descending branch.) */
int tramp_region_log_partition(tramp_ctx* ctx, int dtype, int mode, int nreg, const double* regions_h, size_t n,
                               size_t batch, const double* az_d, const void* bz_d, const double* ax_d,
                               const void* v1_d, double* out_d);
/* Terms of the EP log-evidence (tramp/algos/message_passing.py:306-328): out_d[b] = SUM over the
 * n components of the factor's / variable's log-partition, formed in double.  The caller divides
 * by n where the reference takes the mean (priors, likelihoods: e.g. gauss_bernoulli_prior.py:79-83,
 * sgn_likelihood.py:39-41) and keeps the sum for channels and variables
 * (piecewise_linear_channel.py:55-58, tramp/base.py:146-150).
 *   kind            a0_d, v0_d     a1_d, v1_d     p0, p1, p2
 *   GB_PRIOR        ax, bx         -              rho, mean, var
 *   GAUSS_PRIOR     ax, bx         -              mean, var
 *   L1_PRIOR        ax, bx         -              gamma
 *   GAUSS_LIK       az, bz         -, y           var
 *   SGN_LIK/ABS_LIK az, bz         -, y
 *   PWL_SGN/ABS/RELU az, bz        ax, bx
 *   VARIABLE        a_fwd, b_fwd   a_bwd, b_bwd                                         */
#define TRAMP_LOGZ_GB_PRIOR 0
#define TRAMP_LOGZ_GAUSS_PRIOR 1
#define TRAMP_LOGZ_L1_PRIOR 2
#define TRAMP_LOGZ_GAUSS_LIK 3
#define TRAMP_LOGZ_SGN_LIK 4
#define TRAMP_LOGZ_ABS_LIK 5
#define TRAMP_LOGZ_PWL_SGN 6
#define TRAMP_LOGZ_PWL_ABS 7
#define TRAMP_LOGZ_PWL_RELU 8
#define TRAMP_LOGZ_VARIABLE 9
int tramp_log_partition(tramp_ctx* ctx, int dtype, int kind, size_t n, size_t batch,
                        double p0, double p1, double p2,
                        const double* a0_d, const void* v0_d, const double* a1_d, const void* v1_d,
                        double* out_d);
/* LinearChannel.compute_log_partition (tramp/channels/linear/linear_channel.py:127-132) from the
 * thin factors; `reuse` as in tramp_linear_message (the projections are shared with it). */
int tramp_linear_log_partition(tramp_ctx* ctx, tramp_linear* lin, size_t batch,
                               const double* az_d, const void* bz_d, const double* ax_d, const void* bx_d,
                               int reuse, double* out_d);

/* ---- synthetic data (bench / tests; counter-based, reproducible per seed) --- */
/* out[i] = scale * N(0,1), Philox-4x32 + Box-Muller, element i from counter i. */
int tramp_randn(tramp_ctx* ctx, int dtype, void* out_d, size_t count,
                unsigned long long seed, double scale);

/* instances handled per launch of the VALU GEMV kernels (the dispatcher sends batches
 * of 3 and more to the MFMA kernels, so only forced-VALU calls chunk by this) */
#define TRAMP_GEMV_MAX_BATCH 8

#ifdef __cplusplus
}
#endif
#endif /* TRAMP_HIP_H */
