/*
 * lmm_hip.h -- C ABI of liblmm_hip.so: MI355X (gfx950) ILMM / OILMM inference hot path.
 *
 * The reference (LinearMixingModels.jl 0.1.11) has no FFI; its boundary is Julia multiple dispatch
 * on AbstractGPs' generic functions.  Each entry point below is what ONE reference method body
 * becomes after `ccall`; the method it replaces is cited as  file:line  relative to the reference
 * repository.  INTEGRATION.md shows the Julia-side stubs.
 *
 * Conventions (all entry points)
 *   - Float64, column-major, dense, contiguous (Julia `Array` layout).
 *   - x   : d x n input locations (`Vector{Float64}` => d = 1; `ColVecs(X)` => X).
 *   - y   : length n*p "by-outputs" vector == n x p column-major (MOInputIsotopicByOutputs order,
 *           reference src/ilmm.jl:43 `reshape_y`).
 *   - x, y, xs, z, eps and every OUTPUT array may be HOST or DEVICE (HIP) pointers; the library
 *     detects which (hipPointerGetAttributes).  Small model arrays (U, S, H, gps) are host.
 *   - latents are described by lmm_gp_t (ConstMean + variance * kernel(|x-x'| / lengthscale)).
 *   - `latent_begin, latent_end` select the shard [begin, end) of latent processes this process
 *     (one process per GPU) evaluates; partial results are summed by the caller (RCCL all-reduce
 *     in the Python/Julia host layer).  Use 0, m for the whole model.
 *   - return value: LMM_OK or an lmm_status code; lmm_last_error_string() gives the message,
 *     lmm_last_error_detail() the failing latent and LAPACK-style pivot `info`
 *     (-> Julia `PosDefException(info)`).  Nothing is ever NaN-and-continue.
 *   - calls are blocking; the library never keeps a caller pointer after returning.  Scalar results (log-likelihoods, the
 *     regulariser's residual, pivot info) are written by the kernels straight into a pinned host arena that is mapped into the
 *     device and read after one stream synchronisation (LMM_DIRECT_RESULTS=0 in the environment: device buffers + hipMemcpy).
 *   - reproducibility: by default the last partial scheduling round of a trailing update is split along K and combined with f64
 *     atomics, so results are reproducible to ~1e-13 relative, NOT bitwise; LMM_DETERMINISTIC=1 (environment) disables the split
 *     and makes every result bitwise reproducible (slower tail of the large updates).
 *   - threading: one context per process (= one GPU).  Every entry point takes the context lock, so calls from several
 *     threads are safe but execute one at a time (the context owns ONE set of HIP streams, one device-memory pool and one
 *     pinned staging arena, which concurrent calls would have to share).  Concurrency across GPUs = one process per GPU.
 *   - device pointers produced by another stream (e.g. a PyTorch tensor still being written by torch's stream): call
 *     lmm_stream_wait_caller(that stream) first; the library's streams then order themselves behind it.
 *   - process-global modes: the compute dtype (lmm_set_compute_dtype) and the projection dtype (lmm_set_projection_dtype) are state
 *     of the process's one context, NOT call arguments: they apply to every later call of every thread until changed (a posterior
 *     handle remembers the dtype it was built in and refuses the other one).
 *   - forward progress of the dataflow kernels: potrf_region_kernel runs cooperating workgroups that wait on flags written by other
 *     workgroups of the SAME launch; its deadlock-freedom argument needs "a task waits only for tasks with LOWER indices, which are
 *     running or finished" (one exception: the walker of a matrix waits for the helper of its current row -- a higher index -- only
 *     after it has published everything the lower-indexed helpers need to finish and free their slots).  HIP does not promise that
 *     workgroups start in index order, so by DEFAULT (round 5) a workgroup does not take its task index from blockIdx.x: it takes its
 *     TURN -- a per-matrix counter hands the indices out in order to workgroups that have started; a workgroup whose turn does not
 *     come within 200 us (a lower-indexed workgroup has not started: the device did not dispatch in order) takes the next free index
 *     instead.  The argument then holds in ANY dispatch order, and while the device does dispatch in order every workgroup runs exactly
 *     the task it would have had (values identical; cost 0.1-0.7 % of an evaluation).
 *     lmm_set_strict_progress(0) (LMM_STRICT_PROGRESS=0 at lmm_init) restores the round-4 behaviour: task = blockIdx.x, and the fused
 *     update launches (NODE_FUSE, off by default anyway) allowed -- correct under in-order dispatch, which every AMD GPU to date does.
 *     In both modes every wait is bounded (4 s of the 100-MHz wall clock, or another workgroup's epoch-tagged abort word): should a
 *     launch ever stall, the grid drains and the call returns LMM_ERR_HIP -- no hang, no wrong value.
 */
#ifndef LMM_HIP_H
#define LMM_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  LMM_OK = 0,
  LMM_ERR_DIM = 1,             /* "out dim of x != out dim of f."   (reference src/ilmm.jl:52)          */
  LMM_ERR_NOT_ORTHOGONAL = 2,  /* "`U` is not an orthogonal matrix" (reference src/orthogonal_matrix.jl:22) */
  LMM_ERR_NOT_PD = 3,          /* PosDefException from cholesky                                         */
  LMM_ERR_HIP = 4,
  LMM_ERR_ARG = 5,
  LMM_ERR_UNSUPPORTED = 6,
  LMM_ERR_RCCL = 7             /* a collective failed (lmm_last_error_string carries ncclGetErrorString)             */
} lmm_status;

typedef enum { LMM_KERNEL_SE = 0, LMM_KERNEL_MATERN32 = 1, LMM_KERNEL_MATERN52 = 2 } lmm_kernel_kind;

/* One latent GP: GP(mean, variance * Kernel o ScaleTransform(1/lengthscale)). */
typedef struct {
  int kind;            /* lmm_kernel_kind */
  double variance;
  double lengthscale;
  double mean;         /* ConstMean / ZeroMean */
} lmm_gp_t;

/* The reference's hard-coded numerics constants, passed explicitly so that fp32 callers can
 * widen them (SURVEY.md section 7 "jitter hazards"); pass NULL for the reference values. */
typedef struct {
  double project_jitter;   /* 1e-9  : reference src/ilmm.jl:63                                 */
  double ilmm_rand_jitter; /* 1e-12 : reference src/ilmm.jl:84                                 */
  double default_jitter;   /* 1e-18 : AbstractGPs f(x) default, reference src/oilmm.jl:47,61   */
} lmm_jitters_t;

typedef struct lmm_post lmm_post_t;   /* opaque posterior state (device resident) */

/* ---- lifetime -------------------------------------------------------------------------- */
int lmm_init(int device);                 /* bind this process to HIP device `device`, create streams */
int lmm_shutdown(void);
const char* lmm_last_error_string(void);
int lmm_last_error_detail(int* latent, int* info);
int lmm_device_synchronize(void);
int lmm_release_cached_memory(void);      /* return the caching device-memory pool (factor-matrix slots) to HIP */
/* Compute dtype of the per-latent hot path (SURVEY.md section 8b "dtype selected by symbol suffix or enum"; BASELINE configs[4]).
 * LMM_F64 (default): everything Float64 -- the parity mode (rtol 1e-6 against the reference's CPU path).
 * LMM_F32: the MATRICES -- latent Grams, Cholesky factors, inverse diagonal blocks, cross-solve blocks -- are Float32 and the
 *   trailing updates / TRSMs run on v_mfma_f32_32x32x2_f32 (2x the FP64 matrix rate, half the factor memory); vectors at the
 *   boundary (x, y, normals, outputs), the kernel evaluation, the 64x64 diagonal-block factorisation and all reductions stay
 *   Float64.  Served: OILMM / IndependentMOGP logpdf, posterior, marginals, rand, posterior logpdf, sequential conditioning, the
 *   decoupled dense-H logpdf and (round 3) the OILMM / IndependentMOGP logpdf gradients, prior and predictive
 *   (lmm_oilmm_logpdf_grad, lmm_oilmm_post_logpdf_grad: Float32 factor, triangular inverse and K^-1 on v_mfma_f32, every reduction
 *   Float64; tolerance at sigma2 = 0.1, n ~ 10^3: d/dy, d/dU within 1e-4 of their largest component, d/dsigma2 rtol 1e-4, d/dS and
 *   kernel parameters rtol 2e-3 + 1e-2 absolute -- tests/test_gpu_f32.py) and the dense (mn)x(mn) ILMM logpdf (lmm_ilmm_logpdf[_ex],
 *   lmm_ilmm_logpdf_multi: Float32 (mn)x(mn) matrix, rtol 2e-4 on the value) and (round 4) the dense-H POSTERIOR (reference
 *   src/ilmm.jl:184-198: lmm_ilmm_posterior_create, lmm_ilmm_post_condition, _post_mean_and_var, _post_mean_and_cov, _post_logpdf,
 *   _post_rand, the latent view): Float32 (mn)x(mn) factor, cross-solve block and posterior covariance; means in the rider form
 *   mu + R (L^-1 delta); rtol 2e-4 on means, 1e-3 on variances / covariances / logpdf at sigma2 = 0.1), the full covariance of
 *   independent latents (lmm_lmm_mean_and_cov: Float32 latent covariances, Float64 mixing; entries within 5e-5 of the largest) and the
 *   dense-H GRADIENTS (lmm_ilmm_logpdf_grad, lmm_ilmm_post_logpdf_grad: Float32 (mn)x(mn) factor and explicit inverse; value rtol
 *   2e-5, d/dy within 1e-4, d/dy_train and d/dH within 5e-4 of their largest component, d/dsigma2 rtol 1e-4, kernel parameters
 *   rtol 2e-3 + 1e-2 absolute at sigma2 = 0.1, mn ~ 10^3 -- tests/test_gpu_f32.py).  No entry point refuses the fp32 mode any more.
 *   Jitters stay explicit arguments: the reference's 1e-18 / 1e-12 defaults are below Float32 resolution, so prior sampling
 *   needs a caller-chosen jitter (>= ~1e-5 x kernel variance).  A posterior handle remembers the dtype it was built in. */
typedef enum { LMM_F64 = 0, LMM_F32 = 1 } lmm_dtype;
int lmm_set_compute_dtype(int dtype);
int lmm_get_compute_dtype(void);
/* Strict forward progress of the dataflow kernels (conventions above): 1 (default) = task indices handed out in turn to workgroups
 * that have started, 0 = task = blockIdx.x (relies on in-order dispatch).  Process-global like the dtype modes. */
int lmm_set_strict_progress(int on);
int lmm_get_strict_progress(void);

/* Dtype of the H unprojection of predictive marginals,  M = H M_latent,  V = abs2.(H) V_latent .+ sigma2  (reference
 * src/oilmm.jl:69-72; lmm_oilmm_mean_and_var) -- BASELINE configs[3] "bf16 MFMA covariance projection".
 *   LMM_PROJ_NATIVE (default): Float64 FMAs (the parity mode).
 *   LMM_PROJ_BF16: H (or abs2.(H)) and the latent marginals are rounded to bfloat16 (round-to-nearest-even) and multiplied on
 *     v_mfma_f32_16x16x32_bf16 with Float32 accumulation; sigma2 is added in Float64.  STATED TOLERANCE: each operand carries a
 *     relative rounding error <= 2^-8 (bfloat16 keeps 8 significant bits), so |M - M_f64| <= 2^-7 * sum_l |H[o,l]| |M_latent[l,s]| (plus Float32 accumulation,
 *     ~m 2^-24) and likewise for V; the latent marginals themselves (Gram, Cholesky, triangular solves) stay in the compute dtype.
 *   LMM_PROJ_BF16X2: the same pipe with each operand split into two bfloat16 terms (hi + lo), three MFMA products: error <= 2^-15 * sum_l |H||M_lat| (the dropped lo*lo term and the rounding of the lo parts are ~2^-16; Float32 accumulation on top). */
typedef enum { LMM_PROJ_NATIVE = 0, LMM_PROJ_BF16 = 1, LMM_PROJ_BF16X2 = 2 } lmm_proj_dtype;
int lmm_set_projection_dtype(int dtype);
int lmm_get_projection_dtype(void);

/* Order the library's streams behind everything queued so far on `hip_stream` (a hipStream_t; NULL = the legacy default
 * stream): the NEXT entry point may then be handed device pointers that stream is still producing, or output buffers it is
 * still reading.  Entry points are blocking, so no ordering is needed in the other direction. */
int lmm_stream_wait_caller(void* hip_stream);

/* ---- multi-GPU: one process per GPU, RCCL over xGMI -----------------------------------------
 * The reference has no parallelism; latents shard over ranks (latent_begin/latent_end above) with NO data-path collective,
 * and ONE sum all-reduce finishes logpdf (8 bytes), marginals (2 p n* doubles) or a sample (n p doubles).
 *   rank 0:  lmm_comm_get_unique_id(id)  -> ship the LMM_UNIQUE_ID_BYTES bytes to the other ranks out of band
 *            (MPI.bcast in Julia, a file, torch.distributed's store, ...)
 *   all   :  lmm_comm_init_rank(id, rank, world)   after lmm_init(device); world == 1 is valid (all-reduce = identity)
 *   all   :  lmm_allreduce_sum_f64(buf, count)     in place; buf host or device; blocking                                  */
#define LMM_UNIQUE_ID_BYTES 128
int lmm_comm_get_unique_id(void* id_out);
int lmm_comm_init_rank(const void* id, int rank, int world);
int lmm_comm_info(int* rank, int* world);              /* world = 0 when no communicator exists */
int lmm_allreduce_sum_f64(double* buf, size_t count);
int lmm_allreduce_max_f64(double* buf, size_t count);  /* e.g. max-over-ranks timing */
int lmm_comm_destroy(void);

/* ---- Orthogonal(U, S) validation: reference src/orthogonal_matrix.jl:21-23 -------------- */
int lmm_orthogonal_validate(const double* U, int p, int m);

/* ---- logpdf ------------------------------------------------------------------------------ */
/* logpdf(fx::FiniteGP{<:OILMM}, y): reference src/oilmm.jl:79-93 (+ project :20-30, regulariser
 * :101-113, per-latent generic logpdf).  *out = sum_{l in shard} lml_l + (with_regulariser ? reg : 0). */
int lmm_oilmm_logpdf(const double* x, int d, int n, const double* y, int p,
                     const double* U, const double* S, int m, double sigma2,
                     const lmm_gp_t* gps, int latent_begin, int latent_end, int with_regulariser,
                     double* out);

/* Value and gradient of logpdf(fx::FiniteGP{<:OILMM}, y) -- what `Zygote.gradient(logpdf, fx, y)` differentiates in the
 * reference's tests (test/oilmm.jl:31-32), to be wrapped in a ChainRulesCore.rrule by the
 * Julia shim.  Gradients w.r.t. y (n*p, by-outputs), sigma2, S (m), U (p x m, treated as an unconstrained matrix as Zygote
 * treats the field) and each latent's (variance, lengthscale, mean).  Any grad pointer may be NULL.  Outputs are partial
 * sums over the latent shard; entries of grad_gps outside the shard are 0. */
typedef struct { double variance; double lengthscale; double mean; } lmm_gp_grad_t;
int lmm_oilmm_logpdf_grad(const double* x, int d, int n, const double* y, int p,
                          const double* U, const double* S, int m, double sigma2,
                          const lmm_gp_t* gps, int latent_begin, int latent_end, int with_regulariser,
                          double* out_logpdf, double* grad_y, double* grad_sigma2, double* grad_S, double* grad_U,
                          lmm_gp_grad_t* grad_gps);

/* Value and TOTAL gradient of the predictive logpdf  logpdf(posterior(f(x, sigma2), y)(xs, sigma2_s), ys)  of an OILMM or (U = I,
 * S = 1, with_regulariser = 0) an IndependentMOGP: what the reference differentiates with Zygote.gradient(logpdf, po_x, y*) on its
 * posterior models (test/oilmm.jl:32, test/independent_mogp.jl:66), with the derivatives carried through the posterior
 * (alpha, the factor, the Schur complement).  Gradients w.r.t. y (n*p), ys (ns*p), sigma2 (training noise), sigma2_s (predictive
 * noise), S, U and each latent's (variance, lengthscale, mean).  Any grad pointer may be NULL; partial sums over the shard. */
int lmm_oilmm_post_logpdf_grad(const double* x, int d, int n, const double* y, const double* xs, int ns, const double* ys, int p,
                               const double* U, const double* S, int m, double sigma2, double sigma2_s, const lmm_gp_t* gps,
                               int latent_begin, int latent_end, int with_regulariser, double* out_logpdf, double* grad_y,
                               double* grad_ys, double* grad_sigma2, double* grad_sigma2_s, double* grad_S, double* grad_U,
                               lmm_gp_grad_t* grad_gps);

/* The same after SEQUENTIAL conditioning, posterior(posterior(f(x1, s1), y1)(x2, s2), y2) ... (reference src/oilmm.jl:116-134
 * applied to its own result; nbatch <= 7 batches, each with its OWN noise variance).  x (d x n) and y (n x p, by outputs over the
 * n points) hold the batches' points in conditioning order, n = sum batch_n; grad_batch_sigma2 receives one derivative per batch.
 * Exact conditioning makes this the posterior given all batches at once under per-batch noise, so value and total derivatives are
 * again joint minus marginal.  lmm_oilmm_post_logpdf_grad is this entry with nbatch = 1. */
int lmm_oilmm_post_logpdf_grad_seq(const double* x, int d, int n, const int* batch_n, const double* batch_sigma2, int nbatch,
                                   const double* y, const double* xs, int ns, const double* ys, int p, const double* U,
                                   const double* S, int m, double sigma2_s, const lmm_gp_t* gps, int latent_begin, int latent_end,
                                   int with_regulariser, double* out_logpdf, double* grad_y, double* grad_ys,
                                   double* grad_batch_sigma2, double* grad_sigma2_s, double* grad_S, double* grad_U,
                                   lmm_gp_grad_t* grad_gps);

/* Value and gradient of logpdf(fx::FiniteGP{<:ILMM}, y) for a dense H (reference src/ilmm.jl:150-181; Zygote.gradient(logpdf,
 * ilmmx, y) in test/ilmm.jl:31) w.r.t. y, sigma2, H (p x m, column-major) and each latent's (variance, lengthscale, mean).
 * The reference's dense operation plus the explicit (mn) x (mn) inverse; m*n <= 46000.  Does not shard. */
int lmm_ilmm_logpdf_grad(const double* x, int d, int n, const double* y, int p, const double* H, int m, double sigma2,
                         const lmm_gp_t* gps, const lmm_jitters_t* jit, double* out_logpdf, double* grad_y, double* grad_sigma2,
                         double* grad_H, lmm_gp_grad_t* grad_gps);

/* Value and TOTAL derivatives of logpdf(posterior(f(x, sigma2), y)(xs, sigma2_s), ys) for a dense H -- what
 * Zygote.gradient(logpdf, pi, y_test) differentiates in reference test/ilmm.jl:32 (posterior: src/ilmm.jl:184-198): the joint
 * prior density of (y, ys) under two-block observation noise minus the prior density of y (T y is sufficient for the latents, so
 * the reference's projected posterior is the exact conditional).  Gradients w.r.t. y (n*p), ys (ns*p), sigma2 (training noise),
 * sigma2_s (predictive noise), H (p x m) and each latent's (variance, lengthscale, mean); any grad pointer may be NULL.
 * m*(n + ns) <= 46000.  Does not shard. */
int lmm_ilmm_post_logpdf_grad(const double* x, int d, int n, const double* y, const double* xs, int ns, const double* ys, int p,
                              const double* H, int m, double sigma2, double sigma2_s, const lmm_gp_t* gps, const lmm_jitters_t* jit,
                              double* out_logpdf, double* grad_y, double* grad_ys, double* grad_sigma2, double* grad_sigma2_s,
                              double* grad_H, lmm_gp_grad_t* grad_gps);

/* The same after sequential conditioning (src/ilmm.jl:184-198 applied to its own result): arguments as in
 * lmm_oilmm_post_logpdf_grad_seq; lmm_ilmm_post_logpdf_grad is this entry with nbatch = 1. */
int lmm_ilmm_post_logpdf_grad_seq(const double* x, int d, int n, const int* batch_n, const double* batch_sigma2, int nbatch,
                                  const double* y, const double* xs, int ns, const double* ys, int p, const double* H, int m,
                                  double sigma2_s, const lmm_gp_t* gps, const lmm_jitters_t* jit, double* out_logpdf, double* grad_y,
                                  double* grad_ys, double* grad_batch_sigma2, double* grad_sigma2_s, double* grad_H,
                                  lmm_gp_grad_t* grad_gps);

/* The same for the LATENT view of a dense-H posterior: logpdf(get_latent_gp(posterior(...))(xs, sigma2_s), zs), zs = ns x m by outputs over
 * the m latent processes (reference src/ilmm.jl:39 on the posterior ILMM of :196-197: the coupled PosteriorGP of the IndependentMOGP,
 * whose logpdf Zygote differentiates like any other).  Joint density of the conditioning batches (observed through H) and the latent
 * test block (observed through [I_m; 0], noise sigma2_s I) minus the marginal of the batches.  grad_zs: ns x m; grad_H: through the
 * conditioning batches.  m <= p.  Posteriors conditioned ON latent observations are not served. */
int lmm_ilmm_post_latent_logpdf_grad_seq(const double* x, int d, int n, const int* batch_n, const double* batch_sigma2, int nbatch,
                                         const double* y, const double* xs, int ns, const double* zs, int p, const double* H, int m,
                                         double sigma2_s, const lmm_gp_t* gps, const lmm_jitters_t* jit, double* out_logpdf,
                                         double* grad_y, double* grad_zs, double* grad_batch_sigma2, double* grad_sigma2_s,
                                         double* grad_H, lmm_gp_grad_t* grad_gps);

/* logpdf(fx, Y::AbstractMatrix): one value per column of Y ((n p) x ncol, column-major) from ONE factorisation per latent
 * (the extra columns ride the factorisation as rider rows).  The reference does not overload this (it falls to AbstractGPs'
 * dense generic path); AbstractGPs.TestUtils calls it.  out: ncol values. */
int lmm_oilmm_logpdf_multi(const double* x, int d, int n, const double* Y, int p, int ncol,
                           const double* U, const double* S, int m, double sigma2,
                           const lmm_gp_t* gps, int latent_begin, int latent_end, int with_regulariser,
                           double* out);

/* MOInputIsotopicByFeatures <-> MOInputIsotopicByOutputs reordering of an n*p vector (reference
 * src/independent_mogp.jl:135-159 reorder_by_outputs / its inverse).  to_outputs != 0: out[o n + i] = in[i p + o]. */
int lmm_reorder(const double* in, int n, int p, int to_outputs, double* out);

/* logpdf(fx::FiniteGP{<:ILMM}, y), dense H (p x m): reference src/ilmm.jl:150-163 (+ project :61-68,
 * regulariser :171-181; cov(::IndependentMOGP) src/independent_mogp.jl:60-63): ONE (mn) x (mn)
 * factorisation.  Does not shard (replicas only). */
int lmm_ilmm_logpdf(const double* x, int d, int n, const double* y, int p,
                    const double* H, int m, double sigma2, const lmm_gp_t* gps,
                    const lmm_jitters_t* jit, double* out);

/* Same, with control over the decoupled shortcut: when every latent has the SAME kernel (kind, variance, lengthscale)
 * the latent covariance I (x) K + SigmaT (x) I block-diagonalises under the m x m eigen-rotation of SigmaT (SURVEY.md
 * section 3.2), and the value is obtained from m independent n x n factorisations instead of one (mn) x (mn).
 * allow_decoupled = 0 forces the reference's dense operation; *path_used (may be NULL) = 1 if the shortcut ran.
 * lmm_ilmm_logpdf == allow_decoupled 1. */
int lmm_ilmm_logpdf_ex(const double* x, int d, int n, const double* y, int p,
                       const double* H, int m, double sigma2, const lmm_gp_t* gps,
                       const lmm_jitters_t* jit, int allow_decoupled, int* path_used, double* out);

/* logpdf(fx::FiniteGP{<:ILMM}, Y::AbstractMatrix), dense H: one value per column of Y ((n p) x ncol, column-major) from ONE
 * (mn) x (mn) factorisation (TestUtils on ilmmx, reference test/ilmm.jl:34-37).  out: ncol values. */
int lmm_ilmm_logpdf_multi(const double* x, int d, int n, const double* Y, int p, int ncol, const double* H, int m, double sigma2,
                          const lmm_gp_t* gps, const lmm_jitters_t* jit, double* out);

/* logpdf(ft::FiniteGP{<:IndependentMOGP,<:MOInputIsotopicByOutputs,<:Diagonal{<:Real,<:Fill}}, y):
 * reference src/independent_mogp.jl:74-80.  y is n x m. */
int lmm_mogp_logpdf(const double* x, int d, int n, const double* y, int m, double sigma2,
                    const lmm_gp_t* gps, int latent_begin, int latent_end, double* out);
/* logpdf(ft::FiniteGP{<:IndependentMOGP,<:MOInputIsotopicByFeatures,<:Diagonal{<:Real}}, y) after the reference's
 * reorder_by_outputs (src/independent_mogp.jl:149-159, 222-229): per-point (heteroscedastic) noise variances
 * noise_diag[i + l*n] (by-outputs order, n x m), y n x m by-outputs. */
int lmm_mogp_logpdf_diag(const double* x, int d, int n, const double* y, int m, const double* noise_diag,
                         const lmm_gp_t* gps, int latent_begin, int latent_end, double* out);

/* ---- posterior ---------------------------------------------------------------------------- */
/* posterior(fx::FiniteGP{<:OILMM}, y): reference src/oilmm.jl:116-134.  Keeps, per latent of the
 * shard, the Cholesky factor C_l, alpha_l = C_l \ delta_l and x on the device. */
int lmm_oilmm_posterior_create(const double* x, int d, int n, const double* y, int p,
                               const double* U, const double* S, int m, double sigma2,
                               const lmm_gp_t* gps, int latent_begin, int latent_end,
                               lmm_post_t** out);
/* posterior(ft::IsotropicByOutputsFiniteIndependentMOGP, y): reference src/independent_mogp.jl:119-126. */
int lmm_mogp_posterior_create(const double* x, int d, int n, const double* y, int m, double sigma2,
                              const lmm_gp_t* gps, int latent_begin, int latent_end, lmm_post_t** out);
/* posterior(po(x2, sigma2), y2): condition a posterior OILMM / IndependentMOGP (U = I, S = 1) on further observations
 * (AbstractGPs.TestUtils exercises `posterior` on `po`: reference test/oilmm.jl:34-37, test/independent_mogp.jl:68-76).
 * Returns a NEW handle (the old one stays valid). */
int lmm_post_condition(const lmm_post_t* post, const double* U, const double* S, int p, int m, double sigma2,
                       const double* x2, int d, int n2, const double* y2, lmm_post_t** out);
/* posterior(fx::FiniteGP{<:ILMM}, y), dense H: reference src/ilmm.jl:184-198. */
int lmm_ilmm_posterior_create(const double* x, int d, int n, const double* y, int p,
                              const double* H, int m, double sigma2, const lmm_gp_t* gps,
                              const lmm_jitters_t* jit, lmm_post_t** out);
int lmm_post_destroy(lmm_post_t* post);
/* get_latent_gp(posterior(fx::FiniteGP{<:ILMM}, y)) for a dense H: reference src/ilmm.jl:39 applied to the ILMM of :196-197 -- the
 * latent PosteriorGP{IndependentMOGP}.  Returns a handle that SHARES the device state of `post` and has H = I_m (p = m): the
 * lmm_ilmm_post_* entry points below then answer mean / var / cov / logpdf / rand / posterior for the m latent outputs at
 * MOInputIsotopicByOutputs(xs, m).  Pass jitters {0, sigma2, 0}: with project_jitter = 0 the projection is the identity and the
 * regulariser vanishes (logpdf = the generic Gaussian of the latent posterior + sigma2 I); rand with ilmm_rand_jitter = sigma2 and
 * add_noise = 0 is AbstractGPs' mean + chol(cov + sigma2 I).U' z.  Either handle may be destroyed first. */
int lmm_ilmm_post_latent_view(const lmm_post_t* post, lmm_post_t** out);
/* logpdf(pi(xs, sigma2), ys) on the dense-H posterior ILMM: reference test/ilmm.jl:25 (src/ilmm.jl:150-163 applied to the
 * PosteriorGP latent of :196-197).  One (m ns) x (m ns) factorisation. */
int lmm_ilmm_post_logpdf(const lmm_post_t* post, double sigma2, const double* xs, int d, int ns, const double* ys,
                         const lmm_jitters_t* jit, double* out);
/* rand(rng, pi(xs, sigma2)) on the dense-H posterior ILMM: reference src/ilmm.jl:78-87.  z_lat: m*ns normals (by latents),
 * eps: ns*p normals (by outputs), in the reference's draw order. */
int lmm_ilmm_post_rand(const lmm_post_t* post, double sigma2, int add_noise, const double* xs, int d, int ns,
                       const double* z_lat, const double* eps, const lmm_jitters_t* jit, double* out);
/* mean_and_var / marginals of the dense-H posterior ILMM at xs: reference src/ilmm.jl:108-129,142-145 applied to the
 * PosteriorGP latent of src/ilmm.jl:196-197.  Outputs length ns*p, by-outputs; sigma2 included. */
int lmm_ilmm_post_mean_and_var(const lmm_post_t* post, double sigma2, const double* xs, int d, int ns,
                               const lmm_jitters_t* jit, double* mean_out, double* var_out);
/* mean_and_cov / cov of the dense-H posterior ILMM at xs: reference src/ilmm.jl:132-147 on the PosteriorGP latent
 * (AbstractGPs.TestUtils secondary interface on `pi`, test/ilmm.jl:34-37).  cov_out is (p ns) x (p ns) column-major,
 * by-outputs order, sigma2 on the diagonal; small ns only ((p ns)^2 <= 4e8). */
int lmm_ilmm_post_mean_and_cov(const lmm_post_t* post, double sigma2, const double* xs, int d, int ns,
                               const lmm_jitters_t* jit, double* mean_out, double* cov_out);
/* posterior(pi(x2, sigma2), y2): condition the dense-H posterior ILMM on further observations (reference src/ilmm.jl:184-198
 * applied to the PosteriorGP latent; TestUtils on `pi`, test/ilmm.jl:34-37).  Returns a NEW handle (the old one stays
 * valid); the two batches may carry different sigma2. */
int lmm_ilmm_post_condition(const lmm_post_t* post, double sigma2, const double* x2, int d, int n2, const double* y2,
                            const lmm_jitters_t* jit, lmm_post_t** out);

/* Latent marginals of the (prior if post == NULL, else posterior) latent processes of the shard at xs:
 * mean_lat, var_lat are (m_shard x ns) row-per-latent, i.e. ns x m_shard column-major.  No jitter, no
 * mixing.  AbstractGPs PosteriorGP mean/var reached from reference
 * src/oilmm.jl:61 and src/independent_mogp.jl:50,55. */
int lmm_latent_marginals(const lmm_post_t* post, const lmm_gp_t* gps, int m_shard,
                         const double* xs, int d, int ns, double* mean_lat, double* var_lat);

/* mean_and_var(fx::FiniteGP{<:OILMM}) => marginals / mean / var: reference src/oilmm.jl:57-76.
 * H = U sqrt(S) (pass S == NULL to give a dense H in U: the diagonal-covariance mixing of an ILMM whose
 * latents are independent).  Outputs are the shard's PARTIAL sums over its latents, length ns*p,
 * by-outputs; sigma2 (+ default jitter per latent) is added iff add_noise != 0.
 * post == NULL => prior latents `gps` (m_shard of them, starting at latent_begin).
 * var_out == NULL => means only (AbstractGPs.mean(fx), src/ilmm.jl:142): mu + K(x*, x) alpha per latent, no triangular solve. */
int lmm_oilmm_mean_and_var(const lmm_post_t* post, const lmm_gp_t* gps,
                           const double* U, const double* S, int p, int m,
                           int latent_begin, int latent_end, double sigma2, int add_noise,
                           const double* xs, int d, int ns, const lmm_jitters_t* jit,
                           double* mean_out, double* var_out);

/* mean_and_cov(fx) / cov(fx): reference src/ilmm.jl:132-139,147 (+ src/independent_mogp.jl:60-63 through H = I) for
 * independent latents (OILMM prior or posterior, dense-H prior with S == NULL):
 *   C[(o,i),(o',j)] = sum_l H[o,l] H[o',l] (Cov_l[i,j] + jitter [i==j]) + sigma2 [o==o', i==j],
 * (p ns) x (p ns) column-major, by-outputs ordering.  Same numbers as the reference's Xt_A_X(cholesky(latent_cov), H_full')
 * + sigma2 I without its Cholesky of a 1e-18-jittered covariance.  Partial sums over the shard as in lmm_oilmm_mean_and_var. */
int lmm_lmm_mean_and_cov(const lmm_post_t* post, const lmm_gp_t* gps, const double* U, const double* S, int p, int m,
                         int latent_begin, int latent_end, double sigma2, int add_noise,
                         const double* xs, int d, int ns, const lmm_jitters_t* jit,
                         double* mean_out, double* cov_out);

/* cov(f::IndependentMOGP, x, y) -- the two-input cross-covariance: reference src/independent_mogp.jl:66-71 (x, y both
 * MOInputIsotopicByOutputs: Matrix(BlockDiagonal(cov(f_l, x.x, y.x)))) and :184-215 (either input MOInputIsotopicByFeatures: the same
 * blocks with rows / columns permuted by indices_which_reorder_outputs_to_features); tested by the reference at
 * test/independent_mogp.jl:136-141.  post == NULL: prior latents `gps` (block l = kernelmatrix(k_l, x, y)); post != NULL (a handle
 * of lmm_mogp_posterior_create / lmm_oilmm_posterior_create / lmm_post_condition): PosteriorGP latents, block l =
 * K_l(x, y) - A_x' A_y with A_z = C_l.U' \ K_l(x_train, z).  x is d x n, y is d x n2.  cov_out: (m n) x (m n2) column-major; entry
 * ((l, i), (l', j)) sits at row  l n + i  (x by outputs)  or  i m + l  (x_by_features != 0), column  l' n2 + j  or  j m + l'.  Only the
 * blocks of the shard [latent_begin, latent_end) are filled, everything else is zero (shards sum to the whole).  (m n)(m n2) <= 4e8. */
int lmm_mogp_cross_cov(const lmm_post_t* post, const lmm_gp_t* gps, int m, int latent_begin, int latent_end,
                       const double* x, int d, int n, int x_by_features, const double* y, int n2, int y_by_features,
                       double* cov_out);

/* logpdf(po(xs, sigma2), ys) where po is the posterior OILMM (reference test/oilmm.jl:25; the posterior
 * is again an OILMM with the same H, reference src/oilmm.jl:133): per-latent posterior covariance at
 * xs (Schur complement) + the reference src/oilmm.jl:79-93 algorithm. */
int lmm_oilmm_post_logpdf(const lmm_post_t* post, const double* U, const double* S, int p, int m,
                          double sigma2, const double* xs, int d, int ns, const double* ys,
                          int with_regulariser, double* out);

/* ---- rand ----------------------------------------------------------------------------------- */
/* rand(rng, fx::FiniteGP{<:OILMM}): reference src/oilmm.jl:40-54.  The caller supplies the standard
 * normals in the reference's draw order: z_lat = m blocks of ns (latent order), eps = ns*p (by-outputs),
 * so a Julia shim passing randn(rng, ...) reproduces the reference sample-for-sample.
 * H = U sqrt(S), or dense H in U with S == NULL and latent jitter `ilmm_rand_jitter`
 * (rand(rng, fx::FiniteGP{<:ILMM}): reference src/ilmm.jl:78-87 + src/independent_mogp.jl:83-86).
 * post == NULL => prior latents.  Output: PARTIAL sum over the shard's latents (length ns*p); the
 * noise term sqrt(sigma2)*eps is added iff add_noise != 0. */
int lmm_lmm_rand(const lmm_post_t* post, const lmm_gp_t* gps,
                 const double* U, const double* S, int p, int m,
                 int latent_begin, int latent_end, double sigma2, int add_noise,
                 const double* xs, int d, int ns, const double* z_lat, const double* eps,
                 const lmm_jitters_t* jit, double* out);

/* rand(rng, fx, N): N samples from ONE factorisation per latent (the reference repeats the whole call N times,
 * src/ilmm.jl:90-92, src/independent_mogp.jl:92-96).  z_lat: [sample][m][ns], eps / out: [sample][p][ns]; per sample the
 * draw order is the reference's (m blocks of ns latent normals, then ns*p noise normals). */
int lmm_lmm_rand_multi(const lmm_post_t* post, const lmm_gp_t* gps,
                       const double* U, const double* S, int p, int m,
                       int latent_begin, int latent_end, double sigma2, int add_noise,
                       const double* xs, int d, int ns, int nsamples, const double* z_lat, const double* eps,
                       const lmm_jitters_t* jit, double* out);

/* ---- building blocks exported for tests / profiling (device pointers only) ------------------ */
/* In-place lower Cholesky of the leading ncols columns of an nrows x ncols column-major matrix (ld),
 * rows >= ncols ride along (become A21 * L11^-T).  nrows, ncols multiples of 64.  Winv: ncols/64 dense
 * 64x64 inverse diagonal blocks (scratch / output).  info: device int (0 = ok, k = pivot k failed). */
int lmm_dev_potrf(double* A, int nrows, int ncols, int ld, double* Winv, int n_real, int* info_dev);
/* Host-only (no GPU needed): the status the library derives from `count` pivot-info words of a batch -- LMM_OK; LMM_ERR_NOT_PD for
 * the first non-zero word (lmm_last_error_detail: latent_begin + index, pivot; reference: `cholesky` throwing PosDefException inside
 * src/oilmm.jl:90 / AbstractGPs logpdf); LMM_ERR_HIP when ANY word carries -7777, the marker potrf_region_kernel leaves when one of
 * its bounded dependency waits timed out (never expected; results are then undefined and must not be read as a PosDefException). */
int lmm_dev_check_info(const int* info, int count, int latent_begin);
/* Test hook of the allocation-extent guard (lmm_api.hip guard_extent: every Gram / triangular-solve / Schur-complement / factorisation
 * launch site checks the rows x cols (ld) block it touches against the pooled allocation the pointer lies in and returns LMM_ERR_ARG
 * instead of launching): applies it to a fresh pooled block of alloc_bytes, a block of Float64 elements. */
int lmm_dev_extent_check(size_t alloc_bytes, size_t rows, size_t ld, size_t cols);
/* Test hook of the region kernel's row-task plan (host arithmetic only, no device needed): for a block column of P 128-column panels of
 * nb matrices with rows_below rows under the square (rows_real of them holding data) on a device of `cus` CUs, with `assistants`
 * assistant workgroups available per matrix: out[0] = row tiles that stay 128 rows high, out[1] = row tasks per matrix (128-row tiles
 * + 64-row tiles), out[2] = assistants used.  The tiles cover the rows: 128 out[0] + 64 (out[1] - out[0]) >= rows_below. */
int lmm_dev_region_plan(int P, int nb, int rows_below, int rows_real, int cus, int assistants, int out[3]);
/* Test hook of the dataflow kernels' dependency flags: they carry a 26-bit launch epoch and are never reset; when the epoch wraps,
 * every persistent flag word is cleared.  *old_epoch (may be NULL) = the current epoch; set_to >= 0 replaces it (-1: read only). */
int lmm_dev_flag_epoch(int set_to, int* old_epoch);
/* Test hook of the strict-progress region kernel: on != 0 makes every workgroup ask for the task index a REVERSED dispatch order would
 * give it (the stand-in for a device that does not start workgroups in index order): small launches then run with all roles on the
 * "wrong" workgroups, launches larger than the device go through the 200-us time-out and take the next free index.  Values must not
 * change (tests/test_gpu_r5.py). */
int lmm_dev_claim_scramble(int on);
/* C[MxN] -= A[MxK] * B[NxK]^T (column-major, device). lower != 0: only tiles on/below the diagonal. */
int lmm_dev_gemm_nt_sub(double* C, int ldc, const double* A, int lda, const double* B, int ldb,
                        int M, int N, int K, int lower);
/* Gram assembly of one latent into a padded factor matrix (lower triangle + pad identity). */
int lmm_dev_gram(double* A, int ld, int nrows, int ncols, const double* x, int d, int n,
                 const lmm_gp_t* gp, double diag_add);
/* Standard normals generated on the device (Philox4x32-10 counter RNG + Box-Muller, Float64): out[j] for j < count is a
 * function of (seed, stream, j) only.  Optional companion of lmm_lmm_rand*: the reference draws its normals on the host with
 * the caller's rng (src/oilmm.jl:47,53), and the shim keeps doing so when the reference's random stream matters. */
int lmm_normals(unsigned long long seed, unsigned long long stream, size_t count, double* out);
/* ---- measurement hooks (bench.py roofline leg) ------------------------------------------------ */
/* Between lmm_profile_begin and lmm_profile_end every launch of the classes below is bracketed by HIP events
 * on the stream it is launched on.  serial != 0 forces all latents onto ONE stream, so an event pair times its
 * kernel alone (the production path runs latents on concurrent streams, where durations overlap).
 * work = algorithmic flops (MFMA classes) or algorithmic HBM bytes (Gram assembly) summed over launches. */
typedef enum {
  LMM_PROF_GRAM = 0,          /* gram_batch_kernel: lower-triangular f64 write, bytes                                          */
  LMM_PROF_UPDATE = 1,        /* potrf_node_kernel<2, .> (round 3: SYRK/GEMM trailing update with K >= 1024 + the next panel's leaf128 in
                                 one launch -- at K = 1024, 2048 also that panel's bulk rows --, + gemm16h_kernel<true> for a ragged
                                 last 64 rows; LMM_PANEL128=0 / fp32: every wide
                                 update: gemm16p_kernel / gemm16h_kernel / gemm32_kernel), flops                                */
  LMM_PROF_UPDATE_NARROW = 2, /* gemm44_kernel<64,false>: 64-column update (round-2 path; trailing 64 columns), flops           */
  LMM_PROF_TRSM = 3,          /* potrf_node_kernel<1> in bulk mode (the panels whose bulk rows do not ride in an update launch): panel
                                 rows x 128 x 128 inverse (round-2 path:
                                 gemm44_kernel<64,true>, TRSM by the 64 x 64 inverse block), flops                              */
  LMM_PROF_DIAG = 4,          /* leaf128_kernel (first panel) / diag64m_kernel: diagonal-block factor + inverse, flops          */
  LMM_PROF_REGION = 5,        /* potrf_region_kernel: a block column of <= 8 panels (leaves, bulk products, inner updates) in one
                                 dataflow launch, flops                                                                         */
  LMM_PROF_UPDATE_SHORT = 6,  /* potrf_node_kernel<1>: the same fused update + leaf for K < 1024 (latency- and epilogue-bound levels)  */
  LMM_PROF_SOLVE = 7,         /* gemm16p_kernel inside the triangular solves R <- R L^-T against a stored factor (cross-Gram rows of the
                                 predictive paths, L^-T of the gradient paths): the K = 64 .. n/2 block updates, flops 2 rows cols K;
                                 the dominant class of BASELINE configs[3] (n* n^2 of its n^3/3 + n* n^2 flops per latent)            */
  LMM_PROF_SOLVE_LEAF = 8,    /* gemm44_kernel<64,true>: the 64-column solves by the stored inverse diagonal blocks, flops rows 64^2    */
  LMM_PROF_STRIP = 9,         /* strip_reduce_kernel + strip_finish_kernel: posterior marginals from one read of R (HBM read), bytes   */
  LMM_PROF_COUNT = 10
} lmm_prof_class;
typedef struct { long long launches; double ms; double work; double bytes; /* algorithmic HBM bytes */ } lmm_prof_entry_t;
int lmm_profile_begin(int serial);
int lmm_profile_end(lmm_prof_entry_t* out /* LMM_PROF_COUNT entries */);

/* Write-only yardstick next to the Gram assembly's HBM roofline (bench.py roofline_gram.achievable_write_gbs): GB/s of `reps`
 * hipMemsetAsync fills of a pooled device block of `bytes` (>= 1 MiB), timed with HIP events on the library's main stream. */
int lmm_dev_write_rate(size_t bytes, int reps, double* gbs);

/* f64 MFMA issue-rate microbenchmark: measured TFLOP/s of v_mfma_f64_16x16x4_f64 in the form the update kernels issue it (16
 * accumulator blocks in architectural VGPRs, 4 + 4 operand fragments per k-step; tools/mfma_probe4: 77.7 = 98.9 % of the 78.6
 * datasheet peak -- the same instruction with AccVGPR accumulators issues at 36). */
int lmm_dev_mfma_f64_peak(double* tflops);

#ifdef __cplusplus
}
#endif
#endif /* LMM_HIP_H */
