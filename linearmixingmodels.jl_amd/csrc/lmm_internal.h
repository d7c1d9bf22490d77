// lmm_internal.h -- shared between lmm_kernels.hip (device) and lmm_api.hip (host orchestration).
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/lmm_hip.h"

extern int g_f32;              // compute dtype of the matrices: 0 = Float64, 1 = Float32 (lmm_kernels.hip)
#define LMM_MAX_BATCH 32
struct BatchPtr { double* p[LMM_MAX_BATCH]; };   // base pointers of the matrices of one batch (kernel argument, by value)
struct BatchInfo { int* p[LMM_MAX_BATCH]; };

struct LatentDev {
  int kind;
  double var, inv_ls, mean;
};

// Gram / factor-matrix assembly arguments (see gram_kernel).
struct GramArgs {
  double* A;            // factor matrix base (row 0 = matrix row `row_shift`)
  int ld, nrows, ncols; // nrows = rows covered by the launch (multiple of 64), ncols multiple of 64
  int row_tile0;        // first 64-row tile of the launch (rider-only launches start at ncols/64)
  int row_shift;        // row index stored at A[0]
  int full;             // 1: no lower-triangle skip (rectangular rider matrix)
  const double* x; int d, n;
  int kind; double var, inv_ls, diag_add, pad_diag;
  const double* diag_vec;                      // optional per-point diagonal term (length n), added to diag_add
  const double* rider; int rider_ld, nrider;   // rows ncols + r  <- rider[r*rider_ld + j] - rider_sub
  double rider_sub;                            // constant subtracted from the rider rows (latent mean: delta = T y - mean)
  const double* xs; int ns;                    // rows ncols + r  <- kappa(xs_r, x_j)
  int* info_zero;                              // optional: the matrix's pivot-info word, zeroed by the launch (saves a memset per call)
  int cpw;                                     // column tiles per workgroup (set by the launcher: 4, or 1 when the grid would be small)
};

// The same assembly for up to LMM_MAX_BATCH same-shaped matrices in ONE launch (blockIdx.z = matrix): everything in `base`
// is shared; the fields below replace base's per matrix.  All matrices of a launch have base.kind.
struct GramBatchArgs {
  GramArgs base;
  double* A[LMM_MAX_BATCH];
  double var[LMM_MAX_BATCH], inv_ls[LMM_MAX_BATCH], diag_add[LMM_MAX_BATCH];
  const double* diag_vec[LMM_MAX_BATCH];
  const double* rider[LMM_MAX_BATCH];
  double rider_sub[LMM_MAX_BATCH];
  int* info_zero[LMM_MAX_BATCH];
};

struct DenseArgs {
  double* A; int ld, nrows, ncols;
  const double* x; int d, n, m;
  const LatentDev* lat;      // device, m entries
  const double* sigmaT;      // device, m x m column-major (nbatch of them when sig_idx != nullptr)
  const int* sig_idx;        // optional, device, n entries: which sigmaT the point's noise block uses (sequential conditioning)
  const double* rider; int rider_ld, nrider;
};

void launch_gram(const GramArgs& a, hipStream_t st);
// nb same-shaped assemblies (differing only in A, kind, var, inv_ls, diag_add, diag_vec, rider): one launch per run of equal kinds
void launch_gram_batch(const GramArgs* args, int nb, hipStream_t st);
void launch_dense_assemble(const DenseArgs& a, hipStream_t st);
void launch_dense_cov(const double* S, int lds, int ns, int m, const double* Hm, int p, double jitter, double sigma2, double* T,
                      double* out, hipStream_t st);
void launch_dense_cross(double* R, int ldr, int nrows, int ncols, const double* xs, int ns, const double* x, int n, int d,
                        int m, const LatentDev* lat, hipStream_t st);
size_t dense_var_partial_elems(int ns, int p, int Ncols);
void launch_dense_var(const double* R, int ldr, int ns, int m, int Ncols, const double* Hm, int p, const LatentDev* lat,
                      double jitter, double sigma2, double* partial, double* out, hipStream_t st);
// Arguments of potrf_node_kernel (lmm_kernels.hip K2c): the trailing update of the columns [j0 + h, j0 + h + N) with the factored
// columns [j0, j0 + h), fused with the factorisation of the next 128-column panel's diagonal block; or that panel's bulk rows.
struct NodeArgs {
  BatchPtr A, W, W2;      // factor matrices; 64 x 64 inverse blocks; 128 x 128 inverse panels (scratch)
  BatchInfo info;
  int ld, M, j0, h, N, n_real;   // M: rows of the region (from row j0 + h) this launch covers
  int MT, nb;             // 128-row tiles of the region; matrices in the batch
  int rest_items, full_items, splitk;     // work items of the column tiles 1.. (gemm_work_item's enumeration and split-K tail)
  int full_items_last, splitk_last;       // the same for the LAST matrix of the batch, which carries the launch's tail
  int mode;               // NODE_UPDATE | NODE_LEAF [| NODE_FUSE], or NODE_BULK
  // NODE_FUSE: the bulk rows of the panel this launch's leaf factors run as the LAST work items of the same launch, behind
  // device-side flags (nflags: per matrix [0] abort word, [1] leaf done, [2 + ti] column-0 tile ti updated; values epoch * 32 + 1)
  int* nflags; int nf_stride, epoch;
  int strip_n, strip0;    // column tiles of the ragged last 64 rows run as work items of this launch (0: none / separate launch); their first item
  int Mb, MTb, bulk0;     // rows / 128-row tiles the bulk items cover (a ragged last 64 rows included); index of the first bulk item
};
// Arguments of potrf_region_kernel (lmm_kernels.hip K2d): the columns [c0, c0 + 128 P) of every matrix of the batch, rows c0 .. c0 + M - 1.
struct RegionArgs {
  BatchPtr A, W, W2;
  BatchInfo info, flags;  // flags: P * R readiness words + 1 abort word per matrix (zeroed once per factorisation)
  int ld, M, c0, P, R, n_real, nb, epoch, first_done;
  int M_real;             // rows c0 .. c0 + M_real - 1 hold data, the rest of the M rows is zero padding (rider rows are padded to 64)
  BatchPtr S;             // optional scratch (LMM_REGION_ASST_TILES 64 x 64 tiles per matrix): partial products of the ASSISTANT tasks
  int na;                 // assistant tasks per matrix (square rows LMM_REGION_ASST_MIN_R .. 2P - 1), 0: none
  int ntasks;             // workgroups per matrix (trace layout)
  int n128;               // the first n128 row tiles below the square are 128 rows high, the following ones 64
  long long* trace;       // optional (LMM_REGION_TRACE=1, tools/region_trace.py): start / end wall-clock ticks of every workgroup
  unsigned* claim; unsigned* claim_next;      // strict-progress build: this launch's claim counters, and the set it zeroes for the next one
  int claim_scramble;     // test hook (lmm_dev_claim_scramble): workgroups ask for the indices in REVERSE order, as if dispatched last-first
};
// Strict forward progress (lmm_set_strict_progress, default on): potrf_region_kernel's workgroups take their task INDEX in turn from a
// per-matrix counter (region_claim) instead of reading it from blockIdx.x, and the fused update launches (NODE_FUSE: bulk items that wait for earlier items of the same launch)
// are not used -- no kernel then relies on the order in which workgroups are dispatched.
extern int g_strict_progress;
extern int g_claim_scramble;              // test hook: see RegionArgs.claim_scramble
void strict_ticket_reset();               // drop the claim counters (after an error drained the device; at shutdown)
#define LMM_REGION_MAX_PANELS 8
#define LMM_REGION_ASST_MIN_C 4           // a helper's product for column block c >= this is split with its row's assistant
#define LMM_REGION_ASST_MIN_R (LMM_REGION_ASST_MIN_C + 2)
// the assistant of a row takes the blocks [0, LMM_REGION_ASST_SPLIT(c)) of the helper's K-long product for column block c.  Round 3:
// c / 2; round 4: 3 c / 4 -- once the chain's hand-offs got cheaper (sc1 loads instead of acquire fences) the helpers of rows >= 9 were
// again the slower side of the walker <-> helper cycle (walker waits of 3-10 us at n = 1024), and what a helper does per column AFTER
// W_c arrives cannot shrink, so the part before it must: the blocks [0, 3 c / 4) are final c / 4 - 1 block periods before they are needed.
#define LMM_REGION_ASST_SPLIT(c) ((3 * (c)) / 4)
#define LMM_REGION_ASST_TILES ((2 * LMM_REGION_MAX_PANELS - LMM_REGION_ASST_MIN_R) * 16)
// Bound of every dependency spin of the dataflow kernels, in ticks of the 100 MHz wall clock (4 s).  The deadlock argument (a workgroup
// only waits for workgroups dispatched before it, the walker excepted) covers one launch on an otherwise free device; kernels of other
// streams holding CUs or a serialising profiler can delay the one later-dispatched workgroup a walker waits for, hence seconds, not ms.
#define LMM_REGION_SPIN_TICKS 400000000LL
#define LMM_INFO_SYNC_TIMEOUT (-7777)     // pivot-info value a region launch leaves when a dependency wait timed out (never expected)
void region_flags_register(int* base, size_t ints);     // the context's persistent flag array (cleared when the launch epoch wraps)
extern int g_concurrent_batches;          // batches in flight on the slot streams (set by lmm_api.hip's fork_slots / join_slots)
void region_plan_probe(int P, int nb, int Mb, int Mb_real, int cus, int na_full, int out[3]);   // lmm_dev_region_plan
int region_flag_epoch(int set_to);                       // lmm_dev_flag_epoch: returns the current launch epoch; set_to >= 0 replaces it
size_t region_flag_ints(int NR);          // ints per matrix that the flags of any region of a matrix with NR rows need
void launch_region(const BatchPtr& A, const BatchPtr& W, const BatchPtr& W2, const BatchInfo& info, const BatchInfo& flags, int ld, int NR,
                   int c0, int width, int n_real, int nb, bool first_done, hipStream_t st, int rows_real = -1, const BatchPtr* S = nullptr);
// plain trailing update (no leaf) through the node kernel: C -= A B' for the region at j0 + h
void launch_leaf128(const BatchPtr& A, size_t offD, int ld, const BatchPtr& W, size_t offW, const BatchPtr& W2, size_t offW2,
                    int gcol0, int n_real, const BatchInfo& info, int nb, hipStream_t st);
// bulk rows of the panel at column r0 (its diagonal block factored, its inverse in W2): X = P Dinv' in place
void launch_panel_bulk(const BatchPtr& A, const BatchPtr& W2, int ld, int NR, int r0, int nb, hipStream_t st);
// C -= A B' for the region at r0 = j0 + h (N columns, multiple of 128) + leaf128 on its top-left block
// nflags != nullptr: also the bulk rows of that panel, in the same launch (NODE_FUSE); returns true when it did
bool launch_update_leaf(const BatchPtr& A, const BatchPtr& W, const BatchPtr& W2, const BatchInfo& info, int ld, int NR, int j0, int h,
                        int N, int n_real, int nb, hipStream_t st, int* nflags = nullptr, int nf_stride = 0);
size_t node_flag_ints(int NR);            // ints per matrix of the NODE_FUSE flags of a matrix with NR rows
// lmm_kernels_f32w.hip: fp32 C -= A B' on 256 x 256 tiles (one workgroup per CU); false: not launched (shape / switch), use the 128-tile kernel
bool launch_gemm32w(const BatchPtr& C, size_t offC, int ldc, const BatchPtr& A, size_t offA, int lda, const BatchPtr& B, size_t offB, int ldb,
                    int M, int N, int K, int lower, int nb, int cus, bool deterministic, hipStream_t st);
void launch_diag64(const BatchPtr& A, size_t offA, int ld, const BatchPtr& W, size_t offW, int gcol0, int n_real,
                   const BatchInfo& info, int nb, hipStream_t st);
void launch_gemm_nt(const BatchPtr& C, size_t offC, int ldc, const BatchPtr& A, size_t offA, int lda, const BatchPtr& B,
                    size_t offB, int ldb, int M, int N, int K, int lower, bool set, int nb, hipStream_t st);
void launch_gemm_nt(double* C, int ldc, const double* A, int lda, const double* B, int ldb, int M, int N, int K,
                    int lower, bool set, hipStream_t st);
void launch_lml_reduce(const double* A, int ld, int n, int rider_row0, int nrhs, double* out, hipStream_t st);
// out[b*nrhs + r]; info_out != nullptr: also info_out[b] = *info.p[b] (out / info_out may be device-mapped pinned host memory: the
// results then need no copy back)
void launch_lml_reduce(const BatchPtr& A, int nb, int ld, int n, int rider_row0, int nrhs, double* out, hipStream_t st,
                       const BatchInfo* info = nullptr, int* info_out = nullptr);
void launch_extract_row(const double* A, int ld, int row, int n, double* out, hipStream_t st);
void launch_extract_rows(const BatchPtr& A, int nb, int ld, int row, int n, int nfill, const BatchPtr& o1, const BatchPtr& o2,
                         hipStream_t st);
int strip_kc(int nk);
size_t strip_partial_elems(int nr, int nk, int nv);
void launch_rider_stats(const double* R, int ld, int nr, int nk, const double* z, double mu, double base, double* partial,
                        double* mean_out, double* var_out, hipStream_t st);
void launch_backsolve(const BatchPtr& L, int ld, const BatchPtr& W, int nblk, const BatchPtr& z, int nb, hipStream_t st);
void launch_tall_skinny(const double* In, int ldi, int n, int K, const double* Mx, int ldm, int C, double* Out, int ldo,
                        const double* sub, const double* Ref, int ldr, double* partial, int mode, hipStream_t st);
int tall_skinny_partials(int n, int C);
void launch_sum_partials(const double* partial, int count, double* out, hipStream_t st);
size_t post_mean_partial_elems(int ns, int n);
void launch_post_mean(const double* xs, int ns, const double* x, int n, int d, const double* alpha, LatentDev g,
                      double* partial, double* out, hipStream_t st);
void launch_mix(const double* lat, int ns, int ml, const double* Hm, int p, int pw, double lat_add, double out_add,
                const double* eps, double eps_scale, double* out, hipStream_t st);
void launch_mix_bf16(const double* lat, int ns, int ml, const double* Hm, int p, int pw, double lat_add, double out_add,
                     int terms, double* out, hipStream_t st);
void launch_cov_mix(const BatchPtr& Cl, int ldcl, int nl, const double* Hs, int p, int ns, double jitter, double sigma2,
                    int init, double* out, hipStream_t st);
void launch_trmv_lower(const double* L, int ld, int n, const double* z, double mu, double* partial, double* out,
                       hipStream_t st);
void launch_syrk_upper_set(double* C, int ldc, const double* X, int ldx, int N, hipStream_t st);
void launch_syrk_upper_set(const BatchPtr& C, int ldc, const BatchPtr& X, int ldx, int N, int nb, hipStream_t st);
void launch_set_identity(double* R, int ld, int nc, hipStream_t st);
int grad_partials(int n);
#define LMM_NGRAD 8
void launch_grad_reduce(const double* Kinv, int ld, int n, int nsplit, const double* alpha, const double* delta, const double* x, int d,
                        LatentDev g, double* partial, double* out7, hipStream_t st);
void launch_vec_axpby(const double* a, double sa, const double* b, double sb, size_t n, double* out, hipStream_t st);
void launch_block_trace(const double* Minv, int ld, int n, int m, int i0, int i1, double* out, hipStream_t st);   // points i0..i1-1
// Consecutive point ranges [off[b], off[b + 1]) that carry the observation-noise variance s2[b]: the conditioning batches of a
// sequentially conditioned posterior followed by the test points (gradient of the predictive logpdf).
#define LMM_MAX_NOISE_BLOCKS 8
struct NoiseBlocks {
  int nblk;
  int off[LMM_MAX_NOISE_BLOCKS + 1];
  double s2[LMM_MAX_NOISE_BLOCKS];
  int count(int b) const { return off[b + 1] - off[b]; }
};
// out[k] = a[k] + (num / s2[block of row k]) * b[k]   with row(k) = k mod N  (column-major N x p operands)
void launch_vec_lin_blocks(const double* a, const double* b, const NoiseBlocks& nb, double num, int N, size_t count, double* out, hipStream_t st);
void launch_atb(const double* X, int ldx, const double* Z, int ldz, int n, int na, int nb, double* out, hipStream_t st);
void launch_fill(double* p, int n, double v, hipStream_t st);
void launch_reorder(const double* in, int n, int p, int to_outputs, double* out, hipStream_t st);
void launch_vec_lin(const double* a, const double* b, double sb, int n, double* out, hipStream_t st);  // out = a + sb*b
// out[(row0 + i rs) + (col0 + j cs) ldo] = src[i + j lds] (src a MATRIX in the compute dtype, out Float64): a block of cov(f, x, y)
void launch_block_scatter(const double* src, int lds, int nr, int nc, double* out, size_t ldo, size_t row0, int rs, size_t col0, int cs,
                          hipStream_t st);
void launch_normals(unsigned long long seed, unsigned long long stream, size_t count, double* out, hipStream_t st);
void launch_mfma_peak(double* out, int blocks, int iters, hipStream_t st);
