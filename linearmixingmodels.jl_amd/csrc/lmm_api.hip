// lmm_api.hip -- host orchestration + C ABI (include/lmm_hip.h) of liblmm_hip.so.
//
// One process drives ONE MI355X (one process per GPU; the multi-GPU layer above shards latents and
// sums partial results with one RCCL all-reduce).  Latent problems are independent, so each latent of the
// shard gets a slot = (factor-matrix buffer, HIP stream); slots run concurrently so one latent's
// latency-bound 64x64 diagonal-block step overlaps the other latents' MFMA trailing updates.
//
// Blocked Cholesky: recursive halving on column ranges,
//     potrf(j0, w):  potrf(j0, h);  C[j0+h:, j0+h:j0+w] -= A[j0+h:, j0:j0+h] A[j0+h:j0+w, j0:j0+h]';  potrf(j0+h, w-h)
// so that >95 % of the n^3/3 flops run in large-K f64-MFMA updates that read/write each trailing tile once
// per level (log2(n/64) levels) instead of once per panel.  Leaves (64 columns): diag64 (factor + inverse
// of the diagonal block) then TRSM as a GEMM with the inverse.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "lmm_internal.h"

namespace {

constexpr double kLog2Pi = 1.8378770664093453;
constexpr int kMaxStreams = 16;

struct Ctx {
  bool init = false;
  int device = -1;
  int nstreams = 4;
  hipStream_t streams[kMaxStreams];
  hipEvent_t ev_main;
  hipEvent_t ev_slot[kMaxStreams];
  // pinned host arena for the small per-call uploads / read-backs (projection matrices, per-latent results): copies from / to
  // pinned memory are truly asynchronous, pageable ones stall the calling thread until the stream has drained
  char* pin = nullptr;
  size_t pin_cap = 0, pin_off = 0;
  char* pin_dev = nullptr;           // the arena as the device addresses it (hipHostGetDevicePointer); nullptr: not mapped
  std::vector<void*> scratch;          // device blocks that live until the NEXT API call starts (call_scratch)
  int* region_flags = nullptr;         // dependency flags of potrf_region_kernel: [stream][matrix][region_flag_ints], zeroed ONCE (lmm_init)
                                       // -- every launch tags its flags with a fresh epoch, so they never need resetting
  std::multimap<size_t, void*> pool;   // cached device blocks (size -> ptr)
  std::map<void*, size_t> live;
  std::string err;
  int err_latent = -1, err_info = 0;
  // multi-GPU: one RCCL communicator per process (rank = this GPU)
  ncclComm_t comm = nullptr;
  int comm_rank = 0, comm_world = 0;
  hipEvent_t ev_caller = nullptr;
  // measurement hooks
  bool prof = false, prof_serial = false;
  struct ProfRec { int cls; double work, bytes; hipEvent_t e0, e1; int M, N, K, count; };
  std::vector<ProfRec> prof_recs;
  std::vector<hipEvent_t> ev_pool;
};
Ctx g;
std::mutex g_mu;
int g_proj = 0;   // lmm_proj_dtype of the H unprojection of predictive marginals (lmm_set_projection_dtype)

// H unprojection of latent marginals (reference src/oilmm.jl:69,72) in the selected projection dtype
void mix_marginals(const double* lat, int ns, int ml, const double* Hm, int p, int pw, double lat_add, double out_add, double* out,
                   hipStream_t st) {
  if (g_proj == LMM_PROJ_NATIVE || ml == 0) launch_mix(lat, ns, ml, Hm, p, pw, lat_add, out_add, nullptr, 0.0, out, st);
  else launch_mix_bf16(lat, ns, ml, Hm, p, pw, lat_add, out_add, g_proj == LMM_PROJ_BF16X2 ? 2 : 1, out, st);
}

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g.err = buf;
  return code;
}

#define HIPCHK(expr)                                                                              \
  do {                                                                                            \
    hipError_t e_ = (expr);                                                                       \
    if (e_ != hipSuccess) throw fail(LMM_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                                     __FILE__, __LINE__);                                         \
  } while (0)

// bytes from the pinned arena (valid until the next API call), or nullptr when it is full / absent
void* pin_take(size_t bytes) {
  bytes = (bytes + 63) & ~size_t(63);
  if (g.pin == nullptr || g.pin_off + bytes > g.pin_cap) return nullptr;
  void* p = g.pin + g.pin_off;
  g.pin_off += bytes;
  return p;
}

// device-side alias of a pointer into the pinned arena (kernels write small results straight into host memory: no copy back)
template <typename T>
T* pin_dev(T* host) {
  if (host == nullptr || g.pin_dev == nullptr) return nullptr;
  return reinterpret_cast<T*>(g.pin_dev + (reinterpret_cast<char*>(host) - g.pin));
}

void* dev_alloc(size_t bytes) {
  if (bytes == 0) bytes = 256;
  bytes = (bytes + 255) & ~size_t(255);
  auto it = g.pool.find(bytes);
  void* p = nullptr;
  if (it != g.pool.end()) {
    p = it->second;
    g.pool.erase(it);
  } else {
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) {   // release the cache and retry once
      for (auto& kv : g.pool) (void)hipFree(kv.second);
      g.pool.clear();
      (void)hipGetLastError();
      e = hipMalloc(&p, bytes);
      if (e != hipSuccess) throw fail(LMM_ERR_HIP, "hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
    }
  }
  g.live[p] = bytes;
  return p;
}

void dev_free(void* p) {
  if (!p) return;
  auto it = g.live.find(p);
  if (it == g.live.end()) return;
  g.pool.insert({it->second, p});
  g.live.erase(it);
}

// Scratch that kernels queued by this API call use: handed back to the pool when the NEXT call starts (every entry point returns
// with its streams drained, and error paths drain the device), so it is never recycled while in flight.
double* call_scratch(size_t count) {
  void* p = dev_alloc(count * sizeof(double));
  g.scratch.push_back(p);
  return static_cast<double*>(p);
}
void release_call_scratch() {
  for (void* p : g.scratch) dev_free(p);
  g.scratch.clear();
}

template <typename T>
struct Buf {   // RAII device buffer from the caching pool
  T* p = nullptr;
  size_t n = 0;
  bool own = true;       // false: p is borrowed (a device alias into the pinned arena), not returned to the pool
  Buf() = default;
  explicit Buf(size_t count) : p(static_cast<T*>(dev_alloc(count * sizeof(T)))), n(count) {}
  Buf(const Buf&) = delete;
  Buf& operator=(const Buf&) = delete;
  Buf(Buf&& o) noexcept : p(o.p), n(o.n), own(o.own) { o.p = nullptr; o.n = 0; }
  Buf& operator=(Buf&& o) noexcept { if (this != &o) { if (own) dev_free(p); p = o.p; n = o.n; own = o.own; o.p = nullptr; o.n = 0; } return *this; }
  ~Buf() { if (own) dev_free(p); }
};

bool is_device_ptr(const void* p) {
  hipPointerAttribute_t a;
  hipError_t e = hipPointerGetAttributes(&a, p);
  if (e != hipSuccess) { (void)hipGetLastError(); return false; }
  return a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged;
}

// A read-only input that may live on host or device: gives a device pointer valid on stream st.
struct DevIn {
  const double* p = nullptr;
  Buf<double> own;
  DevIn(const double* src, size_t count, hipStream_t st) {
    if (src == nullptr) return;
    if (is_device_ptr(src)) { p = src; return; }
    own = Buf<double>(count);
    HIPCHK(hipMemcpyAsync(own.p, src, count * sizeof(double), hipMemcpyHostToDevice, st));
    p = own.p;
  }
};

// An output that may live on host or device.
struct DevOut {
  double* dst; double* p; size_t count; Buf<double> own; bool direct;
  DevOut(double* dst_, size_t count_) : dst(dst_), p(nullptr), count(count_), direct(false) {
    if (dst == nullptr) return;
    if (is_device_ptr(dst)) { p = dst; direct = true; }
    else { own = Buf<double>(count); p = own.p; }
  }
  void finish(hipStream_t st) {
    if (dst && !direct) HIPCHK(hipMemcpyAsync(dst, p, count * sizeof(double), hipMemcpyDeviceToHost, st));
  }
};

inline int rup(int v, int m) { return (v + m - 1) / m * m; }

// fp32 compute mode (lmm_set_compute_dtype): MATRICES (factor matrices, inverse diagonal blocks, cross-solve blocks) are float
// buffers; they are still carried as Buf<double> / double* (opaque to the host, which never dereferences them), sized by
// mat_count(elements) doubles.  Vectors stay double.
inline size_t mat_count(size_t elems) { return g_f32 ? (elems + 1) / 2 : elems; }
inline double mat_bytes(double elems) { return elems * (g_f32 ? 4.0 : 8.0); }
// element `off` of a matrix buffer in the current storage type, as the double* the launch wrappers take
inline double* mat_at(double* p, size_t off) { return g_f32 ? reinterpret_cast<double*>(reinterpret_cast<float*>(p) + off) : p + off; }

// ---- allocation-extent guard -------------------------------------------------------------------------------------------------
// Every launch below that reads or writes a rows x cols block (leading dimension ld) of a POOLED buffer is preceded by this check of
// the block against the allocation the pointer lies in: a mismatch between two roundings of the same size (round 3: cross-solve
// blocks kept rup(n*, 64) rows while the Schur complement read rup(n*, 128) of them -- an out-of-bounds device read at n* = 9 that a
// later run hid) becomes LMM_ERR_ARG before anything is launched, instead of a fault.  Pointers outside the pool (caller memory)
// are not checked.  matrix: the block is in the compute dtype (Float32 elements in the fp32 mode), else Float64.
void guard_extent(const void* p, size_t rows, size_t ld, size_t cols, bool matrix, const char* what) {
  if (p == nullptr || rows == 0 || cols == 0) return;
  auto it = g.live.upper_bound(const_cast<void*>(p));
  if (it == g.live.begin()) return;
  --it;
  const char* b0 = static_cast<const char*>(it->first);
  const char* b1 = b0 + it->second;
  const char* q = static_cast<const char*>(p);
  if (q >= b1) return;                                   // not inside a pooled block
  const size_t eb = (matrix && g_f32) ? 4 : 8;
  const size_t need = ((cols - 1) * ld + rows) * eb;
  if (rows > ld || q + need > b1)
    throw fail(LMM_ERR_ARG, "internal extent check failed: %s touches %zu x %zu (ld %zu) = %zu bytes at offset %zu of a %zu-byte allocation",
               what, rows, cols, ld, need, (size_t)(q - b0), it->second);
}
// (a launch covers the matrix rows [64 row_tile0, nrows), stored from buffer row 64 row_tile0 - row_shift on: buffer rows < nrows - row_shift)
void guard_gram(const GramArgs& a, const char* what) {
  if (64 * a.row_tile0 < a.row_shift) throw fail(LMM_ERR_ARG, "internal extent check failed: %s starts above its buffer (row tile %d, shift %d)", what, a.row_tile0, a.row_shift);
  guard_extent(a.A, (size_t)(a.nrows - a.row_shift), (size_t)a.ld, (size_t)a.ncols, true, what);
}
void gram_g(const GramArgs& a, hipStream_t st, const char* what = "Gram assembly") { guard_gram(a, what); launch_gram(a, st); }
void gram_batch_g(const GramArgs* ga, int nb, hipStream_t st, const char* what = "Gram assembly") {
  for (int j = 0; j < nb; ++j) guard_gram(ga[j], what);
  launch_gram_batch(ga, nb, st);
}
// C (M x N, ldc) -= A (M x K, lda) B (N x K, ldb)'
void gemm_nt_g(double* C, int ldc, const double* A, int lda, const double* B, int ldb, int M, int N, int K, int lower, bool set,
               hipStream_t st, const char* what) {
  guard_extent(C, M, ldc, N, true, what); guard_extent(A, M, lda, K, true, what); guard_extent(B, N, ldb, K, true, what);
  launch_gemm_nt(C, ldc, A, lda, B, ldb, M, N, K, lower, set, st);
}
void gemm_nt_g(const BatchPtr& C, size_t offC, int ldc, const BatchPtr& A, size_t offA, int lda, const BatchPtr& B, size_t offB, int ldb,
               int M, int N, int K, int lower, bool set, int nb, hipStream_t st, const char* what) {
  const size_t eb = g_f32 ? 4 : 8;
  for (int j = 0; j < nb; ++j) {
    guard_extent(reinterpret_cast<const char*>(C.p[j]) + offC * eb, M, ldc, N, true, what);
    guard_extent(reinterpret_cast<const char*>(A.p[j]) + offA * eb, M, lda, K, true, what);
    guard_extent(reinterpret_cast<const char*>(B.p[j]) + offB * eb, N, ldb, K, true, what);
  }
  launch_gemm_nt(C, offC, ldc, A, offA, lda, B, offB, ldb, M, N, K, lower, set, nb, st);
}
void rider_stats_g(const double* R, int ld, int nr, int nk, const double* z, double mu, double base, double* partial,
                   double* mean_out, double* var_out, hipStream_t st) {
  if (R) guard_extent(R, nr, ld, nk, true, "rider statistics (R)");
  launch_rider_stats(R, ld, nr, nk, z, mu, base, partial, mean_out, var_out, st);
}

// Brackets one launch with events when profiling is on (lmm_profile_begin); otherwise just launches.
struct ProfScope {
  bool on; hipStream_t st; size_t idx;
  ProfScope(int cls, double work, hipStream_t st_, int M = 0, int N = 0, int K = 0, double bytes = 0.0, int count = 1)
      : on(g.prof), st(st_), idx(0) {      // count: launches bracketed by this one event pair
    if (!on) return;
    Ctx::ProfRec r; r.cls = cls; r.work = work; r.bytes = bytes; r.M = M; r.N = N; r.K = K; r.count = count;
    for (hipEvent_t* e : {&r.e0, &r.e1}) {
      if (!g.ev_pool.empty()) { *e = g.ev_pool.back(); g.ev_pool.pop_back(); }
      else HIPCHK(hipEventCreate(e));
    }
    HIPCHK(hipEventRecord(r.e0, st));
    idx = g.prof_recs.size();
    g.prof_recs.push_back(r);
  }
  ~ProfScope() { if (on) (void)hipEventRecord(g.prof_recs[idx].e1, st); }
};

inline int eff_streams() { return (g.prof && g.prof_serial) ? 1 : g.nstreams; }

struct Dims {
  int n, NC, NR, ld;
  Dims(int n_, int nrider) : n(n_) {
    NC = rup(std::max(n, 1), 128);      // whole 128-column panels (round 3: the panel / region kernels work on them; the pad is identity)
    NR = rup(NC + std::max(nrider, 0), 64);
    ld = NR;
    if ((ld % 512) == 0) ld += 16;   // keep column starts off the same HBM channel / L2 set
  }
  size_t elems() const { return (size_t)ld * NC; }
};

LatentDev to_dev(const lmm_gp_t& gp) {
  LatentDev d;
  d.kind = gp.kind; d.var = gp.variance; d.inv_ls = 1.0 / gp.lengthscale; d.mean = gp.mean;
  return d;
}

int check_gps(const lmm_gp_t* gps, int m) {
  if (!gps) return fail(LMM_ERR_ARG, "gps is NULL");
  for (int l = 0; l < m; ++l) {
    if (gps[l].kind < 0 || gps[l].kind > 2) return fail(LMM_ERR_UNSUPPORTED, "latent %d: unsupported kernel kind %d", l, gps[l].kind);
    if (!(gps[l].variance > 0.0) || !(gps[l].lengthscale > 0.0)) return fail(LMM_ERR_ARG, "latent %d: variance and lengthscale must be > 0", l);
  }
  return LMM_OK;
}

// ------------------------------------------------------------------------------------------------
// blocked factorisation drivers
// ------------------------------------------------------------------------------------------------
inline int split(int w) {            // left width of the recursive split (multiple of 64; of 128 when w >= 256)
  if (w >= 256) return rup(w / 2, 128);
  return (w == 192) ? 128 : 64;
}

// A batch of up to LMM_MAX_BATCH same-shaped factor matrices factored in lock-step by the same launches
// (blockIdx.y / blockIdx.x selects the matrix): the small recursion levels then fill the chip and the launch
// count per latent drops by the batch size.
struct Batch {
  int nb = 0;
  BatchPtr A{}, W{};
  BatchInfo info{};
  void add(double* a, double* w, int* i) { A.p[nb] = a; W.p[nb] = w; info.p[nb] = i; ++nb; }
};

// Factor columns [j0, j0+w) of every matrix of the batch (rows j0..NR-1 participate).  W: NC/64 inverse diagonal blocks.
void potrf_rec(const Batch& B, int ld, int NR, int j0, int w, int n_real, hipStream_t st) {
  const double nb = B.nb;
  if (w <= 64) {
    const size_t offW = (size_t)(j0 / 64) * 4096;
    {
      ProfScope ps(LMM_PROF_DIAG, nb * 2.0 * 64.0 * 64.0 * 64.0 / 3.0, st);
      launch_diag64(B.A, (size_t)j0 * ld + j0, ld, B.W, offW, j0, n_real, B.info, B.nb, st);
    }
    const int M = NR - (j0 + 64);
    if (M > 0) {
      const size_t offP = (size_t)j0 * ld + (j0 + 64);
      ProfScope ps(LMM_PROF_TRSM, nb * (double)M * 64.0 * 64.0, st);   // triangular solve: M * 64^2 flops
      launch_gemm_nt(B.A, offP, ld, B.A, offP, ld, B.W, offW, 64, M, 64, 64, 0, true, B.nb, st);
    }
    return;
  }
  const int h = split(w);
  potrf_rec(B, ld, NR, j0, h, n_real, st);
  const int r0 = j0 + h;
  {
    const double Mr = NR - r0, Nc = w - h;     // lower trapezoid: Nc(Nc+1)/2 + (Mr-Nc)Nc outputs, 2h flops each
    const double outs = Nc * (Nc + 1.0) / 2.0 + (Mr - Nc) * Nc;
    // algorithmic bytes: C read + written once (16 B per output), the A panel (Mr x h, which contains B) read once
    ProfScope ps(Nc <= 64 ? LMM_PROF_UPDATE_NARROW : LMM_PROF_UPDATE, nb * 2.0 * h * outs, st, NR - r0, w - h, h,
                 nb * (16.0 * outs + 8.0 * Mr * h));
    const size_t offA = (size_t)j0 * ld + r0;
    launch_gemm_nt(B.A, (size_t)r0 * ld + r0, ld, B.A, offA, ld, B.A, offA, ld, NR - r0, w - h, h, 1, false, B.nb, st);
  }
  potrf_rec(B, ld, NR, r0, w - h, n_real, st);
}

// Round 3: the same recursion with 128-column panels (lmm_kernels.hip K2c).  A panel = leaf128 (its 128 x 128 diagonal block: factor,
// 64 x 64 inverse blocks, full inverse into the W2 scratch) + ONE bulk GEMM (rows below: X = P Dinv'); and every trailing update
// also runs the leaf of the panel that follows it (launch_update_leaf), so that per 128 columns the stream sees two launches
// (update + leaf, bulk) instead of six.  first_done: the diagonal block of the first panel of [j0, j0 + w) is already factored.
// Widths that are not multiples of 128 (NC = 64 mod 128) end in the round-2 path for their last 64 columns.
static int g_slots_in_flight = 1;       // batches that run concurrently on the slot streams (set by fork_slots; read by potrf_batch's base-case rule)
static int g_region_whole = 1;          // LMM_REGION_ALL=1: also as the base case of the recursion for larger matrices (measured: no gain)
static int g_region_cols = -1;          // widest block column potrf_region_kernel takes in one launch (LMM_REGION=<columns>, up to 1024; default 0: off)
// bulk_done (implies first_done): the rows below that block are solved as well (the update launch that factored it ran them too).
struct NodeFlags { int* p = nullptr; int stride = 0; int min_k = 0, max_k = 1 << 30; int rows_real = -1; BatchPtr S{}; int region_cols = 0; };   // region_cols: widest block column that becomes ONE dataflow launch in this factorisation   // S: region assistants' scratch      // rows_real: rows that hold data (-1: all NR)
void potrf_rec_panel(const Batch& B, const BatchPtr& W2, const BatchInfo& flags, const NodeFlags& nfl, int ld, int NR, int j0, int w, int n_real,
                     hipStream_t st, bool first_done, bool bulk_done = false) {
  const double nb = B.nb;
  if (nfl.region_cols > 0 && w <= nfl.region_cols && w >= 128 && (w % 128) == 0 && flags.p[0] != nullptr) {
    // the whole block column as ONE dataflow launch (lmm_kernels.hip K2d): leaves, bulk products and all updates inside it
    const double Mr = (nfl.rows_real >= 0 ? std::min(NR, nfl.rows_real) : NR) - j0, Wd = w;
    const double fl = Mr * Wd * Wd - 2.0 * Wd * Wd * Wd / 3.0;           // flops of factoring an Mr x Wd tall panel: Mr Wd^2 - 2 Wd^3 / 3
    ProfScope ps(LMM_PROF_REGION, nb * fl, st, NR - j0, w, w);
    launch_region(B.A, B.W, W2, B.info, flags, ld, NR, j0, w, n_real, B.nb, first_done, st, nfl.rows_real, &nfl.S);
    return;
  }
  if (w == 128) {
    if (!first_done) {
      ProfScope ps(LMM_PROF_DIAG, nb * 2.0 * 128.0 * 128.0 * 128.0 / 3.0, st, 128, 128, 128);
      launch_leaf128(B.A, (size_t)j0 * ld + j0, ld, B.W, (size_t)(j0 / 64) * 4096, W2, (size_t)(j0 / 128) * 16384, j0, n_real, B.info, B.nb, st);
    }
    const int M = NR - (j0 + 128);
    if (M > 0 && !bulk_done) {
      const double Mreal = (nfl.rows_real >= 0 ? std::min(NR, nfl.rows_real) : NR) - (j0 + 128);
      ProfScope ps(LMM_PROF_TRSM, nb * Mreal * 128.0 * 128.0, st, M, 128, 128);       // triangular solve: (real rows) * 128^2 flops
      launch_panel_bulk(B.A, W2, ld, NR, j0, B.nb, st);
    }
    return;
  }
  if (w <= 64) { potrf_rec(B, ld, NR, j0, w, n_real, st); return; }          // a trailing 64-column leaf (never pre-factored)
  const int h = split(w);
  potrf_rec_panel(B, W2, flags, nfl, ld, NR, j0, h, n_real, st, first_done, bulk_done);
  const int r0 = j0 + h, Nc = w - h;
  // algorithmic rows: those that hold data (rows_real: the Gram rows + the real rider rows), not the 64-row padding of the riders
  const double Mr = (nfl.rows_real >= 0 ? std::min(NR, nfl.rows_real) : NR) - r0;
  const double outs = (double)Nc * (Nc + 1.0) / 2.0 + (Mr - Nc) * Nc;
  const size_t offA = (size_t)j0 * ld + r0;
  if (Nc >= 128) {
    // + the leaf's 2 * 128^3 / 3 flops, run by one workgroup of this launch
    bool fused;
    {
      // K >= 1024: potrf_node_kernel<2> (+ gemm16h_kernel for a ragged last 64 rows) -- the dominant kernel; below: potrf_node_kernel<1>.
      // With the bulk rows of the next panel in the same launch (nfl.p): + their Mb * 128^2 flops and 16 B per entry
      const double Mb = (nfl.p && h >= nfl.min_k && h <= nfl.max_k) ? std::max(0.0, Mr - 128.0) : 0;
      ProfScope ps(h >= 1024 ? LMM_PROF_UPDATE : LMM_PROF_UPDATE_SHORT,
                   nb * (2.0 * h * outs + 2.0 * 128.0 * 128.0 * 128.0 / 3.0 + Mb * 128.0 * 128.0), st, NR - r0, Nc, h,
                   nb * (16.0 * outs + 8.0 * Mr * h + 16.0 * Mb * 128.0));
      fused = launch_update_leaf(B.A, B.W, W2, B.info, ld, NR, j0, h, Nc, n_real, B.nb, st, (h >= nfl.min_k && h <= nfl.max_k) ? nfl.p : nullptr, nfl.stride);
    }
    potrf_rec_panel(B, W2, flags, nfl, ld, NR, r0, Nc, n_real, st, true, fused);
  } else {
    {
      ProfScope ps(LMM_PROF_UPDATE_NARROW, nb * 2.0 * h * outs, st, NR - r0, Nc, h, nb * (16.0 * outs + 8.0 * Mr * h));
      launch_gemm_nt(B.A, (size_t)r0 * ld + r0, ld, B.A, offA, ld, B.A, offA, ld, NR - r0, Nc, h, 1, false, B.nb, st);
    }
    potrf_rec(B, ld, NR, r0, Nc, n_real, st);
  }
}

// Entry point of the factorisation of a batch: columns [0, NC) of every matrix.  Float64 batches take the 128-column panel path
// (LMM_PANEL128=0: the round-2 path); its W2 scratch -- one 128 x 128 inverse per panel and matrix -- lives until the API call ends.
void potrf_batch(const Batch& B, int ld, int NR, int NC, int n_real, hipStream_t st, int rows_real = -1) {
  for (int j = 0; j < B.nb; ++j) {
    guard_extent(B.A.p[j], NR, ld, NC, true, "factorisation (factor matrix)");
    guard_extent(B.W.p[j], 64, 64, (size_t)(NC / 64) * 64, true, "factorisation (inverse diagonal blocks)");
  }
  static int panel128 = -1;
  if (panel128 < 0) { const char* e = getenv("LMM_PANEL128"); panel128 = e ? (atoi(e) != 0) : 1; }
  if (g_f32 || !panel128 || NC < 128 || (ld & 1)) { potrf_rec(B, ld, NR, 0, NC, n_real, st); return; }
  if (g_region_cols < 0) { const char* e = getenv("LMM_REGION"); g_region_cols = e ? atoi(e) : 1024; const char* ea = getenv("LMM_REGION_ALL"); g_region_whole = (ea && atoi(ea) != 0) ? 0 : 1; if (g_region_cols > 128 * LMM_REGION_MAX_PANELS) g_region_cols = 128 * LMM_REGION_MAX_PANELS; }
  const size_t per = (size_t)(NC / 128) * 16384;
  double* w2 = call_scratch(per * B.nb);
  BatchPtr W2{};
  for (int j = 0; j < B.nb; ++j) W2.p[j] = w2 + per * j;
  BatchInfo flags{};
  // Default: the region kernel serves matrices that are ONE region (NC <= 1024: the whole factorisation in one launch, the small-n
  // path); larger matrices take the panel recursion throughout (as their base case the region kernel measured no faster than the
  // panel launches: DESIGN.md).  LMM_REGION_ALL=1 enables it there too, LMM_REGION=0 disables it.
  // Mid sizes / few matrices (round 3, tools/mid_probe.py): while a block column's dataflow launch -- 2 P square tasks + one per
  // 128-row tile below, per matrix -- is at most ~2.3 workgroups per CU, it beats the panel launches it replaces (8 latents: n = 1536
  // 1.34 -> 0.96 ms, 2048 1.97 -> 1.51, 3072 3.50 -> 2.93, 4096 5.80 -> 5.2, 8192 27.1 -> 26.5; 16 x 2048: 2.19 -> 2.01; a rank's
  // 4-latent share of C2: 94.5 -> 93.9); with more work per launch it does not (16 x 4096: a tie; the two concurrent 16-latent batches
  // of C2 at N = 1: 696 -> 706 ms).  LMM_REGION_ALL=1 forces it, =0 (explicit) never.
  static int region_auto = -1, cus = 0;
  if (region_auto < 0) { const char* ea = getenv("LMM_REGION_ALL"); region_auto = ea ? 0 : 1; }
  if (cus == 0) { int dev = 0; cus = 256; if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev); }
  bool region_base = !g_region_whole;
  if (region_auto && g_region_cols >= 1024 && NC > g_region_cols && (NC % 128) == 0) {
    const long long tasks = 2LL * (g_region_cols / 128) + ((NR + 127) / 128 - g_region_cols / 128);      // of the first (tallest) block column
    // (concurrent batches count together: configs[3]'s 8-latent batches at n = 8192 fit the rule one by one, but four of them in
    // flight do not -- 724 -> 748 ms per step with the dataflow base case, the same effect as for C2's two 16-latent batches)
    region_base = tasks * B.nb * g_slots_in_flight <= (long long)(2.3 * cus);
  }
  // Round 4: when the 1024-column block column is over that bound -- many matrices per launch: C2's 16-latent batches, 32 latents at
  // n = 2048 ... 4096 -- the base case is a 512-column block column instead of the panel launches: its row chains are 10 units long
  // instead of 36 (finer tasks for a machine that is oversubscribed many times over), the square's workgroups hold their CUs half as
  // long, and the K = 512 update between two of them runs at 60-64 TFLOP/s.  32 x 2048: 2.98 -> 2.64 ms, 32 x 4096: 15.2 -> 14.0,
  // 16 x 16384 on one stream: 346.8 -> 343.6, C2 at N = 1: 685.7-687.5 -> 683.1-684.2 ms (LMM_REGION_SMALL=<columns>, 0: panel launches).
  static int region_small = -1;
  if (region_small < 0) { const char* e = getenv("LMM_REGION_SMALL"); region_small = e ? atoi(e) : 512; if (region_small % 128) region_small = 0; if (region_small > g_region_cols) region_small = g_region_cols; }
  int region_cols = g_region_cols;
  if (region_auto && g_region_cols >= 1024 && NC > g_region_cols && (NC % 128) == 0 && !region_base && region_small > 0) { region_cols = region_small; region_base = true; }
  const bool region_here = region_cols > 0 && (region_base || (NC <= region_cols && (NC % 128) == 0));
  if (region_here) {                       // dependency flags of the region launches: this stream's slice of the persistent, once-zeroed
    int si = -1;                           // array (launches are told apart by epoch); an unknown stream gets a zeroed scratch
    for (int s = 0; s < kMaxStreams; ++s) if (g.streams[s] == st) si = s;
    const size_t fi = region_flag_ints(NR);
    int* fl;
    if (si >= 0 && g.region_flags) fl = g.region_flags + (size_t)si * LMM_MAX_BATCH * fi;
    else {
      fl = reinterpret_cast<int*>(call_scratch((fi * B.nb + 1) / 2));
      HIPCHK(hipMemsetAsync(fl, 0, fi * sizeof(int) * B.nb, st));
    }
    for (int j = 0; j < B.nb; ++j) flags.p[j] = fl + fi * j;
  }
  // LMM_FUSE_BULK (default 1): the bulk rows of a panel ride in the update launch that factors its diagonal block (NODE_FUSE), behind
  // per-call flags zeroed here; not when the region kernel serves as the base case (it solves the first panel's rows itself).
  static int fuse_bulk = -1;
  if (fuse_bulk < 0) { const char* e = getenv("LMM_FUSE_BULK"); fuse_bulk = e ? (atoi(e) != 0) : 1; }
  // ... in the launches with 512 <= K <= 2048 (LMM_FUSE_BULK_MINK / _MAXK): below, the chain update -> leaf -> bulk inside one launch
  // is no shorter than two launches (K = 128: 115 us against 72 + 34); above, the fused build's 3-4 spilled registers cost the long
  // launches more (0.3 % of 11-30 ms) than the bulk launch they absorb (33 us)
  constexpr int fuse_min_k = 512, fuse_max_k = 2048;
  NodeFlags nfl;
  nfl.min_k = fuse_min_k; nfl.max_k = fuse_max_k;
  nfl.rows_real = rows_real;
  nfl.region_cols = region_here ? region_cols : 0;
  if (fuse_bulk && !g_strict_progress && !region_here && NC > 128) {
    nfl.stride = (int)node_flag_ints(NR);
    const size_t ints = (size_t)nfl.stride * B.nb;
    nfl.p = reinterpret_cast<int*>(call_scratch((ints + 1) / 2));
    HIPCHK(hipMemsetAsync(nfl.p, 0, ints * sizeof(int), st));
  }
  if (region_here && std::min(NC, region_cols) > 64 * LMM_REGION_ASST_MIN_R) {      // block columns with assistant rows: their scratch tiles
    const size_t per_s = (size_t)LMM_REGION_ASST_TILES * 4096;
    double* sb = call_scratch(per_s * B.nb);
    for (int j = 0; j < B.nb; ++j) nfl.S.p[j] = sb + per_s * j;
  }
  potrf_rec_panel(B, W2, flags, nfl, ld, NR, 0, NC, n_real, st, false);
}

void potrf_rec(double* A, int ld, int NR, int j0, int w, double* W, int n_real, int* info, hipStream_t st) {
  Batch B; B.add(A, W, info);
  if (j0 == 0) potrf_batch(B, ld, NR, w, n_real, st);
  else potrf_rec(B, ld, NR, j0, w, n_real, st);
}

// R (nr x NC, ldr) <- R * L^-T for an already factored L (ld) with inverse diagonal blocks W.
// tri: R starts as the identity and becomes the upper triangular L^-T; rows below the current column block are still
// zero and are skipped (~n^3/3 flops instead of n^3 for a rectangular solve).
// The same solve for a batch of (R_j, L_j, W_j) of identical shapes in lock-step launches (blockIdx.y = matrix).
void trsm_rec(const BatchPtr& R, int ldr, int nr, const BatchPtr& L, int ld, const BatchPtr& W, int nb, int j0, int w,
              hipStream_t st, bool tri = false, bool top = true) {
  if (top)
    for (int j = 0; j < nb; ++j) {
      guard_extent(R.p[j], nr, ldr, (size_t)j0 + w, true, "batched triangular solve (R)");
      guard_extent(L.p[j], (size_t)j0 + w, ld, (size_t)j0 + w, true, "batched triangular solve (L)");
      guard_extent(W.p[j], 64, 64, (size_t)((j0 + w) / 64) * 64, true, "batched triangular solve (inverse blocks)");
    }
  if (w <= 64) {
    const int rows = tri ? std::min(nr, j0 + 64) : nr;
    ProfScope ps(LMM_PROF_SOLVE_LEAF, (double)nb * rows * 64.0 * 64.0, st, rows, 64, 64);       // 64-column solve by the inverse block: rows * 64^2 flops
    launch_gemm_nt(R, (size_t)j0 * ldr, ldr, R, (size_t)j0 * ldr, ldr, W, (size_t)(j0 / 64) * 4096, 64, rows, 64, 64, 0, true, nb, st);
    return;
  }
  const int h = split(w);
  trsm_rec(R, ldr, nr, L, ld, W, nb, j0, h, st, tri, false);
  const int rows = tri ? std::min(nr, j0 + h) : nr;
  {
    // R[:, j0+h : j0+w] -= R[:, j0 : j0+h] L[j0+h : j0+w, j0 : j0+h]': 2 rows (w - h) h flops; bytes: the target block read + written,
    // both operand blocks read once
    const double r = rows, c = w - h, k = h;
    ProfScope ps(LMM_PROF_SOLVE, (double)nb * 2.0 * r * c * k, st, rows, w - h, h, (double)nb * (16.0 * r * c + 8.0 * r * k + 8.0 * c * k));
    launch_gemm_nt(R, (size_t)(j0 + h) * ldr, ldr, R, (size_t)j0 * ldr, ldr, L, (size_t)j0 * ld + (j0 + h), ld, rows, w - h, h, 0, false,
                   nb, st);
  }
  trsm_rec(R, ldr, nr, L, ld, W, nb, j0 + h, w - h, st, tri, false);
}

// One matrix: the batch of one (element offsets are applied inside the kernels in units of the storage type, so the solve is
// correct in the fp32 compute mode too -- pointer arithmetic on the opaque double* would not be)
void trsm_rec(double* R, int ldr, int nr, const double* L, int ld, const double* W, int j0, int w, hipStream_t st, bool tri = false) {
  BatchPtr Rb{}, Lb{}, Wb{};
  Rb.p[0] = R; Lb.p[0] = const_cast<double*>(L); Wb.p[0] = const_cast<double*>(W);
  trsm_rec(Rb, ldr, nr, Lb, ld, Wb, 1, j0, w, st, tri);
}

// alpha (in place over z = L^-1 delta) <- L^-T z for one factor matrix
void backsolve1(const double* L, int ld, const double* W, int nblk, double* z, hipStream_t st) {
  BatchPtr Lb{}, Wb{}, zb{};
  Lb.p[0] = const_cast<double*>(L); Wb.p[0] = const_cast<double*>(W); zb.p[0] = z;
  launch_backsolve(Lb, ld, Wb, nblk, zb, 1, st);
}

// How many latents share one batch and how many streams carry batches, for a shard of ms latents.
// bytes_per_latent > 0: the working set one latent needs while its batch is in flight; the plan is then shrunk (batch first,
// streams second) until nb_per * nstreams * bytes_per_latent fits in 80 % of the free device memory (+ this library's cache).
void batch_plan(int ms, int* nb_per, int* nstreams_used, double bytes_per_latent = 0.0) {
  static int bmax = -1;
  if (bmax < 0) { const char* e = getenv("LMM_BATCH"); bmax = e ? atoi(e) : 16; if (bmax < 1) bmax = 1; if (bmax > LMM_MAX_BATCH) bmax = LMM_MAX_BATCH; }   // round 2: 16 (C2: 690 vs 696 ms at 8)
  // ONE lock-step batch per `bmax` latents (with the software-pipelined update kernel a fuller batch beats two smaller ones on two
  // streams at every size: C2 share of 4 latents 99.5 ms as one batch, 102.3 as 2 x 2, 105.4 as 4 x 1, round 2); concurrent streams
  // carry the batches of larger shards.
  // small factor matrices (n <= ~2000) are latency-bound end to end: one lock-step batch of up to LMM_MAX_BATCH latents costs the
  // same leaf chain as a batch of 8 (reference notebook shape, 20 latents: 3 batches of 8/8/4 -> one of 20)
  constexpr int bsmall = LMM_MAX_BATCH;
  const int bcap = (bytes_per_latent > 0.0 && bytes_per_latent <= 4e7) ? std::max(bmax, bsmall) : bmax;
  int b = std::min(bcap, std::max(1, ms));
  if (g.prof && g.prof_serial) b = std::min(bcap, ms);  // instrumented pass: production-sized batches on ONE stream
  int nbatches = (ms + b - 1) / b;
  int ns = std::max(1, std::min(nbatches, eff_streams()));
  if ((double)b * ns * bytes_per_latent > 8e9) {      // small working sets never need the (slow) driver query
    size_t fr = 0, tot = 0;
    if (hipMemGetInfo(&fr, &tot) == hipSuccess) {
      double avail = (double)fr;
      for (const auto& kv : g.pool) avail += (double)kv.first;
      const double budget = 0.8 * avail;
      while ((double)b * ns * bytes_per_latent > budget && (b > 1 || ns > 1)) {
        if (b > 1) b = (b + 1) / 2; else --ns;
        nbatches = (ms + b - 1) / b;
        ns = std::max(1, std::min(ns, nbatches));
      }
    } else (void)hipGetLastError();
  }
  *nb_per = b;
  *nstreams_used = ns;
}

struct Slot {                 // one stream + the factor matrices of the batch it carries
  std::vector<Buf<double>> A, W;
  hipStream_t st;
};

void make_slots(std::vector<Slot>& slots, int count, int nb_per, size_t a_elems, int NC) {
  slots.resize(count);
  for (int s = 0; s < count; ++s) {
    for (int j = 0; j < nb_per; ++j) {
      slots[s].A.emplace_back(mat_count(a_elems));
      slots[s].W.emplace_back(mat_count((size_t)(NC / 64) * 4096));
    }
    slots[s].st = g.streams[s];
  }
}

void fork_slots(int count) {       // slot streams wait for everything queued on the main stream
  g_slots_in_flight = count > 1 ? count : 1; g_concurrent_batches = g_slots_in_flight;
  if (count <= 1) return;          // (one slot = the main stream itself: no event -- a recorded event is a marker packet the queue
                                   // takes microseconds to retire, in front of a small problem's Gram launch)
  HIPCHK(hipEventRecord(g.ev_main, g.streams[0]));
  for (int s = 1; s < count; ++s) HIPCHK(hipStreamWaitEvent(g.streams[s], g.ev_main, 0));
}

void join_slots(int count) {       // main stream waits for every slot stream
  g_slots_in_flight = 1; g_concurrent_batches = 1;
  for (int s = 1; s < count; ++s) {
    HIPCHK(hipEventRecord(g.ev_slot[s], g.streams[s]));
    HIPCHK(hipStreamWaitEvent(g.streams[0], g.ev_slot[s], 0));
  }
}

// Pivot-info words of a batch -> status.  A dependency-wait timeout of potrf_region_kernel (LMM_INFO_SYNC_TIMEOUT in ANY word) outranks
// a PosDefException in an earlier latent: it means the launch was drained with results undefined, which must never be reported as a
// property of the caller's matrix.  (The abort words the kernel raised are epoch-tagged, so they need no reset: a later launch never
// matches them.)
int check_info(const int* info, size_t count, int latent_begin) {
  for (size_t k = 0; k < count; ++k)
    if (info[k] == LMM_INFO_SYNC_TIMEOUT)
      return fail(LMM_ERR_HIP, "potrf_region_kernel: a dependency wait timed out (latent %d); the grid was drained, results are invalid",
                  latent_begin + (int)k);
  for (size_t k = 0; k < count; ++k)
    if (info[k] != 0) {
      g.err_latent = latent_begin + (int)k;
      g.err_info = info[k];
      return fail(LMM_ERR_NOT_PD, "PosDefException: matrix is not positive definite; Cholesky factorization failed "
                                  "(latent %d, pivot %d)", g.err_latent, g.err_info);
    }
  return LMM_OK;
}
int check_info(const std::vector<int>& info, int latent_begin) { return check_info(info.data(), info.size(), latent_begin); }

// ------------------------------------------------------------------------------------------------
// host-side small dense algebra (m, p <= a few hundred): projections and regulariser scalars
// ------------------------------------------------------------------------------------------------
// reference src/oilmm.jl:20-30:  T = sqrt(S) \ U'  (m x p, column-major), SigmaT = sigma2 ./ S
// T, H: caller's buffers (m p doubles each; may be pinned memory), ST: m.  H by contiguous columns, T (the transposed image) in
// blocks of 16 outputs so that both its reads and its writes stay in a few cache lines: 11 us instead of 90 at p = 600, m = 20
// (the reference notebook's shape, where this host loop was a fifth of the evaluation); same operations, same results.
void project_orthogonal_into(const double* U, const double* S, int p, int m, double s2, double* T, double* ST, double* H) {
  for (int l = 0; l < m; ++l) {
    const double rs = std::sqrt(S[l]);
    ST[l] = s2 / S[l];
    const double* u = U + (size_t)l * p;
    double* h = H + (size_t)l * p;
    for (int o = 0; o < p; ++o) h[o] = u[o] * rs;          // reference src/orthogonal_matrix.jl:27-30
  }
  for (int o0 = 0; o0 < p; o0 += 16) {
    const int o1 = std::min(p, o0 + 16);
    for (int l = 0; l < m; ++l) {
      const double rs = std::sqrt(S[l]);
      const double* u = U + (size_t)l * p;
      for (int o = o0; o < o1; ++o) T[l + (size_t)o * m] = u[o] / rs;
    }
  }
}
void project_orthogonal(const double* U, const double* S, int p, int m, double s2, std::vector<double>& T,
                        std::vector<double>& ST, std::vector<double>& H) {
  T.resize((size_t)m * p); ST.resize(m); H.resize((size_t)p * m);
  project_orthogonal_into(U, S, p, m, s2, T.data(), ST.data(), H.data());
}

bool host_cholesky(std::vector<double>& A, int m) {   // lower, in place, column-major
  for (int j = 0; j < m; ++j) {
    double d = A[j + (size_t)j * m];
    for (int k = 0; k < j; ++k) d -= A[j + (size_t)k * m] * A[j + (size_t)k * m];
    if (!(d > 0.0)) return false;
    d = std::sqrt(d);
    A[j + (size_t)j * m] = d;
    for (int i = j + 1; i < m; ++i) {
      double s = A[i + (size_t)j * m];
      for (int k = 0; k < j; ++k) s -= A[i + (size_t)k * m] * A[j + (size_t)k * m];
      A[i + (size_t)j * m] = s / d;
    }
  }
  return true;
}

// Symmetric eigendecomposition A = Q diag(lam) Q' by cyclic Jacobi (m <= a few hundred; host).  A, Q column-major.
void host_jacobi_eig(std::vector<double> A, int m, std::vector<double>& lam, std::vector<double>& Q) {
  Q.assign((size_t)m * m, 0.0);
  for (int i = 0; i < m; ++i) Q[i + (size_t)i * m] = 1.0;
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0.0, diag = 0.0;
    for (int i = 0; i < m; ++i) for (int j = 0; j < m; ++j) (i == j ? diag : off) += A[i + (size_t)j * m] * A[i + (size_t)j * m];
    if (off <= 1e-30 * diag) break;
    for (int pi = 0; pi < m - 1; ++pi)
      for (int qi = pi + 1; qi < m; ++qi) {
        const double apq = A[pi + (size_t)qi * m];
        if (apq == 0.0) continue;
        const double theta = (A[qi + (size_t)qi * m] - A[pi + (size_t)pi * m]) / (2.0 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < m; ++k) {       // A <- A J
          const double akp = A[k + (size_t)pi * m], akq = A[k + (size_t)qi * m];
          A[k + (size_t)pi * m] = c * akp - s * akq; A[k + (size_t)qi * m] = s * akp + c * akq;
        }
        for (int k = 0; k < m; ++k) {       // A <- J' A
          const double apk = A[pi + (size_t)k * m], aqk = A[qi + (size_t)k * m];
          A[pi + (size_t)k * m] = c * apk - s * aqk; A[qi + (size_t)k * m] = s * apk + c * aqk;
        }
        for (int k = 0; k < m; ++k) {       // Q <- Q J
          const double qkp = Q[k + (size_t)pi * m], qkq = Q[k + (size_t)qi * m];
          Q[k + (size_t)pi * m] = c * qkp - s * qkq; Q[k + (size_t)qi * m] = s * qkp + c * qkq;
        }
      }
  }
  lam.resize(m);
  for (int i = 0; i < m; ++i) lam[i] = A[i + (size_t)i * m];
}

// reference src/ilmm.jl:61-68.  T m x p, ST m x m (column-major); also logdet(ST) for src/ilmm.jl:179.
int project_dense(const double* H, int p, int m, double s2, double jitter, std::vector<double>& T,
                  std::vector<double>& ST, double* logdetST) {
  std::vector<double> G((size_t)m * m, 0.0);
  for (int a = 0; a < m; ++a)
    for (int b = 0; b < m; ++b) {
      double s = 0.0;
      for (int o = 0; o < p; ++o) s += H[o + (size_t)a * p] * H[o + (size_t)b * p];
      G[a + (size_t)b * m] = s / s2 + (a == b ? jitter : 0.0);
    }
  if (!host_cholesky(G, m)) return fail(LMM_ERR_NOT_PD, "PosDefException in project(H, sigma2): H'H/sigma2 + 1e-9 I not PD");
  T.assign((size_t)m * p, 0.0);
  for (int o = 0; o < p; ++o) {     // solve (L L') t = H[o,:]' / s2
    std::vector<double> v(m);
    for (int a = 0; a < m; ++a) {
      double s = H[o + (size_t)a * p] / s2;
      for (int k = 0; k < a; ++k) s -= G[a + (size_t)k * m] * v[k];
      v[a] = s / G[a + (size_t)a * m];
    }
    for (int a = m - 1; a >= 0; --a) {
      double s = v[a];
      for (int k = a + 1; k < m; ++k) s -= G[k + (size_t)a * m] * v[k];
      v[a] = s / G[a + (size_t)a * m];
    }
    for (int a = 0; a < m; ++a) T[a + (size_t)o * m] = v[a];
  }
  ST.assign((size_t)m * m, 0.0);
  for (int a = 0; a < m; ++a)
    for (int b = 0; b < m; ++b) {
      double s = 0.0;
      for (int o = 0; o < p; ++o) s += T[a + (size_t)o * m] * T[b + (size_t)o * m];
      ST[a + (size_t)b * m] = s2 * s;
    }
  if (logdetST) {
    std::vector<double> C = ST;
    if (!host_cholesky(C, m)) return fail(LMM_ERR_NOT_PD, "PosDefException: SigmaT not PD");
    double ld = 0.0;
    for (int a = 0; a < m; ++a) ld += std::log(C[a + (size_t)a * m]);
    *logdetST = 2.0 * ld;
  }
  return LMM_OK;
}

struct Uploaded {   // small host arrays staged on the device
  Buf<double> buf;
  Uploaded() = default;
  // direct_small: up to 256 values that kernels only READ (a few times, as uniform loads) are left in the pinned arena and read
  // from there through its device mapping -- one copy operation less per call, which is ~8 % of a C0-sized evaluation
  Uploaded(const std::vector<double>& v, hipStream_t st, bool direct_small = false) {
    void* pp = pin_take(v.size() * sizeof(double));
    if (pp) std::memcpy(pp, v.data(), v.size() * sizeof(double));
    stage(pp ? static_cast<const double*>(pp) : v.data(), pp != nullptr, v.size(), st, direct_small);
  }
  // values the caller already wrote into the pinned arena (pin_take)
  Uploaded(const double* pinned, size_t count, hipStream_t st, bool direct_small) { stage(pinned, true, count, st, direct_small); }
 private:
  void stage(const double* src, bool pinned, size_t count, hipStream_t st, bool direct_small) {
    if (direct_small && pinned && count <= 256 && g.pin_dev) {
      buf.p = pin_dev(const_cast<double*>(src)); buf.n = count; buf.own = false;
      return;
    }
    buf = Buf<double>(count);
    HIPCHK(hipMemcpyAsync(buf.p, src, count * sizeof(double), hipMemcpyHostToDevice, st));
  }
};

// Ty (n x C, ld n) = In (n x K) * Mx' - sub;  optionally residual sum of squares |Ref - Ty * Hm'|_F^2.
void project_on_device(const double* Y, int n, int p, const double* Td, int m, int c0, int C,
                       const double* sub_dev, double* Ty, hipStream_t st) {
  launch_tall_skinny(Y, n, n, p, Td + c0, m, C, Ty, n, sub_dev, nullptr, 0, nullptr, 0, st);
}
void project_on_device(const double* Y, int n, int p, const Buf<double>& Td, int m, int c0, int C,
                       const double* sub_dev, double* Ty, hipStream_t st) {
  project_on_device(Y, n, p, Td.p, m, c0, C, sub_dev, Ty, st);
}

void residual_on_device(const double* Y, int n, int p, const double* Ty_all, int m, const double* Hd,
                        double* partial, double* out1, hipStream_t st) {
  launch_tall_skinny(Ty_all, n, n, m, Hd, p, p, nullptr, 0, nullptr, Y, n, partial, 1, st);
  launch_sum_partials(partial, tall_skinny_partials(n, p), out1, st);
}
void residual_on_device(const double* Y, int n, int p, const double* Ty_all, int m, const Buf<double>& Hd,
                        double* partial, double* out1, hipStream_t st) {
  residual_on_device(Y, n, p, Ty_all, m, Hd.p, partial, out1, st);
}

// Core: per-latent log marginal likelihoods for latents [l0, l1) given the device rider vectors
// delta ([latent][rhs][n]) and per-latent noise.  Returns lml[latent * nrhs + rhs] (host).  nrhs > 1: several
// right-hand sides (matrix-Y logpdf) ride one factorisation.  noisevec ([latent of the shard][n], device) replaces the
// scalar per-latent noise by a per-point diagonal.
int latent_lmls(const double* xd, int d, int n, const lmm_gp_t* gps, const double* noise, int l0, int l1,
                const double* delta, std::vector<double>& lml, int nrhs = 1, const double* noisevec = nullptr,
                const double* rider_sub = nullptr,         // rider_sub[latent] (host): subtracted from that latent's riders
                const std::function<void()>* pre_launch = nullptr) {      // launches that PRODUCE delta, issued on streams[0] once this
                                                           // function's host-side preparation is done (see lmm_oilmm_logpdf)
  const int ms = l1 - l0;
  lml.assign((size_t)ms * nrhs, 0.0);
  if (ms == 0) {
    if (pre_launch) (*pre_launch)();
    // callers read pinned results (regulariser residual) and release their device buffers after this returns: the main stream
    // must be drained even when this rank holds no latent
    HIPCHK(hipStreamSynchronize(g.streams[0]));
    return LMM_OK;
  }
  Dims D(n, nrhs);
  int nb_per = 1, nslots = 1;
  batch_plan(ms, &nb_per, &nslots, mat_bytes((double)D.elems()));
  std::vector<Slot> slots;
  make_slots(slots, nslots, nb_per, D.elems(), D.NC);
  // results: [ms * nrhs doubles | ms pivot-info ints] in ONE buffer.  When the pinned arena is mapped into the device, lml_reduce
  // writes both straight into host memory (pk) and nothing is copied back; else one copy brings both back.  The pivot-info words
  // the kernels work on are zeroed by each latent's Gram launch (no memset).
  const size_t nout = (size_t)ms * nrhs;
  const size_t nbytes = (nout + ((size_t)ms + 1) / 2) * sizeof(double);
  Buf<double> out(nout + ((size_t)ms + 1) / 2);
  struct { int* p; } info{reinterpret_cast<int*>(out.p + nout)};
  std::vector<double> pageable;
  char* pk = static_cast<char*>(pin_take(nbytes));
  char* pk_dev = pin_dev(pk);
  if (!pk) { pageable.resize(nbytes / sizeof(double)); pk = reinterpret_cast<char*>(pageable.data()); }
  // The kernels that produce the riders go out only now: issued before the plan / slot / pool work above, they finished while the
  // host was still preparing and the device then idled ~5 us ahead of the Gram launch (a twentieth of a C0-sized evaluation)
  if (pre_launch) (*pre_launch)();
  fork_slots(nslots);
  int bi = 0;
  for (int k0 = 0; k0 < ms; k0 += nb_per, ++bi) {
    Slot& s = slots[bi % nslots];
    const int nb = std::min(nb_per, ms - k0);
    Batch B;
    {
    // the batch's Gram launches share one event pair (back-to-back launches: the event overhead is not charged per launch)
    const double gb = (double)n * ((double)n + 1.0) / 2.0 * 8.0;
    ProfScope ps(LMM_PROF_GRAM, nb * gb, s.st, 0, 0, 0, nb * gb, nb);
    GramArgs ga[LMM_MAX_BATCH];
    for (int j = 0; j < nb; ++j) {
      const int k = k0 + j;
      const lmm_gp_t& gp = gps[l0 + k];
      GramArgs a{};
      a.A = s.A[j].p; a.ld = D.ld; a.nrows = D.NR; a.ncols = D.NC; a.row_tile0 = 0; a.row_shift = 0; a.full = 0;
      a.x = xd; a.d = d; a.n = n; a.kind = gp.kind; a.var = gp.variance; a.inv_ls = 1.0 / gp.lengthscale;
      a.diag_add = noisevec ? 0.0 : noise[l0 + k]; a.pad_diag = 1.0;
      a.diag_vec = noisevec ? noisevec + (size_t)k * n : nullptr;      // per-point noise of latent k (device, n values)
      a.rider = delta + (size_t)k * nrhs * n; a.rider_ld = n; a.nrider = nrhs; a.xs = nullptr; a.ns = 0;
      a.rider_sub = rider_sub ? rider_sub[l0 + k] : 0.0;
      a.info_zero = info.p + k;
      ga[j] = a;
      B.add(s.A[j].p, s.W[j].p, info.p + k);
    }
    gram_batch_g(ga, nb, s.st);       // one launch per run of equal kernel kinds (blockIdx.z = latent)
    }
    potrf_batch(B, D.ld, D.NR, D.NC, n, s.st, D.NC + nrhs);      // the rider rows NC + nrhs .. NR - 1 are zero padding
    if (pk_dev) launch_lml_reduce(B.A, nb, D.ld, n, D.NC, nrhs, reinterpret_cast<double*>(pk_dev) + (size_t)k0 * nrhs, s.st, &B.info,
                                  reinterpret_cast<int*>(pk_dev + nout * sizeof(double)) + k0);
    else launch_lml_reduce(B.A, nb, D.ld, n, D.NC, nrhs, out.p + (size_t)k0 * nrhs, s.st);
  }
  join_slots(nslots);
  std::vector<int> hinfo(ms);
  if (!pk_dev) HIPCHK(hipMemcpyAsync(pk, out.p, nbytes, hipMemcpyDeviceToHost, g.streams[0]));
  HIPCHK(hipStreamSynchronize(g.streams[0]));
  std::memcpy(lml.data(), pk, nout * sizeof(double));
  std::memcpy(hinfo.data(), pk + nout * sizeof(double), (size_t)ms * sizeof(int));
  return check_info(hinfo, l0);
}

const lmm_jitters_t kDefaultJit = {1e-9, 1e-12, 1e-18};

void drain_after_error() {
  if (g.init) { (void)hipDeviceSynchronize(); (void)hipGetLastError(); }
  // a throw between fork_slots and join_slots must not leave "several batches in flight" behind: the base-case rule of potrf_batch
  // and the ragged-row choice of the update launches read these, so later calls would silently take another launch plan
  g_slots_in_flight = 1; g_concurrent_batches = 1;
  if (g.init) strict_ticket_reset();
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// posterior handle
// ------------------------------------------------------------------------------------------------
struct lmm_post {
  int f32 = 0;                // compute dtype the state was built in (its matrices are float buffers when 1)
  int kind = 0;               // 0: per-latent (OILMM / MOGP), 1: dense ILMM
  int n = 0, d = 0, l0 = 0, l1 = 0, m = 0;
  int NC = 0, NR = 0, ld = 0;
  std::vector<lmm_gp_t> gps;  // all m latents (host)
  Buf<double> x;              // d x n
  std::vector<Buf<double>> L; // per latent of the shard: factor matrix (NR x NC, ld)
  std::vector<Buf<double>> W; // inverse diagonal blocks
  std::vector<Buf<double>> alpha;     // C \\ delta
  std::vector<Buf<double>> z;         // per latent: L^-1 delta (contiguous copy of the rider row)
  // kept for sequential conditioning: projected residuals (T y)_l - mean_l, [latent of the shard][n], and the projected noise:
  // one scalar per latent after a first conditioning, per-point values ([latent][n]) once batches with different noise mix
  Buf<double> delta_all, noise_all;
  std::vector<double> noise_scalar;
  // dense ILMM (kind 1): L[0] is the (mn) x (mn) factor, alpha[0] the (mn) weights
  int p = 0;
  Buf<double> ddelta;           // (mn): projected residuals [latent][point]  (kept for sequential conditioning)
  std::vector<double> sigs;     // nbatch x (m x m): SigmaT of every conditioning batch (host)
  std::vector<int> sigidx;      // n: batch index of every training point (host)
  std::vector<double> H;        // p x m column-major (host)
  Buf<LatentDev> latd;          // device latent descriptors
  // latent view of a dense-H posterior (lmm_ilmm_post_latent_view): the device state belongs to `base`; this handle only replaces
  // H by I_m (p = m).  `views` counts the views alive on a base handle; destroying a base that still has views defers its release
  // (zombie) until the last view is destroyed.
  lmm_post* base = nullptr;
  int views = 0;
  bool zombie = false;
};
// the handle that owns the device state of a dense-H posterior (itself, or the base of a latent view)
static inline const lmm_post* dense_state(const lmm_post* P) { return P->base ? P->base : P; }

#define LMM_TRY try {
// A throw unwinds through Buf destructors, which hand device blocks back to the caching pool while slot streams may still be
// running kernels on them: drain the device before the caller can issue the next call (which could be given those blocks).
#define LMM_CATCH                                   \
  }                                                 \
  catch (int code) { drain_after_error(); return code; }                 \
  catch (const std::exception& e) { drain_after_error(); return fail(LMM_ERR_HIP, "exception: %s", e.what()); }

#define REQUIRE_INIT()                                                           \
  if (!g.init) return fail(LMM_ERR_ARG, "lmm_init() has not been called");      \
  release_call_scratch();                                                        \
  g.pin_off = 0

extern "C" {

int lmm_init(int device) {
  std::lock_guard<std::mutex> lk(g_mu);
  LMM_TRY
  if (g.init) {
    if (g.device == device) return LMM_OK;
    return fail(LMM_ERR_ARG, "already initialised on device %d (one process per GPU)", g.device);
  }
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count == 0) return fail(LMM_ERR_HIP, "no HIP device available (%s)", hipGetErrorString(e));
  if (device < 0 || device >= count) return fail(LMM_ERR_ARG, "device %d out of range (0..%d)", device, count - 1);
  HIPCHK(hipSetDevice(device));
  const char* ns = getenv("LMM_NSTREAMS");
  g.nstreams = ns ? std::max(1, std::min(kMaxStreams, atoi(ns))) : 4;
  for (int s = 0; s < kMaxStreams; ++s) {
    HIPCHK(hipStreamCreateWithFlags(&g.streams[s], hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&g.ev_slot[s], hipEventDisableTiming));
  }
  HIPCHK(hipEventCreateWithFlags(&g.ev_main, hipEventDisableTiming));
  g.pin_cap = 1u << 20;
  if (hipHostMalloc((void**)&g.pin, g.pin_cap, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); g.pin = nullptr; g.pin_cap = 0; }
  g.pin_dev = nullptr;
  {
    static int direct = -1;            // LMM_DIRECT_RESULTS=0: results come back by hipMemcpy (the round-2 path)
    if (direct < 0) { const char* e = getenv("LMM_DIRECT_RESULTS"); direct = e ? (atoi(e) != 0) : 1; }
    void* dp = nullptr;
    if (direct && g.pin && hipHostGetDevicePointer(&dp, g.pin, 0) == hipSuccess) g.pin_dev = static_cast<char*>(dp);
    else (void)hipGetLastError();
  }
  {
    const size_t fi = region_flag_ints(0) * (size_t)LMM_MAX_BATCH * kMaxStreams;
    HIPCHK(hipMalloc((void**)&g.region_flags, fi * sizeof(int)));
    HIPCHK(hipMemset(g.region_flags, 0, fi * sizeof(int)));
    region_flags_register(g.region_flags, fi);
  }
  { const char* e = getenv("LMM_STRICT_PROGRESS"); if (e) g_strict_progress = atoi(e) != 0; }
  g.device = device;
  g.init = true;
  return LMM_OK;
  LMM_CATCH
}

int lmm_shutdown(void) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g.init) return LMM_OK;
  (void)hipDeviceSynchronize();
  strict_ticket_reset();
  if (g.comm) { (void)ncclCommDestroy(g.comm); g.comm = nullptr; g.comm_world = 0; }
  if (g.ev_caller) { (void)hipEventDestroy(g.ev_caller); g.ev_caller = nullptr; }
  release_call_scratch();
  if (g.region_flags) { (void)hipFree(g.region_flags); g.region_flags = nullptr; region_flags_register(nullptr, 0); }
  for (auto& kv : g.pool) (void)hipFree(kv.second);
  g.pool.clear();
  for (int s = 0; s < kMaxStreams; ++s) { (void)hipStreamDestroy(g.streams[s]); (void)hipEventDestroy(g.ev_slot[s]); }
  (void)hipEventDestroy(g.ev_main);
  if (g.pin) { (void)hipHostFree(g.pin); g.pin = nullptr; g.pin_cap = 0; g.pin_dev = nullptr; }
  g.init = false;
  return LMM_OK;
}

const char* lmm_last_error_string(void) { return g.err.c_str(); }

int lmm_last_error_detail(int* latent, int* info) {
  if (latent) *latent = g.err_latent;
  if (info) *info = g.err_info;
  return LMM_OK;
}

int lmm_release_cached_memory(void) {
  std::lock_guard<std::mutex> lk(g_mu);
  REQUIRE_INIT();
  (void)hipDeviceSynchronize();
  for (auto& kv : g.pool) (void)hipFree(kv.second);
  g.pool.clear();
  return LMM_OK;
}

int lmm_stream_wait_caller(void* hip_stream) {
  std::lock_guard<std::mutex> lk(g_mu);
  REQUIRE_INIT();
  LMM_TRY
  if (!g.ev_caller) HIPCHK(hipEventCreateWithFlags(&g.ev_caller, hipEventDisableTiming));
  HIPCHK(hipEventRecord(g.ev_caller, static_cast<hipStream_t>(hip_stream)));
  HIPCHK(hipStreamWaitEvent(g.streams[0], g.ev_caller, 0));      // slot streams fork from streams[0] (fork_slots)
  return LMM_OK;
  LMM_CATCH
}

// ---- RCCL (the ONE exchange step of the sharded paths) -------------------------------------------
#define NCCLCHK(expr)                                                                                  \
  do {                                                                                                 \
    ncclResult_t r_ = (expr);                                                                          \
    if (r_ != ncclSuccess) throw fail(LMM_ERR_RCCL, "%s failed: %s (%s:%d)", #expr, ncclGetErrorString(r_), __FILE__, __LINE__); \
  } while (0)

int lmm_comm_get_unique_id(void* id_out) {
  std::lock_guard<std::mutex> lk(g_mu);
  LMM_TRY
  if (!id_out) return fail(LMM_ERR_ARG, "id_out is NULL");
  static_assert(sizeof(ncclUniqueId) == LMM_UNIQUE_ID_BYTES, "LMM_UNIQUE_ID_BYTES must match ncclUniqueId");
  ncclUniqueId id;
  NCCLCHK(ncclGetUniqueId(&id));
  std::memcpy(id_out, &id, sizeof id);
  return LMM_OK;
  LMM_CATCH
}

int lmm_comm_init_rank(const void* id, int rank, int world) {
  std::lock_guard<std::mutex> lk(g_mu);
  REQUIRE_INIT();
  LMM_TRY
  if (!id || world < 1 || rank < 0 || rank >= world) return fail(LMM_ERR_ARG, "bad arguments");
  if (g.comm) return fail(LMM_ERR_ARG, "a communicator already exists (rank %d of %d): lmm_comm_destroy first", g.comm_rank, g.comm_world);
  ncclUniqueId uid;
  std::memcpy(&uid, id, sizeof uid);
  HIPCHK(hipSetDevice(g.device));
  NCCLCHK(ncclCommInitRank(&g.comm, world, uid, rank));
  g.comm_rank = rank; g.comm_world = world;
  return LMM_OK;
  LMM_CATCH
}

int lmm_comm_info(int* rank, int* world) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (rank) *rank = g.comm ? g.comm_rank : 0;
  if (world) *world = g.comm ? g.comm_world : 0;
  return LMM_OK;
}

static int allreduce_f64(double* buf, size_t count, ncclRedOp_t op) {
  std::lock_guard<std::mutex> lk(g_mu);
  REQUIRE_INIT();
  LMM_TRY
  if (!g.comm) return fail(LMM_ERR_ARG, "no communicator: call lmm_comm_init_rank first");
  if (count == 0) return LMM_OK;
  if (!buf) return fail(LMM_ERR_ARG, "buf is NULL");
  hipStream_t st0 = g.streams[0];
  if (is_device_ptr(buf)) {
    NCCLCHK(ncclAllReduce(buf, buf, count, ncclDouble, op, g.comm, st0));
    HIPCHK(hipStreamSynchronize(st0));
    return LMM_OK;
  }
  // host buffer: stage through the pinned arena when it fits (the scalar logpdf sum always does)
  Buf<double> dev(count);
  double* pin = static_cast<double*>(pin_take(count * sizeof(double)));
  if (pin) std::memcpy(pin, buf, count * sizeof(double));
  HIPCHK(hipMemcpyAsync(dev.p, pin ? pin : buf, count * sizeof(double), hipMemcpyHostToDevice, st0));
  NCCLCHK(ncclAllReduce(dev.p, dev.p, count, ncclDouble, op, g.comm, st0));
  HIPCHK(hipMemcpyAsync(pin ? pin : buf, dev.p, count * sizeof(double), hipMemcpyDeviceToHost, st0));
  HIPCHK(hipStreamSynchronize(st0));
  if (pin) std::memcpy(buf, pin, count * sizeof(double));
  return LMM_OK;
  LMM_CATCH
}
int lmm_allreduce_sum_f64(double* buf, size_t count) { return allreduce_f64(buf, count, ncclSum); }
int lmm_allreduce_max_f64(double* buf, size_t count) { return allreduce_f64(buf, count, ncclMax); }

int lmm_comm_destroy(void) {
  std::lock_guard<std::mutex> lk(g_mu);
  LMM_TRY
  if (g.comm) {
    if (g.init) (void)hipDeviceSynchronize();
    NCCLCHK(ncclCommDestroy(g.comm));
    g.comm = nullptr; g.comm_world = 0; g.comm_rank = 0;
  }
  return LMM_OK;
  LMM_CATCH
}

int lmm_set_compute_dtype(int dtype) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (dtype != LMM_F64 && dtype != LMM_F32) return fail(LMM_ERR_ARG, "dtype must be LMM_F64 or LMM_F32");
  g_f32 = (dtype == LMM_F32) ? 1 : 0;
  return LMM_OK;
}
int lmm_get_compute_dtype(void) { return g_f32 ? LMM_F32 : LMM_F64; }

// Strict forward progress of the dataflow kernels (include/lmm_hip.h, conventions): tasks by arrival ticket instead of blockIdx.x.
int lmm_set_strict_progress(int on) {
  std::lock_guard<std::mutex> lk(g_mu);
  g_strict_progress = on ? 1 : 0;
  return LMM_OK;
}
int lmm_get_strict_progress(void) { return g_strict_progress; }
// test hook: the strict build's workgroups ask for their task indices in reverse order (include/lmm_hip.h)
int lmm_dev_claim_scramble(int on) {
  std::lock_guard<std::mutex> lk(g_mu);
  g_claim_scramble = on ? 1 : 0;
  return LMM_OK;
}

int lmm_set_projection_dtype(int dtype) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (dtype != LMM_PROJ_NATIVE && dtype != LMM_PROJ_BF16 && dtype != LMM_PROJ_BF16X2)
    return fail(LMM_ERR_ARG, "projection dtype must be LMM_PROJ_NATIVE, LMM_PROJ_BF16 or LMM_PROJ_BF16X2");
  g_proj = dtype;
  return LMM_OK;
}
int lmm_get_projection_dtype(void) { return g_proj; }

int lmm_device_synchronize(void) {
  std::lock_guard<std::mutex> lk(g_mu);
  REQUIRE_INIT();
  LMM_TRY
  HIPCHK(hipDeviceSynchronize());
  return LMM_OK;
  LMM_CATCH
}

// reference src/orthogonal_matrix.jl:21-23: isapprox(U'U, I) (Frobenius norm, rtol = sqrt(eps)).
int lmm_orthogonal_validate(const double* U, int p, int m) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (!U || p <= 0 || m <= 0) return fail(LMM_ERR_ARG, "bad arguments");
  double diff2 = 0.0, g2 = 0.0;
  for (int a = 0; a < m; ++a)
    for (int b = 0; b < m; ++b) {
      double s = 0.0;
      for (int o = 0; o < p; ++o) s += U[o + (size_t)a * p] * U[o + (size_t)b * p];
      g2 += s * s;
      const double dlt = s - (a == b ? 1.0 : 0.0);
      diff2 += dlt * dlt;
    }
  const double rtol = std::sqrt(2.220446049250313e-16);
  if (!(std::sqrt(diff2) <= rtol * std::max(std::sqrt(g2), std::sqrt((double)m))))
    return fail(LMM_ERR_NOT_ORTHOGONAL, "`U` is not an orthogonal matrix");
  return LMM_OK;
}

int lmm_oilmm_logpdf(const double* x, int d, int n, const double* y, int p, const double* U, const double* S, int m,
                     double sigma2, const lmm_gp_t* gps, int latent_begin, int latent_end, int with_regulariser,
                     double* out) {
  std::lock_guard<std::mutex> lk(g_mu);
  REQUIRE_INIT();
  LMM_TRY
  if (!x || !y || !U || !S || !out || d <= 0 || n <= 0 || p <= 0 || m <= 0) return fail(LMM_ERR_ARG, "bad arguments");
  if (m > p) return fail(LMM_ERR_DIM, "out dim of x != out dim of f.");
  if (latent_begin < 0 || latent_end > m || latent_begin > latent_end) return fail(LMM_ERR_ARG, "bad latent shard");
  if (int rc = check_gps(gps, m)) return rc;
  if (!(sigma2 > 0.0)) return fail(LMM_ERR_ARG, "sigma2 must be > 0");
  hipStream_t st0 = g.streams[0];
  std::vector<double> T, ST, H;
  // [T | H] straight into the pinned arena when the regulariser path (which uploads both as one block) has room there
  double* packp = with_regulariser ? static_cast<double*>(pin_take(2 * (size_t)m * p * sizeof(double))) : nullptr;
  if (packp) { ST.resize(m); project_orthogonal_into(U, S, p, m, sigma2, packp, ST.data(), packp + (size_t)m * p); }
  else project_orthogonal(U, S, p, m, sigma2, T, ST, H);
  DevIn xd(x, (size_t)d * n, st0), yd(y, (size_t)n * p, st0);
  std::vector<double> means(m);
  for (int l = 0; l < m; ++l) means[l] = gps[l].mean;
  const int l0 = latent_begin, l1 = latent_end, ms = l1 - l0;
  double resid_pageable = 0.0;
  double* resid = static_cast<double*>(pin_take(sizeof(double)));      // pinned: the read-back below does not stall the host
  if (resid == nullptr) resid = &resid_pageable;
  *resid = 0.0;
  std::vector<double> lml;
  if (with_regulariser) {
    // ONE upload [T | H]; ONE projection T*Y of all m latents serves the regulariser's residual and, through rider rows that
    // subtract the latent mean inside the Gram kernel, the per-latent right-hand sides delta_l = (T y)_l - mean_l
    std::vector<double> pack;
    if (!packp) { pack = T; pack.insert(pack.end(), H.begin(), H.end()); }
    Uploaded THd = packp ? Uploaded(packp, 2 * (size_t)m * p, st0, true) : Uploaded(pack, st0, true);
    const double* Tdev = THd.buf.p;
    const double* Hdev = THd.buf.p + (size_t)m * p;
    Buf<double> Ty((size_t)n * m), resid_dev(1), partial(tall_skinny_partials(n, p));
    double* resid_direct = (resid != &resid_pageable) ? pin_dev(resid) : nullptr;      // the reduction writes into host memory
    const std::function<void()> produce = [&]() {
      project_on_device(yd.p, n, p, Tdev, m, 0, m, nullptr, Ty.p, st0);
      // reference src/oilmm.jl:112: sum(abs2, (I - U U') Y)  ==  |Y - H T Y|_F^2 since H T = U U'
      residual_on_device(yd.p, n, p, Ty.p, m, Hdev, partial.p, resid_direct ? resid_direct : resid_dev.p, st0);
      if (!resid_direct) HIPCHK(hipMemcpyAsync(resid, resid_dev.p, sizeof(double), hipMemcpyDeviceToHost, st0));    // read after latent_lmls' sync
    };
    if (int rc = latent_lmls(xd.p, d, n, gps, ST.data(), l0, l1, Ty.p + (size_t)l0 * n, lml, 1, nullptr, means.data(), &produce)) return rc;
  } else {
    Uploaded Td(T, st0), meansd(means, st0);
    Buf<double> delta((size_t)n * std::max(ms, 1));
    if (ms > 0) project_on_device(yd.p, n, p, Td.buf, m, l0, ms, meansd.buf.p + l0, delta.p, st0);
    if (int rc = latent_lmls(xd.p, d, n, gps, ST.data(), l0, l1, delta.p, lml)) return rc;
  }
  double total = 0.0;
  for (int k = 0; k < ms; ++k) total += lml[k];
  if (with_regulariser) {
    // reference src/oilmm.jl:101-113
    double logdetS = 0.0;
    for (int l = 0; l < m; ++l) logdetS += std::log(S[l]);
    total += -((double)n * (logdetS + (double)(p - m) * std::log(2.0 * M_PI * sigma2)) + *resid / sigma2) / 2.0;
  }
  *out = total;
  return LMM_OK;
  LMM_CATCH
}

}  // extern "C"

namespace {

NoiseBlocks one_noise_block(int n, double s2) {
  NoiseBlocks nb{};
  nb.nblk = 1; nb.off[0] = 0; nb.off[1] = n; nb.s2[0] = s2;
  return nb;
}
// the conditioning batches (batch_n[b] points with variance batch_s2[b]), optionally followed by ns test points with variance s2s
NoiseBlocks batch_noise_blocks(const int* batch_n, const double* batch_s2, int nbatch, int ns, double s2s) {
  NoiseBlocks nb{};
  nb.off[0] = 0;
  for (int b = 0; b < nbatch; ++b) { nb.off[b + 1] = nb.off[b] + batch_n[b]; nb.s2[b] = batch_s2[b]; }
  nb.nblk = nbatch;
  if (ns > 0) { nb.off[nbatch + 1] = nb.off[nbatch] + ns; nb.s2[nbatch] = s2s; nb.nblk = nbatch + 1; }
  return nb;
}
// argument checks shared by the two *_post_logpdf_grad_seq entries; n = total number of conditioning points
int check_batches(const int* batch_n, const double* batch_s2, int nbatch, int n) {
  if (!batch_n || !batch_s2 || nbatch < 1) return fail(LMM_ERR_ARG, "bad arguments");
  if (nbatch > LMM_MAX_NOISE_BLOCKS - 1) return fail(LMM_ERR_UNSUPPORTED, "more than 7 conditioning batches with their own noise variance");
  long long tot = 0;
  for (int b = 0; b < nbatch; ++b) {
    if (batch_n[b] <= 0) return fail(LMM_ERR_ARG, "empty conditioning batch");
    if (!(batch_s2[b] > 0.0)) return fail(LMM_ERR_ARG, "sigma2 must be > 0");
    tot += batch_n[b];
  }
  if (tot != n) return fail(LMM_ERR_ARG, "batch sizes do not add up to n");
  return LMM_OK;
}

struct OilmmGrad {          // host results of oilmm_grad_core (partial sums over the latent shard)
  double value = 0.0;
  std::vector<double> gs2;    // one per noise block
  std::vector<double> gS, gU;
  std::vector<lmm_gp_grad_t> ggps;
};

// Value and gradient of the OILMM logpdf (reference src/oilmm.jl:79-113 differentiated) over N points in NB.nblk consecutive
// blocks, block b carrying observation noise NB.s2[b] (one block: the plain logpdf; several: the joint density of the
// conditioning batches and the test points that the predictive logpdf is the difference of).  Per latent: factor, alpha = Kt^-1 delta, Kt^-1 = L^-T L^-1
// (triangular solve of identity riders + an upper-triangular SYRK on the MFMA kernels), one fused contraction kernel; the chain
// rule through T = S^-1/2 U', the projected noise s2/S and the regulariser is small host algebra.
// xd: d x N (device), yd: N x p column-major (device), gy_dev: N x p device output or nullptr.  Caller holds g_mu.
int oilmm_grad_core(const double* xd, int d, int N, const NoiseBlocks& NB, const double* yd, int p, const double* U, const double* S,
                    int m, const lmm_gp_t* gps, int l0, int l1, int with_regulariser, OilmmGrad& G, double* gy_dev) {
  hipStream_t st0 = g.streams[0];
  const int ms = l1 - l0, n = N, nblk = NB.nblk;
  const bool two = nblk > 1;
  const int nsplit = nblk == 2 ? NB.off[1] : N;        // the contraction kernel splits its trace / alpha.alpha sums once
  std::vector<double> T, STa, H;
  project_orthogonal(U, S, p, m, NB.s2[0], T, STa, H);
  std::vector<std::vector<double>> ST(nblk, std::vector<double>(m));      // projected noise s2[b] / S[l]
  for (int b = 0; b < nblk; ++b)
    for (int l = 0; l < m; ++l) ST[b][l] = NB.s2[b] / S[l];
  Uploaded Td(T, st0);
  std::vector<double> means(m);
  for (int l = 0; l < m; ++l) means[l] = gps[l].mean;
  Uploaded meansd(means, st0);
  // projections: Ty (all m, for dS), delta for the shard
  Buf<double> Ty((size_t)n * m), delta((size_t)n * std::max(ms, 1));
  project_on_device(yd, n, p, Td.buf, m, 0, m, nullptr, Ty.p, st0);
  if (ms > 0) project_on_device(yd, n, p, Td.buf, m, l0, ms, meansd.buf.p + l0, delta.p, st0);
  Buf<double> nv(two ? (size_t)n * std::max(ms, 1) : 1);          // per-point projected noise of the shard's latents
  if (two)
    for (int k = 0; k < ms; ++k)
      for (int b = 0; b < nblk; ++b) launch_fill(nv.p + (size_t)k * n + NB.off[b], NB.count(b), ST[b][l0 + k], st0);
  Dims D(n, 1);
  int nb_per = 1, nslots = 1;
  batch_plan(std::max(ms, 1), &nb_per, &nslots, 2.0 * mat_bytes((double)D.elems()));     // factor + inverse-factor matrices
  std::vector<std::vector<Buf<double>>> Am(nslots), Wm(nslots), Rm(nslots);
  std::vector<Buf<double>> part;
  for (int s = 0; s < nslots; ++s) {
    for (int j = 0; j < nb_per; ++j) {
      Am[s].emplace_back(mat_count(D.elems())); Wm[s].emplace_back(mat_count((size_t)(D.NC / 64) * 4096));
      Rm[s].emplace_back(mat_count((size_t)D.ld * D.NC));
    }
    part.emplace_back((size_t)grad_partials(n));
  }
  const int NGR = LMM_NGRAD;
  Buf<double> alpha((size_t)D.NC * std::max(ms, 1)), lmld(std::max(ms, 1)), red((size_t)NGR * std::max(ms, 1));
  // more than two noise blocks: [tr Kinv, alpha.alpha] per (latent, block) from the small per-range kernels
  Buf<double> blksum(nblk > 2 ? (size_t)2 * nblk * std::max(ms, 1) : 1);
  Buf<int> info(std::max(ms, 1));
  HIPCHK(hipMemsetAsync(info.p, 0, std::max(ms, 1) * sizeof(int), st0));
  HIPCHK(hipMemsetAsync(alpha.p, 0, (size_t)D.NC * std::max(ms, 1) * sizeof(double), st0));
  fork_slots(nslots);
  int bi = 0;
  for (int k0 = 0; k0 < ms; k0 += nb_per, ++bi) {
    const int s = bi % nslots, nb = std::min(nb_per, ms - k0);
    hipStream_t st = g.streams[s];
    Batch B;
    BatchPtr Rb{}, alb{};
    GramArgs ga[LMM_MAX_BATCH];
    for (int j = 0; j < nb; ++j) {
      const int k = k0 + j;
      const lmm_gp_t& gp = gps[l0 + k];
      GramArgs a{};
      a.A = Am[s][j].p; a.ld = D.ld; a.nrows = D.NR; a.ncols = D.NC; a.x = xd; a.d = d; a.n = n;
      a.kind = gp.kind; a.var = gp.variance; a.inv_ls = 1.0 / gp.lengthscale; a.pad_diag = 1.0;
      a.diag_add = two ? 0.0 : STa[l0 + k];
      a.diag_vec = two ? nv.p + (size_t)k * n : nullptr;
      a.rider = delta.p + (size_t)k * n; a.rider_ld = n; a.nrider = 1;
      ga[j] = a;
      B.add(Am[s][j].p, Wm[s][j].p, info.p + k);
      Rb.p[j] = Rm[s][j].p; alb.p[j] = alpha.p + (size_t)k * D.NC;
    }
    gram_batch_g(ga, nb, st);
    potrf_batch(B, D.ld, D.NR, D.NC, n, st, D.NC + 1);        // one rider row (delta); rows NC + 1 .. NR - 1 are zero padding
    launch_lml_reduce(B.A, nb, D.ld, n, D.NC, 1, lmld.p + k0, st);
    for (int j = 0; j < nb; ++j) {
      launch_extract_row(Am[s][j].p, D.ld, D.NC, n, alb.p[j], st);
      launch_set_identity(Rm[s][j].p, D.ld, D.NC, st);
    }
    launch_backsolve(B.A, D.ld, B.W, D.NC / 64, alb, nb, st);
    trsm_rec(Rb, D.ld, D.NC, B.A, D.ld, B.W, nb, 0, D.NC, st, true);        // R = L^-T (upper triangular), whole batch
    launch_syrk_upper_set(B.A, D.ld, Rb, D.ld, D.NC, nb, st);                // lower(A) = L^-T L^-1 = Kt^-1
    for (int j = 0; j < nb; ++j) {
      const int k = k0 + j;
      launch_grad_reduce(Am[s][j].p, D.ld, n, nsplit, alb.p[j], delta.p + (size_t)k * n, xd, d, to_dev(gps[l0 + k]), part[s].p,
                         red.p + (size_t)NGR * k, st);
      if (nblk > 2)
        for (int b = 0; b < nblk; ++b) {
          double* o = blksum.p + ((size_t)k * nblk + b) * 2;
          launch_block_trace(Am[s][j].p, D.ld, n, 1, NB.off[b], NB.off[b + 1], o, st);
          launch_atb(alb.p[j] + NB.off[b], n, alb.p[j] + NB.off[b], n, NB.count(b), 1, 1, o + 1, st);
        }
    }
  }
  join_slots(nslots);
  std::vector<double> lml(std::max(ms, 1), 0.0), hred((size_t)NGR * std::max(ms, 1), 0.0);
  std::vector<int> hinfo(std::max(ms, 1), 0);
  std::vector<double> hblk(nblk > 2 ? (size_t)2 * nblk * std::max(ms, 1) : 0, 0.0);
  HIPCHK(hipMemcpyAsync(lml.data(), lmld.p, std::max(ms, 1) * sizeof(double), hipMemcpyDeviceToHost, st0));
  HIPCHK(hipMemcpyAsync(hred.data(), red.p, (size_t)NGR * std::max(ms, 1) * sizeof(double), hipMemcpyDeviceToHost, st0));
  if (!hblk.empty() && ms > 0) HIPCHK(hipMemcpyAsync(hblk.data(), blksum.p, hblk.size() * sizeof(double), hipMemcpyDeviceToHost, st0));
  HIPCHK(hipMemcpyAsync(hinfo.data(), info.p, std::max(ms, 1) * sizeof(int), hipMemcpyDeviceToHost, st0));
  // small dense products needed by the chain rule: YA = Y' alpha (p x ms), aTy = alpha_l . (T y)_l, M2 = Y Y' (p x p) per noise block
  const size_t pp = (size_t)p * p;
  Buf<double> YAd((size_t)p * std::max(ms, 1)), aTyd((size_t)std::max(ms, 1) * m), M2d(pp * nblk);
  if (ms > 0) {
    launch_atb(yd, n, alpha.p, D.NC, n, p, ms, YAd.p, st0);
    launch_atb(alpha.p, D.NC, Ty.p, n, n, ms, m, aTyd.p, st0);       // [k, l]; only l = l0 + k is used
  }
  std::vector<double> YA((size_t)p * std::max(ms, 1), 0.0), aTy((size_t)std::max(ms, 1) * m, 0.0), M2all(pp * nblk, 0.0);
  if (with_regulariser) {
    for (int b = 0; b < nblk; ++b) launch_atb(yd + NB.off[b], n, yd + NB.off[b], n, NB.count(b), p, p, M2d.p + pp * b, st0);
    HIPCHK(hipMemcpyAsync(M2all.data(), M2d.p, M2all.size() * sizeof(double), hipMemcpyDeviceToHost, st0));
  }
  HIPCHK(hipMemcpyAsync(YA.data(), YAd.p, YA.size() * sizeof(double), hipMemcpyDeviceToHost, st0));
  HIPCHK(hipMemcpyAsync(aTy.data(), aTyd.p, aTy.size() * sizeof(double), hipMemcpyDeviceToHost, st0));
  HIPCHK(hipStreamSynchronize(st0));
  if (int rc = check_info(hinfo, l0)) return rc;

  // ---- host chain rule ----
  double total = 0.0;
  G.gs2.assign(nblk, 0.0);
  G.gS.assign(m, 0.0); G.gU.assign((size_t)p * m, 0.0);
  G.ggps.assign(m, lmm_gp_grad_t{0.0, 0.0, 0.0});
  for (int k = 0; k < ms; ++k) {
    const int l = l0 + k;
    total += lml[k];
    const double* r = &hred[(size_t)NGR * k];
    const double cl = r[0], ad = r[3], sa = r[4], v = gps[l].variance;
    double D_aa = 0.0, D_tr = 0.0, g_s2 = 0.0;       // a'Da, tr(Kt^-1 D) with D the projected noise; sum_b s2[b] dlml/dnoise_b
    for (int b = 0; b < nblk; ++b) {
      // tr Kinv and alpha.alpha over the block's rows
      const double tr = nblk > 2 ? hblk[((size_t)k * nblk + b) * 2] : r[b ? 5 : 1];
      const double aa = nblk > 2 ? hblk[((size_t)k * nblk + b) * 2 + 1] : r[b ? 6 : 2];
      const double gb = 0.5 * (aa - tr);                                       // d lml / d (projected noise of block b)
      D_aa += ST[b][l] * aa; D_tr += ST[b][l] * tr;
      G.gs2[b] += gb / S[l];
      g_s2 += gb * NB.s2[b];
    }
    // 1/2 tr((aa' - Kt^-1) K) / v  with K = Kt - D:  a'delta - a'Da - (n - tr(Kt^-1 D))
    G.ggps[l].variance = 0.5 * ((ad - D_aa) - ((double)n - D_tr)) / v;
    G.ggps[l].lengthscale = cl;                                                // sum_{i>j} (a_i a_j - Kinv_ij) dK_ij/dl (x2 / 2)
    G.ggps[l].mean = sa;
    G.gS[l] += -g_s2 / (S[l] * S[l]) + 0.5 * aTy[k + (size_t)l * ms] / S[l];
    for (int o = 0; o < p; ++o) G.gU[o + (size_t)l * p] += -YA[o + (size_t)k * p] / std::sqrt(S[l]);
  }
  std::vector<double> PtP;       // P'P for the regulariser's dY
  if (with_regulariser) {
    std::vector<double> Pm((size_t)p * p, 0.0);
    for (int a1 = 0; a1 < p; ++a1)
      for (int b1 = 0; b1 < p; ++b1) {
        double s = (a1 == b1) ? 1.0 : 0.0;
        for (int l = 0; l < m; ++l) s -= U[a1 + (size_t)l * p] * U[b1 + (size_t)l * p];
        Pm[a1 + (size_t)b1 * p] = s;
      }
    auto matmul = [&](const std::vector<double>& A1, int r, int c, const std::vector<double>& B1, int c2) {
      std::vector<double> Cc((size_t)r * c2, 0.0);
      for (int j = 0; j < c2; ++j) for (int kk = 0; kk < c; ++kk) { const double b = B1[kk + (size_t)j * c]; for (int i = 0; i < r; ++i) Cc[i + (size_t)j * r] += A1[i + (size_t)kk * r] * b; }
      return Cc;
    };
    PtP = matmul(Pm, p, p, Pm, p);                               // P symmetric: P'P = P P
    double logdetS = 0.0;
    for (int l = 0; l < m; ++l) logdetS += std::log(S[l]);
    std::vector<double> Uv(U, U + (size_t)p * m);
    std::vector<double> PU = matmul(Pm, p, p, Uv, m);
    for (int blk = 0; blk < nblk; ++blk) {                        // reference src/oilmm.jl:101-113, once per noise block
      const std::vector<double> M2(M2all.begin() + pp * blk, M2all.begin() + pp * (blk + 1));
      const double s2 = NB.s2[blk], cnt = NB.count(blk);
      double Rn = 0.0;                                           // |P Y|_F^2 = tr(P'P Y Y')
      for (int a1 = 0; a1 < p; ++a1) for (int b1 = 0; b1 < p; ++b1) Rn += PtP[a1 + (size_t)b1 * p] * M2[b1 + (size_t)a1 * p];
      total += -(cnt * (logdetS + (double)(p - m) * std::log(2.0 * M_PI * s2)) + Rn / s2) / 2.0;
      for (int l = 0; l < m; ++l) G.gS[l] += -cnt / (2.0 * S[l]);
      G.gs2[blk] += -0.5 * (cnt * (double)(p - m) / s2 - Rn / (s2 * s2));
      std::vector<double> M2U = matmul(M2, p, p, Uv, m);
      std::vector<double> t1 = matmul(Pm, p, p, M2U, m), t2 = matmul(M2, p, p, PU, m);
      for (size_t q = 0; q < G.gU.size(); ++q) G.gU[q] += (t1[q] + t2[q]) / s2;
    }
  }
  G.value = total;
  if (gy_dev) {
    // dL/dY[o, i] = - sum_l T[l, o] alpha_l[i]  - (P'P Y)[o, i] / sigma2(i)
    std::vector<double> negTt((size_t)p * std::max(ms, 1), 0.0);
    for (int k = 0; k < ms; ++k) for (int o = 0; o < p; ++o) negTt[o + (size_t)k * p] = -T[(l0 + k) + (size_t)o * m];
    Uploaded nT(negTt, st0);
    Buf<double> ga((size_t)n * p);
    // mix reads lat[l*ns + s] with ns = n: alpha is stored with stride NC -> compact copy first
    Buf<double> ac((size_t)n * std::max(ms, 1));
    for (int k = 0; k < ms; ++k) HIPCHK(hipMemcpyAsync(ac.p + (size_t)k * n, alpha.p + (size_t)k * D.NC, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, st0));
    launch_mix(ac.p, n, ms, nT.buf.p, p, 1, 0.0, 0.0, nullptr, 0.0, with_regulariser ? ga.p : gy_dev, st0);
    if (with_regulariser) {
      Uploaded Qd(PtP, st0);
      Buf<double> gr((size_t)n * p);
      launch_tall_skinny(yd, n, n, p, Qd.buf.p, p, p, gr.p, n, nullptr, nullptr, 0, nullptr, 0, st0);   // (Y' (P'P)')' rows
      launch_vec_lin_blocks(ga.p, gr.p, NB, -1.0, n, (size_t)n * p, gy_dev, st0);
      HIPCHK(hipStreamSynchronize(st0));       // ga, gr, Qd are released on return
    } else {
      HIPCHK(hipStreamSynchronize(st0));
    }
  }
  return LMM_OK;
}

void write_oilmm_grad(const OilmmGrad& G, int m, int p, double* out_logpdf, double* grad_sigma2, double* grad_S, double* grad_U,
                      lmm_gp_grad_t* grad_gps) {
  *out_logpdf = G.value;
  if (grad_sigma2) { *grad_sigma2 = 0.0; for (double v : G.gs2) *grad_sigma2 += v; }
  if (grad_S) std::copy(G.gS.begin(), G.gS.end(), grad_S);
  if (grad_U) std::copy(G.gU.begin(), G.gU.end(), grad_U);
  if (grad_gps) std::copy(G.ggps.begin(), G.ggps.end(), grad_gps);
  (void)m; (void)p;
}

}  // namespace

extern "C" {

// Value and gradient of logpdf(fx::FiniteGP{<:OILMM}, y) (reference src/oilmm.jl:79-93; what the reference's
// Zygote.gradient(logpdf, fx, y) differentiates, test/oilmm.jl:31-32) w.r.t. y, sigma2, S, U and every latent's
// (variance, lengthscale, mean).  Partial sums over the shard.
int lmm_oilmm_logpdf_grad(const double* x, int d, int n, const double* y, int p, const double* U, const double* S, int m,
                          double sigma2, const lmm_gp_t* gps, int latent_begin, int latent_end, int with_regulariser,
                          double* out_logpdf, double* grad_y, double* grad_sigma2, double* grad_S, double* grad_U,
                          lmm_gp_grad_t* grad_gps) {
  std::lock_guard<std::mutex> lk(g_mu);
  REQUIRE_INIT();
  LMM_TRY
  // fp32 compute mode (round 3): Float32 factor, triangular inverse and K^-1 = L^-T L^-1 (all on v_mfma_f32), Float64 reductions;
  // stated tolerance against the Float64 gradient: include/lmm_hip.h
  if (!x || !y || !U || !S || !out_logpdf || d <= 0 || n <= 0 || p <= 0 || m <= 0) return fail(LMM_ERR_ARG, "bad arguments");
  if (m > p) return fail(LMM_ERR_DIM, "out dim of x != out dim of f.");
  if (latent_begin < 0 || latent_end > m || latent_begin > latent_end) return fail(LMM_ERR_ARG, "bad latent shard");
  if (int rc = check_gps(gps, m)) return rc;
  if (!(sigma2 > 0.0)) return fail(LMM_ERR_ARG, "sigma2 must be > 0");
  hipStream_t st0 = g.streams[0];
  DevIn xd(x, (size_t)d * n, st0), yd(y, (size_t)n * p, st0);
  DevOut gy(grad_y, (size_t)n * p);
  OilmmGrad G;
  if (int rc = oilmm_grad_core(xd.p, d, n, one_noise_block(n, sigma2), yd.p, p, U, S, m, gps, latent_begin, latent_end, with_regulariser,
                               G, gy.p))
    return rc;
  write_oilmm_grad(G, m, p, out_logpdf, grad_sigma2, grad_S, grad_U, grad_gps);
  if (grad_y) { gy.finish(st0); HIPCHK(hipStreamSynchronize(st0)); }
  return LMM_OK;
  LMM_CATCH
}

// Value and gradient of the predictive logpdf  logpdf(posterior(f(x, sigma2), y)(xs, sigma2_s), ys)  of an OILMM (or, with
// U = I, S = 1, an IndependentMOGP) -- the reference takes Zygote.gradient(logpdf, po_x, y*) on the posterior models
// (test/oilmm.jl:32, test/independent_mogp.jl:66).  Exact conditioning gives, latent by latent and for the regulariser,
//     log p(ys | y) = log p(y, ys) - log p(y),
// so value and TOTAL derivatives (through alpha, the factor and the Schur complement of the posterior) are the difference of
// two evaluations of the prior-logpdf gradient: the joint over [x; xs] with per-block noise, and the marginal over x.
// _seq: the posterior was conditioned SEQUENTIALLY, posterior(posterior(f(x1, s1), y1)(x2, s2), y2) ... (reference
// src/oilmm.jl:116-134 applied to its own result); exact conditioning makes that the posterior given all batches at once with
// per-batch noise.  x (d x n) and y (n x p by outputs) hold the batches' points in conditioning order, n = sum batch_n.
int lmm_oilmm_post_logpdf_grad_seq(const double* x, int d, int n, const int* batch_n, const double* batch_sigma2, int nbatch,
                                   const double* y, const double* xs, int ns, const double* ys, int p, const double* U,
                                   const double* S, int m, double sigma2_s, const lmm_gp_t* gps, int latent_begin, int latent_end,
                                   int with_regulariser, double* out_logpdf, double* grad_y, double* grad_ys,
                                   double* grad_batch_sigma2, double* grad_sigma2_s, double* grad_S, double* grad_U,
                                   lmm_gp_grad_t* grad_gps) {
  std::lock_guard<std::mutex> lk(g_mu);
  REQUIRE_INIT();
  LMM_TRY
  // (the same core with one noise block per conditioning batch + one for the test points: served in the fp32 compute mode too)
  if (!x || !y || !xs || !ys || !U || !S || !out_logpdf || d <= 0 || n <= 0 || ns <= 0 || p <= 0 || m <= 0)
    return fail(LMM_ERR_ARG, "bad arguments");
  if (m > p) return fail(LMM_ERR_DIM, "out dim of x != out dim of f.");
  if (latent_begin < 0 || latent_end > m || latent_begin > latent_end) return fail(LMM_ERR_ARG, "bad latent shard");
  if (int rc = check_gps(gps, m)) return rc;
  if (int rc = check_batches(batch_n, batch_sigma2, nbatch, n)) return rc;
  if (!(sigma2_s > 0.0)) return fail(LMM_ERR_ARG, "sigma2 must be > 0");
  hipStream_t st0 = g.streams[0];
  const int N = n + ns;
  DevIn xd(x, (size_t)d * n, st0), yd(y, (size_t)n * p, st0), xsd(xs, (size_t)d * ns, st0), ysd(ys, (size_t)ns * p, st0);
  Buf<double> xj((size_t)d * N), yj((size_t)N * p), gj((size_t)N * p), gm((size_t)n * p);
  HIPCHK(hipMemcpyAsync(xj.p, xd.p, (size_t)d * n * sizeof(double), hipMemcpyDeviceToDevice, st0));
  HIPCHK(hipMemcpyAsync(xj.p + (size_t)d * n, xsd.p, (size_t)d * ns * sizeof(double), hipMemcpyDeviceToDevice, st0));
  HIPCHK(hipMemcpy2DAsync(yj.p, (size_t)N * sizeof(double), yd.p, (size_t)n * sizeof(double), (size_t)n * sizeof(double), p, hipMemcpyDeviceToDevice, st0));
  HIPCHK(hipMemcpy2DAsync(yj.p + n, (size_t)N * sizeof(double), ysd.p, (size_t)ns * sizeof(double), (size_t)ns * sizeof(double), p, hipMemcpyDeviceToDevice, st0));
  const bool want_gy = grad_y != nullptr || grad_ys != nullptr;
  OilmmGrad GJ, GM;
  if (int rc = oilmm_grad_core(xj.p, d, N, batch_noise_blocks(batch_n, batch_sigma2, nbatch, ns, sigma2_s), yj.p, p, U, S, m, gps,
                               latent_begin, latent_end, with_regulariser, GJ, want_gy ? gj.p : nullptr)) return rc;
  if (int rc = oilmm_grad_core(xd.p, d, n, batch_noise_blocks(batch_n, batch_sigma2, nbatch, 0, 0.0), yd.p, p, U, S, m, gps,
                               latent_begin, latent_end, with_regulariser, GM, grad_y ? gm.p : nullptr)) return rc;
  *out_logpdf = GJ.value - GM.value;
  if (grad_batch_sigma2) for (int b = 0; b < nbatch; ++b) grad_batch_sigma2[b] = GJ.gs2[b] - GM.gs2[b];
  if (grad_sigma2_s) *grad_sigma2_s = GJ.gs2[nbatch];
  for (int l = 0; l < m; ++l) {
    if (grad_S) grad_S[l] = GJ.gS[l] - GM.gS[l];
    if (grad_gps) {
      grad_gps[l].variance = GJ.ggps[l].variance - GM.ggps[l].variance;
      grad_gps[l].lengthscale = GJ.ggps[l].lengthscale - GM.ggps[l].lengthscale;
      grad_gps[l].mean = GJ.ggps[l].mean - GM.ggps[l].mean;
    }
  }
  if (grad_U) for (size_t q = 0; q < (size_t)p * m; ++q) grad_U[q] = GJ.gU[q] - GM.gU[q];
  if (grad_y) {
    DevOut gy(grad_y, (size_t)n * p);
    Buf<double> top((size_t)n * p);
    HIPCHK(hipMemcpy2DAsync(top.p, (size_t)n * sizeof(double), gj.p, (size_t)N * sizeof(double), (size_t)n * sizeof(double), p, hipMemcpyDeviceToDevice, st0));
    launch_vec_lin(top.p, gm.p, -1.0, n * p, gy.p, st0);
    gy.finish(st0);
    HIPCHK(hipStreamSynchronize(st0));
  }
  if (grad_ys) {
    DevOut gys(grad_ys, (size_t)ns * p);
    HIPCHK(hipMemcpy2DAsync(gys.p, (size_t)ns * sizeof(double), gj.p + n, (size_t)N * sizeof(double), (size_t)ns * sizeof(double), p, hipMemcpyDeviceToDevice, st0));
    gys.finish(st0);
    HIPCHK(hipStreamSynchronize(st0));
  }
  return LMM_OK;
  LMM_CATCH
}

// One conditioning batch: posterior(f(x, sigma2), y).
int lmm_oilmm_post_logpdf_grad(const double* x, int d, int n, const double* y, const double* xs, int ns, const double* ys, int p,
                               const double* U, const double* S, int m, double sigma2, double sigma2_s, const lmm_gp_t* gps,
                               int latent_begin, int latent_end, int with_regulariser, double* out_logpdf, double* grad_y,
                               double* grad_ys, double* grad_sigma2, double* grad_sigma2_s, double* grad_S, double* grad_U,
                               lmm_gp_grad_t* grad_gps) {
  return lmm_oilmm_post_logpdf_grad_seq(x, d, n, &n, &sigma2, 1, y, xs, ns, ys, p, U, S, m, sigma2_s, gps, latent_begin, latent_end,
                                        with_regulariser, out_logpdf, grad_y, grad_ys, grad_sigma2, grad_sigma2_s, grad_S, grad_U, grad_gps);
}

// logpdf(fx, Y::AbstractMatrix) -- one logpdf per column of Y with ONE factorisation per latent (SURVEY.md 8f next #3;
// the reference answers it through AbstractGPs' dense generic fallback).  Y is (n p) x ncol column-major.
int lmm_oilmm_logpdf_multi(const double* x, int d, int n, const double* Y, int p, int ncol, const double* U, const double* S,
                           int m, double sigma2, const lmm_gp_t* gps, int latent_begin, int latent_end,
                           int with_regulariser, double* out) {
  std::lock_guard<std::mutex> lk(g_mu);
  REQUIRE_INIT();
  LMM_TRY
  if (!x || !Y || !U || !S || !out || d <= 0 || n <= 0 || p <= 0 || m <= 0 || ncol <= 0) return fail(LMM_ERR_ARG, "bad arguments");
  if (m > p) return fail(LMM_ERR_DIM, "out dim of x != out dim of f.");
  if (latent_begin < 0 || latent_end > m || latent_begin > latent_end) return fail(LMM_ERR_ARG, "bad latent shard");
  if (int rc = check_gps(gps, m)) return rc;
  hipStream_t st0 = g.streams[0];
  std::vector<double> T, ST, H;
  project_orthogonal(U, S, p, m, sigma2, T, ST, H);
  DevIn xd(x, (size_t)d * n, st0), yd(Y, (size_t)n * p * ncol, st0);
  Uploaded Td(T, st0), Hd(H, st0);
  std::vector<double> means(m);
  for (int l = 0; l < m; ++l) means[l] = gps[l].mean;
  Uploaded meansd(means, st0);
  const int l0 = latent_begin, l1 = latent_end, ms = l1 - l0;
  Buf<double> delta((size_t)n * std::max(ms, 1) * ncol), Ty((size_t)n * m), partial(tall_skinny_partials(n, p)), resid_dev(ncol);
  std::vector<double> resid(ncol, 0.0);
  for (int c = 0; c < ncol; ++c) {
    const double* yc = yd.p + (size_t)c * n * p;
    // rider [latent k][column c]: delta + (k ncol + c) n  ==  output column stride ncol*n
    if (ms > 0) launch_tall_skinny(yc, n, n, p, Td.buf.p + l0, m, ms, delta.p + (size_t)c * n, ncol * n, meansd.buf.p + l0, nullptr, 0,
                                   nullptr, 0, st0);
    if (with_regulariser) {
      project_on_device(yc, n, p, Td.buf, m, 0, m, nullptr, Ty.p, st0);
      residual_on_device(yc, n, p, Ty.p, m, Hd.buf, partial.p, resid_dev.p + c, st0);
    }
  }
  if (with_regulariser) HIPCHK(hipMemcpyAsync(resid.data(), resid_dev.p, ncol * sizeof(double), hipMemcpyDeviceToHost, st0));
  std::vector<double> lml;
  if (int rc = latent_lmls(xd.p, d, n, gps, ST.data(), l0, l1, delta.p, lml, ncol)) return rc;
  double logdetS = 0.0;
  for (int l = 0; l < m; ++l) logdetS += std::log(S[l]);
  for (int c = 0; c < ncol; ++c) {
    double total = 0.0;
    for (int k = 0; k < ms; ++k) total += lml[(size_t)k * ncol + c];
    if (with_regulariser) total += -((double)n * (logdetS + (double)(p - m) * std::log(2.0 * M_PI * sigma2)) + resid[c] / sigma2) / 2.0;
    out[c] = total;
  }
  return LMM_OK;
  LMM_CATCH
}

// MOInputIsotopicByFeatures <-> MOInputIsotopicByOutputs reordering of a length n*p vector (reference
// src/independent_mogp.jl:135-159): to_outputs != 0: out[o n + i] = in[i p + o]; else the inverse.
int lmm_reorder(const double* in, int n, int p, int to_outputs, double* out) {
  std::lock_guard<std::mutex> lk(g_mu);
  REQUIRE_INIT();
  LMM_TRY
  if (!in || !out || n <= 0 || p <= 0) return fail(LMM_ERR_ARG, "bad arguments");
  hipStream_t st0 = g.streams[0];
  DevIn ind(in, (size_t)n * p, st0);
  DevOut od(out, (size_t)n * p);
  launch_reorder(ind.p, n, p, to_outputs, od.p, st0);
  od.finish(st0);
  HIPCHK(hipStreamSynchronize(st0));
  return LMM_OK;
  LMM_CATCH
}

int lmm_mogp_logpdf(const double* x, int d, int n, const double* y, int m, double sigma2, const lmm_gp_t* gps,
                    int latent_begin, int latent_end, double* out) {
  std::lock_guard<std::mutex> lk(g_mu);
  REQUIRE_INIT();
  LMM_TRY
  if (!x || !y || !out || d <= 0 || n <= 0 || m <= 0) return fail(LMM_ERR_ARG, "bad arguments");
  if (latent_begin < 0 || latent_end > m || latent_begin > latent_end) return fail(LMM_ERR_ARG, "bad latent shard");
  if (int rc = check_gps(gps, m)) return rc;
  hipStream_t st0 = g.streams[0];
  const int l0 = latent_begin, l1 = latent_end, ms = l1 - l0;
  DevIn xd(x, (size_t)d * n, st0), yd(y, (size_t)n * m, st0);
  // delta_l = y_l - mean_l  via the projection kernel with T = I restricted to the shard
  std::vector<double> T((size_t)m * m, 0.0), means(m), noise(m, sigma2);
  for (int l = 0; l < m; ++l) { T[l + (size_t)l * m] = 1.0; means[l] = gps[l].mean; }
  Uploaded Td(T, st0), meansd(means, st0);
  Buf<double> delta((size_t)n * std::max(ms, 1));
  if (ms > 0) project_on_device(yd.p, n, m, Td.buf, m, l0, ms, meansd.buf.p + l0, delta.p, st0);
  std::vector<double> lml;
  if (int rc = latent_lmls(xd.p, d, n, gps, noise.data(), l0, l1, delta.p, lml)) return rc;
  double total = 0.0;
  for (int k = 0; k < ms; ++k) total += lml[k];
  *out = total;
  return LMM_OK;
  LMM_CATCH
}

int lmm_mogp_logpdf_diag(const double* x, int d, int n, const double* y, int m, const double* noise_diag, const lmm_gp_t* gps,
                         int latent_begin, int latent_end, double* out) {
  std::lock_guard<std::mutex> lk(g_mu);
  REQUIRE_INIT();
  LMM_TRY
  if (!x || !y || !noise_diag || !out || d <= 0 || n <= 0 || m <= 0) return fail(LMM_ERR_ARG, "bad arguments");
  if (latent_begin < 0 || latent_end > m || latent_begin > latent_end) return fail(LMM_ERR_ARG, "bad latent shard");
  if (int rc = check_gps(gps, m)) return rc;
  hipStream_t st0 = g.streams[0];
  const int l0 = latent_begin, l1 = latent_end, ms = l1 - l0;
  DevIn xd(x, (size_t)d * n, st0), yd(y, (size_t)n * m, st0), nd(noise_diag, (size_t)n * m, st0);
  std::vector<double> T((size_t)m * m, 0.0), means(m), noise(m, 0.0);
  for (int l = 0; l < m; ++l) { T[l + (size_t)l * m] = 1.0; means[l] = gps[l].mean; }
  Uploaded Td(T, st0), meansd(means, st0);
  Buf<double> delta((size_t)n * std::max(ms, 1));
  if (ms > 0) project_on_device(yd.p, n, m, Td.buf, m, l0, ms, meansd.buf.p + l0, delta.p, st0);
  std::vector<double> lml;
  if (int rc = latent_lmls(xd.p, d, n, gps, noise.data(), l0, l1, delta.p, lml, 1, nd.p + (size_t)l0 * n)) return rc;
  double total = 0.0;
  for (int k = 0; k < ms; ++k) total += lml[k];
  *out = total;
  return LMM_OK;
  LMM_CATCH
}

int lmm_ilmm_logpdf_ex(const double* x, int d, int n, const double* y, int p, const double* H, int m, double sigma2,
                       const lmm_gp_t* gps, const lmm_jitters_t* jit, int allow_decoupled, int* path_used, double* out) {
  std::lock_guard<std::mutex> lk(g_mu);
  REQUIRE_INIT();
  LMM_TRY
  if (!x || !y || !H || !out || d <= 0 || n <= 0 || p <= 0 || m <= 0) return fail(LMM_ERR_ARG, "bad arguments");
  if (int rc = check_gps(gps, m)) return rc;
  if (!jit) jit = &kDefaultJit;
  if ((long long)m * n > 2000000000LL / 64) return fail(LMM_ERR_UNSUPPORTED, "m*n too large for the dense path");
  hipStream_t st0 = g.streams[0];
  std::vector<double> T, ST;
  double logdetST = 0.0;
  if (int rc = project_dense(H, p, m, sigma2, jit->project_jitter, T, ST, &logdetST)) return rc;
  DevIn xd(x, (size_t)d * n, st0), yd(y, (size_t)n * p, st0);
  Uploaded Td(T, st0), STd(ST, st0);
  std::vector<double> Hv(H, H + (size_t)p * m), means(m);
  std::vector<LatentDev> lat(m);
  for (int l = 0; l < m; ++l) { means[l] = gps[l].mean; lat[l] = to_dev(gps[l]); }
  Uploaded Hd(Hv, st0), meansd(means, st0);
  Buf<LatentDev> latd(m);
  HIPCHK(hipMemcpyAsync(latd.p, lat.data(), m * sizeof(LatentDev), hipMemcpyHostToDevice, st0));
  // projection, residual, rider = vec(T Y) - mean
  Buf<double> Ty((size_t)n * m), delta((size_t)n * m), partial(tall_skinny_partials(n, p)), resid_dev(1);
  project_on_device(yd.p, n, p, Td.buf, m, 0, m, nullptr, Ty.p, st0);
  residual_on_device(yd.p, n, p, Ty.p, m, Hd.buf, partial.p, resid_dev.p, st0);
  // reference src/ilmm.jl:171-181 (scalar part; the residual comes from the device below)
  auto regulariser = [&](double resid) {
    return -((double)n * ((double)(p - m) * kLog2Pi + ((double)p * std::log(sigma2) - logdetST)) + resid / sigma2) / 2.0;
  };
  bool identical = allow_decoupled != 0;
  for (int l = 1; l < m && identical; ++l)
    identical = gps[l].kind == gps[0].kind && gps[l].variance == gps[0].variance && gps[l].lengthscale == gps[0].lengthscale;
  if (path_used) *path_used = identical ? 1 : 0;
  if (identical) {
    // Decoupled shortcut: with one shared latent kernel the covariance is I (x) K + SigmaT (x) I;
    // SigmaT = Q Lam Q' rotates it to blockdiag(K + lam_a I), so the (mn)^3/3 factorisation becomes m independent n^3/3
    // ones on the rotated projections (Q'T) Y - Q' mu.  Same value up to rounding; not the reference's operation count.
    std::vector<double> lam, Q;
    host_jacobi_eig(ST, m, lam, Q);
    std::vector<double> T2((size_t)m * p, 0.0), mu2(m, 0.0);
    for (int aI = 0; aI < m; ++aI) {
      for (int o = 0; o < p; ++o) {
        double s = 0.0;
        for (int b = 0; b < m; ++b) s += Q[b + (size_t)aI * m] * T[b + (size_t)o * m];
        T2[aI + (size_t)o * m] = s;
      }
      for (int b = 0; b < m; ++b) mu2[aI] += Q[b + (size_t)aI * m] * gps[b].mean;
      if (!(lam[aI] > 0.0)) return fail(LMM_ERR_NOT_PD, "PosDefException: SigmaT has a non-positive eigenvalue");
    }
    Uploaded T2d(T2, st0), mu2d(mu2, st0);
    project_on_device(yd.p, n, p, T2d.buf, m, 0, m, mu2d.buf.p, delta.p, st0);
    std::vector<lmm_gp_t> g2(m, gps[0]);
    std::vector<double> lml;
    double resid = 0.0;
    HIPCHK(hipMemcpyAsync(&resid, resid_dev.p, sizeof(double), hipMemcpyDeviceToHost, st0));
    if (int rc = latent_lmls(xd.p, d, n, g2.data(), lam.data(), 0, m, delta.p, lml)) return rc;   // synchronises st0
    double total = 0.0;
    for (int l = 0; l < m; ++l) total += lml[l];
    *out = total + regulariser(resid);
    return LMM_OK;
  }
  project_on_device(yd.p, n, p, Td.buf, m, 0, m, meansd.buf.p, delta.p, st0);
  // one dense (mn) x (mn) factorisation: reference src/ilmm.jl:160-162 (fp32 compute mode: a Float32 matrix, as the per-latent paths)
  const int N = m * n;
  Dims D(N, 1);
  Buf<double> A(mat_count(D.elems())), W(mat_count((size_t)(D.NC / 64) * 4096)), lml_dev(1);
  Buf<int> info(1);
  HIPCHK(hipMemsetAsync(info.p, 0, sizeof(int), st0));
  DenseArgs a{};
  a.A = A.p; a.ld = D.ld; a.nrows = D.NR; a.ncols = D.NC; a.x = xd.p; a.d = d; a.n = n; a.m = m;
  a.lat = latd.p; a.sigmaT = STd.buf.p; a.rider = delta.p; a.rider_ld = N; a.nrider = 1;
  launch_dense_assemble(a, st0);
  potrf_rec(A.p, D.ld, D.NR, 0, D.NC, W.p, N, info.p, st0);
  launch_lml_reduce(A.p, D.ld, N, D.NC, 1, lml_dev.p, st0);
  double lml = 0.0, resid = 0.0;
  int hinfo = 0;
  HIPCHK(hipMemcpyAsync(&lml, lml_dev.p, sizeof(double), hipMemcpyDeviceToHost, st0));
  HIPCHK(hipMemcpyAsync(&resid, resid_dev.p, sizeof(double), hipMemcpyDeviceToHost, st0));
  HIPCHK(hipMemcpyAsync(&hinfo, info.p, sizeof(int), hipMemcpyDeviceToHost, st0));
  HIPCHK(hipStreamSynchronize(st0));
  if (int rc = check_info(std::vector<int>{hinfo}, 0)) return rc;
  *out = lml + regulariser(resid);
  return LMM_OK;
  LMM_CATCH
}

int lmm_ilmm_logpdf(const double* x, int d, int n, const double* y, int p, const double* H, int m, double sigma2,
                    const lmm_gp_t* gps, const lmm_jitters_t* jit, double* out) {
  return lmm_ilmm_logpdf_ex(x, d, n, y, p, H, m, sigma2, gps, jit, 1, nullptr, out);
}

// logpdf(fx::FiniteGP{<:ILMM}, Y::AbstractMatrix), dense H: one value per column of Y ((n p) x ncol) from ONE (mn) x (mn)
// factorisation -- the columns ride it as rider rows (AbstractGPs.TestUtils calls logpdf(fx, Y) on ilmmx, reference
// test/ilmm.jl:34-37; the reference answers through the generic dense fallback, one factorisation per call all the same).
int lmm_ilmm_logpdf_multi(const double* x, int d, int n, const double* Y, int p, int ncol, const double* H, int m, double sigma2,
                          const lmm_gp_t* gps, const lmm_jitters_t* jit, double* out) {
  std::lock_guard<std::mutex> lk(g_mu);
  REQUIRE_INIT();
  LMM_TRY
  if (!x || !Y || !H || !out || d <= 0 || n <= 0 || p <= 0 || m <= 0 || ncol <= 0) return fail(LMM_ERR_ARG, "bad arguments");
  if (int rc = check_gps(gps, m)) return rc;
  if (!(sigma2 > 0.0)) return fail(LMM_ERR_ARG, "sigma2 must be > 0");
  if (!jit) jit = &kDefaultJit;
  if ((long long)m * n > 2000000000LL / 64) return fail(LMM_ERR_UNSUPPORTED, "m*n too large for the dense path");
  hipStream_t st0 = g.streams[0];
  std::vector<double> T, ST;
  double logdetST = 0.0;
  if (int rc = project_dense(H, p, m, sigma2, jit->project_jitter, T, ST, &logdetST)) return rc;
  DevIn xd(x, (size_t)d * n, st0), yd(Y, (size_t)n * p * ncol, st0);
  Uploaded Td(T, st0), STd(ST, st0);
  std::vector<double> Hv(H, H + (size_t)p * m), means(m);
  std::vector<LatentDev> lat(m);
  for (int l = 0; l < m; ++l) { means[l] = gps[l].mean; lat[l] = to_dev(gps[l]); }
  Uploaded Hd(Hv, st0), meansd(means, st0);
  Buf<LatentDev> latd(m);
  HIPCHK(hipMemcpyAsync(latd.p, lat.data(), m * sizeof(LatentDev), hipMemcpyHostToDevice, st0));
  const int N = m * n;
  Buf<double> Ty((size_t)N), delta((size_t)N * ncol), partial(tall_skinny_partials(n, p)), resid_dev(ncol), lml_dev(ncol);
  for (int c = 0; c < ncol; ++c) {
    const double* yc = yd.p + (size_t)c * n * p;
    project_on_device(yc, n, p, Td.buf, m, 0, m, nullptr, Ty.p, st0);
    residual_on_device(yc, n, p, Ty.p, m, Hd.buf, partial.p, resid_dev.p + c, st0);
    project_on_device(yc, n, p, Td.buf, m, 0, m, meansd.buf.p, delta.p + (size_t)c * N, st0);      // rider c: [latent][point]
  }
  Dims D(N, ncol);
  Buf<double> A(mat_count(D.elems())), W(mat_count((size_t)(D.NC / 64) * 4096));
  Buf<int> info(1);
  HIPCHK(hipMemsetAsync(info.p, 0, sizeof(int), st0));
  DenseArgs a{};
  a.A = A.p; a.ld = D.ld; a.nrows = D.NR; a.ncols = D.NC; a.x = xd.p; a.d = d; a.n = n; a.m = m;
  a.lat = latd.p; a.sigmaT = STd.buf.p; a.rider = delta.p; a.rider_ld = N; a.nrider = ncol;
  launch_dense_assemble(a, st0);
  potrf_rec(A.p, D.ld, D.NR, 0, D.NC, W.p, N, info.p, st0);
  launch_lml_reduce(A.p, D.ld, N, D.NC, ncol, lml_dev.p, st0);
  std::vector<double> lml(ncol), resid(ncol);
  int hinfo = 0;
  HIPCHK(hipMemcpyAsync(lml.data(), lml_dev.p, ncol * sizeof(double), hipMemcpyDeviceToHost, st0));
  HIPCHK(hipMemcpyAsync(resid.data(), resid_dev.p, ncol * sizeof(double), hipMemcpyDeviceToHost, st0));
  HIPCHK(hipMemcpyAsync(&hinfo, info.p, sizeof(int), hipMemcpyDeviceToHost, st0));
  HIPCHK(hipStreamSynchronize(st0));
  if (int rc = check_info(std::vector<int>{hinfo}, 0)) return rc;
  for (int c = 0; c < ncol; ++c)      // reference src/ilmm.jl:171-181
    out[c] = lml[c] - ((double)n * ((double)(p - m) * kLog2Pi + ((double)p * std::log(sigma2) - logdetST)) + resid[c] / sigma2) / 2.0;
  return LMM_OK;
  LMM_CATCH
}

} // extern "C"

namespace {

struct IlmmGrad {            // host results of ilmm_grad_core
  double value = 0.0, gs2[LMM_MAX_NOISE_BLOCKS] = {};
  std::vector<double> gH;    // p x m
  std::vector<lmm_gp_grad_t> ggps;
};

// Value and gradient of the dense-H ILMM prior logpdf over n points in NB.nblk consecutive blocks, block b carrying observation
// noise NB.s2[b].  x (d x n), y (n x p by outputs) are DEVICE pointers; gy_dev (n x p, device) may be null.  The multi-block form
// exists for the posterior's predictive density: log p(y* | y) = log p(y, y*) - log p(y)   (T y is sufficient for the latents,
// so the reference's projected posterior, src/ilmm.jl:184-198, is the exact conditional), one block per conditioning batch.
// Hblk (optional): block b is observed through the mixing matrix Hblk[b] (p x m, host) instead of H -- the latent view of a posterior
// (lmm_ilmm_post_latent_logpdf_grad_seq) observes its test block through [I_m; 0]; G.gH collects the blocks observed through H itself.
int ilmm_grad_core(const double* xd, int d, int n, const NoiseBlocks& NB, const double* yd, int p, const double* H, int m,
                   const lmm_gp_t* gps, const lmm_jitters_t* jit, IlmmGrad& G, double* gy_dev, const double* const* Hblk = nullptr) {
  if ((long long)m * n > 46000) return fail(LMM_ERR_UNSUPPORTED, "m*n too large for the dense gradient (explicit (mn)^2 inverse)");
  hipStream_t st0 = g.streams[0];
  constexpr int KB = LMM_MAX_NOISE_BLOCKS;
  const int nblk = NB.nblk;
  int bi0[KB] = {}, bn[KB] = {};
  double s2[KB] = {};
  for (int b = 0; b < nblk; ++b) { bi0[b] = NB.off[b]; bn[b] = NB.count(b); s2[b] = NB.s2[b]; }
  const double* Hq[KB] = {};                       // the mixing matrix block b is observed through
  for (int b = 0; b < nblk; ++b) Hq[b] = (Hblk && Hblk[b]) ? Hblk[b] : H;
  std::vector<double> T[KB], ST[KB];
  double logdetST[KB] = {};
  for (int b = 0; b < nblk; ++b)
    if (int rc = project_dense(Hq[b], p, m, s2[b], jit->project_jitter, T[b], ST[b], &logdetST[b])) return rc;
  // host copies in the layouts the kernels read: Tt = T' (p x m), Ht = H' (m x p); the blocks' T, T', H and H' back to back
  std::vector<double> Hv, Ht, means(m), STall, Tall, Ttall;
  for (int b = 0; b < nblk; ++b) {
    std::vector<double> Ttb((size_t)p * m), Htb((size_t)m * p);
    for (int l = 0; l < m; ++l) for (int o = 0; o < p; ++o) { Ttb[o + (size_t)l * p] = T[b][l + (size_t)o * m]; Htb[l + (size_t)o * m] = Hq[b][o + (size_t)l * p]; }
    STall.insert(STall.end(), ST[b].begin(), ST[b].end());
    Tall.insert(Tall.end(), T[b].begin(), T[b].end());
    Ttall.insert(Ttall.end(), Ttb.begin(), Ttb.end());
    Hv.insert(Hv.end(), Hq[b], Hq[b] + (size_t)p * m);
    Ht.insert(Ht.end(), Htb.begin(), Htb.end());
  }
  for (int l = 0; l < m; ++l) means[l] = gps[l].mean;
  std::vector<LatentDev> lat(m);
  for (int l = 0; l < m; ++l) lat[l] = to_dev(gps[l]);
  std::vector<int> sidx(n);
  for (int b = 0; b < nblk; ++b)
    for (int i = bi0[b]; i < bi0[b] + bn[b]; ++i) sidx[i] = b;
  Uploaded Tdall(Tall, st0), STd(STall, st0), Hd(Hv, st0), Ttdall(Ttall, st0), Htd(Ht, st0), meansd(means, st0);
  const double* Tdv[KB] = {};
  const double* Ttdv[KB] = {};
  const double* Hdv[KB] = {};
  const double* Htdv[KB] = {};
  for (int b = 0; b < nblk; ++b) {
    Tdv[b] = Tdall.buf.p + (size_t)b * m * p; Ttdv[b] = Ttdall.buf.p + (size_t)b * m * p;
    Hdv[b] = Hd.buf.p + (size_t)b * m * p; Htdv[b] = Htd.buf.p + (size_t)b * m * p;
  }
  Buf<LatentDev> latd(m);
  Buf<int> sidxd(n);
  HIPCHK(hipMemcpyAsync(latd.p, lat.data(), m * sizeof(LatentDev), hipMemcpyHostToDevice, st0));
  HIPCHK(hipMemcpyAsync(sidxd.p, sidx.data(), n * sizeof(int), hipMemcpyHostToDevice, st0));
  const int N = m * n;
  Buf<double> Ty((size_t)N), delta((size_t)N), partial(tall_skinny_partials(n, p)), resid_dev(KB);
  for (int b = 0; b < nblk; ++b) {
    const int i0 = bi0[b], nb_ = bn[b];
    launch_tall_skinny(yd + i0, n, nb_, p, Tdv[b], m, m, Ty.p + i0, n, nullptr, nullptr, 0, nullptr, 0, st0);
    launch_tall_skinny(yd + i0, n, nb_, p, Tdv[b], m, m, delta.p + i0, n, meansd.buf.p, nullptr, 0, nullptr, 0, st0);
    // reference src/ilmm.jl:171-181: |Y - H T Y|_F^2 of the block
    launch_tall_skinny(Ty.p + i0, n, nb_, m, Hdv[b], p, p, nullptr, 0, nullptr, yd + i0, n, partial.p, 1, st0);
    launch_sum_partials(partial.p, tall_skinny_partials(nb_, p), resid_dev.p + b, st0);
  }
  Dims D(N, 1);
  Buf<double> A(mat_count(D.elems())), W(mat_count((size_t)(D.NC / 64) * 4096)), R(mat_count((size_t)D.ld * D.NC)), alpha((size_t)D.NC), lml_dev(1);
  Buf<int> info(1);
  HIPCHK(hipMemsetAsync(info.p, 0, sizeof(int), st0));
  HIPCHK(hipMemsetAsync(alpha.p, 0, (size_t)D.NC * sizeof(double), st0));
  DenseArgs a{};
  a.A = A.p; a.ld = D.ld; a.nrows = D.NR; a.ncols = D.NC; a.x = xd; a.d = d; a.n = n; a.m = m;
  a.lat = latd.p; a.sigmaT = STd.buf.p; a.sig_idx = nblk > 1 ? sidxd.p : nullptr; a.rider = delta.p; a.rider_ld = N; a.nrider = 1;
  launch_dense_assemble(a, st0);
  potrf_rec(A.p, D.ld, D.NR, 0, D.NC, W.p, N, info.p, st0);
  launch_lml_reduce(A.p, D.ld, N, D.NC, 1, lml_dev.p, st0);
  launch_extract_row(A.p, D.ld, D.NC, N, alpha.p, st0);
  backsolve1(A.p, D.ld, W.p, D.NC / 64, alpha.p, st0);
  launch_set_identity(R.p, D.ld, D.NC, st0);
  trsm_rec(R.p, D.ld, D.NC, A.p, D.ld, W.p, 0, D.NC, st0, true);                 // R = L^-T
  launch_syrk_upper_set(A.p, D.ld, R.p, D.ld, D.NC, st0);                         // lower(A) = Sigma^-1
  const int NGR = LMM_NGRAD;
  const size_t mm = (size_t)m * m, mp = (size_t)m * p;
  Buf<double> red((size_t)NGR * m), gpart((size_t)grad_partials(n)), Btr(KB * mm), AAt(KB * mm), AY(KB * mp);
  for (int l = 0; l < m; ++l)
    launch_grad_reduce(mat_at(A.p, (size_t)l * n * D.ld + (size_t)l * n), D.ld, n, n, alpha.p + (size_t)l * n, delta.p + (size_t)l * n, xd, d,
                       lat[l], gpart.p, red.p + (size_t)NGR * l, st0);
  // regulariser pieces: Rm = Y - (T Y)' H' (n x p), RH = Rm H (n x m), per block Rm' Ty (p x m), RH' Y (m x p)
  Buf<double> HTY((size_t)n * p), Rm((size_t)n * p), RH((size_t)N), RtTy(KB * mp), RHtY(KB * mp);
  for (int b = 0; b < nblk; ++b)
    launch_tall_skinny(Ty.p + bi0[b], n, bn[b], m, Hdv[b], p, p, HTY.p + bi0[b], n, nullptr, nullptr, 0, nullptr, 0, st0);
  launch_vec_lin(yd, HTY.p, -1.0, n * p, Rm.p, st0);
  for (int b = 0; b < nblk; ++b)
    launch_tall_skinny(Rm.p + bi0[b], n, bn[b], p, Htdv[b], m, m, RH.p + bi0[b], n, nullptr, nullptr, 0, nullptr, 0, st0);
  for (int b = 0; b < nblk; ++b) {
    const int i0 = bi0[b], nb_ = bn[b];
    launch_block_trace(A.p, D.ld, n, m, i0, i0 + nb_, Btr.p + b * mm, st0);
    launch_atb(alpha.p + i0, n, alpha.p + i0, n, nb_, m, m, AAt.p + b * mm, st0);      // (alpha_l . alpha_l') over the block
    launch_atb(alpha.p + i0, n, yd + i0, n, nb_, m, p, AY.p + b * mp, st0);            // sum_i alpha_l[i] Y[i, o]   (m x p)
    launch_atb(Rm.p + i0, n, Ty.p + i0, n, nb_, p, m, RtTy.p + b * mp, st0);
    launch_atb(RH.p + i0, n, yd + i0, n, nb_, m, p, RHtY.p + b * mp, st0);
  }
  std::vector<double> hred((size_t)NGR * m), hB(KB * mm), hAAt(KB * mm), hAY(KB * mp), hRtTy(KB * mp), hRHtY(KB * mp);
  double lml = 0.0, resid[KB] = {};
  int hinfo = 0;
  HIPCHK(hipMemcpyAsync(&lml, lml_dev.p, sizeof(double), hipMemcpyDeviceToHost, st0));
  HIPCHK(hipMemcpyAsync(resid, resid_dev.p, nblk * sizeof(double), hipMemcpyDeviceToHost, st0));
  HIPCHK(hipMemcpyAsync(&hinfo, info.p, sizeof(int), hipMemcpyDeviceToHost, st0));
  HIPCHK(hipMemcpyAsync(hred.data(), red.p, hred.size() * sizeof(double), hipMemcpyDeviceToHost, st0));
  HIPCHK(hipMemcpyAsync(hB.data(), Btr.p, nblk * mm * sizeof(double), hipMemcpyDeviceToHost, st0));
  HIPCHK(hipMemcpyAsync(hAAt.data(), AAt.p, nblk * mm * sizeof(double), hipMemcpyDeviceToHost, st0));
  HIPCHK(hipMemcpyAsync(hAY.data(), AY.p, nblk * mp * sizeof(double), hipMemcpyDeviceToHost, st0));
  HIPCHK(hipMemcpyAsync(hRtTy.data(), RtTy.p, nblk * mp * sizeof(double), hipMemcpyDeviceToHost, st0));
  HIPCHK(hipMemcpyAsync(hRHtY.data(), RHtY.p, nblk * mp * sizeof(double), hipMemcpyDeviceToHost, st0));
  HIPCHK(hipStreamSynchronize(st0));
  if (int rc = check_info(std::vector<int>{hinfo}, 0)) return rc;
  // value: reference src/ilmm.jl:150-163 + :171-181
  G.value = lml;
  for (int b = 0; b < nblk; ++b)
    G.value -= ((double)bn[b] * ((double)(p - m) * kLog2Pi + ((double)p * std::log(s2[b]) - logdetST[b])) + resid[b] / s2[b]) / 2.0;
  // ---- kernel-parameter gradients: 1/2 tr((aa' - Sigma^-1) dSigma/dtheta_l), dSigma = E_ll (x) dK_l ----
  G.ggps.assign(m, lmm_gp_grad_t{});
  for (int l = 0; l < m; ++l) {
    const double* r = &hred[(size_t)NGR * l];
    G.ggps[l].lengthscale = r[0];
    G.ggps[l].variance = (r[7] + 0.5 * gps[l].variance * (r[2] - r[1])) / gps[l].variance;     // K_ii = variance
    G.ggps[l].mean = r[4];
  }
  std::vector<double> Hacc((size_t)p * m, 0.0), HtH(mm, 0.0);
  for (int b = 0; b < nblk; ++b) {
    const double* H = Hq[b];                        // (shadows the argument: everything below differentiates through THIS block's mixing matrix)
    std::vector<double> Hb((size_t)p * m, 0.0);     // its cotangent; kept only for the blocks observed through the model's H
    for (int aI = 0; aI < m; ++aI)
      for (int bb = 0; bb < m; ++bb) {
        double acc = 0.0;
        for (int o = 0; o < p; ++o) acc += H[o + (size_t)aI * p] * H[o + (size_t)bb * p];
        HtH[aI + (size_t)bb * m] = acc;
      }
    const double sigma2 = s2[b], s = 1.0 / sigma2;
    const std::vector<double>& Tq = T[b];
    const double* bAY = &hAY[b * mp]; const double* bRHtY = &hRHtY[b * mp]; const double* bRtTy = &hRtTy[b * mp];
    const double* bAAt = &hAAt[b * mm]; const double* bB = &hB[b * mm];
    // ---- cotangents of T (m x p) and SigmaT (m x m) of this block ----
    std::vector<double> Tb((size_t)m * p, 0.0), Gs(mm, 0.0);
    double s2g = 0.0;
    for (int l = 0; l < m; ++l)
      for (int o = 0; o < p; ++o) Tb[l + (size_t)o * m] = -bAY[l + (size_t)o * m] + s * bRHtY[l + (size_t)o * m];   // lml + regulariser (residual)
    // SigmaT^-1 for the +n/2 logdet SigmaT term of the regulariser
    std::vector<double> STc = ST[b], STinv(mm, 0.0);
    if (!host_cholesky(STc, m)) return fail(LMM_ERR_NOT_PD, "PosDefException: SigmaT not PD");
    for (int c = 0; c < m; ++c) {       // solve (L L') col = e_c
      std::vector<double> v(m, 0.0);
      for (int aI = 0; aI < m; ++aI) { double t = (aI == c) ? 1.0 : 0.0; for (int k = 0; k < aI; ++k) t -= STc[aI + (size_t)k * m] * v[k]; v[aI] = t / STc[aI + (size_t)aI * m]; }
      for (int aI = m - 1; aI >= 0; --aI) { double t = v[aI]; for (int k = aI + 1; k < m; ++k) t -= STc[k + (size_t)aI * m] * v[k]; v[aI] = t / STc[aI + (size_t)aI * m]; }
      for (int aI = 0; aI < m; ++aI) STinv[aI + (size_t)c * m] = v[aI];
    }
    for (int aI = 0; aI < m; ++aI)
      for (int bb = 0; bb < m; ++bb) Gs[aI + (size_t)bb * m] = 0.5 * (bAAt[aI + (size_t)bb * m] - bB[aI + (size_t)bb * m]) + 0.5 * (double)bn[b] * STinv[aI + (size_t)bb * m];
    // explicit sigma2 of the regulariser and explicit H of the residual
    s2g += -0.5 * ((double)bn[b] * (double)p / sigma2 - resid[b] / (sigma2 * sigma2));
    for (int o = 0; o < p; ++o) for (int l = 0; l < m; ++l) Hb[o + (size_t)l * p] += s * bRtTy[o + (size_t)l * p];
    // ---- backward through project(H, sigma2): P = s H'H + eps I,  T = P^-1 H' s,  SigmaT = sigma2 T T' ----
    // SigmaT = sigma2 T T':  Tb += sigma2 (Gs + Gs') T;  s2g += <Gs, T T'>
    for (int aI = 0; aI < m; ++aI)
      for (int o = 0; o < p; ++o) {
        double acc = 0.0;
        for (int bb = 0; bb < m; ++bb) acc += (Gs[aI + (size_t)bb * m] + Gs[bb + (size_t)aI * m]) * Tq[bb + (size_t)o * m];
        Tb[aI + (size_t)o * m] += sigma2 * acc;
      }
    for (int aI = 0; aI < m; ++aI)
      for (int bb = 0; bb < m; ++bb) {
        double tt = 0.0;
        for (int o = 0; o < p; ++o) tt += Tq[aI + (size_t)o * m] * Tq[bb + (size_t)o * m];
        s2g += Gs[aI + (size_t)bb * m] * tt;
      }
    // P and its Cholesky
    std::vector<double> P(mm, 0.0);
    for (int aI = 0; aI < m; ++aI)
      for (int bb = 0; bb < m; ++bb) P[aI + (size_t)bb * m] = s * HtH[aI + (size_t)bb * m] + (aI == bb ? jit->project_jitter : 0.0);
    if (!host_cholesky(P, m)) return fail(LMM_ERR_NOT_PD, "PosDefException in project(H, sigma2)");
    // Mb = P^-1 Tb (m x p);  Pb = -Mb T'
    std::vector<double> Mb((size_t)m * p, 0.0), Pb(mm, 0.0);
    for (int o = 0; o < p; ++o) {
      std::vector<double> v(m);
      for (int aI = 0; aI < m; ++aI) { double t = Tb[aI + (size_t)o * m]; for (int k = 0; k < aI; ++k) t -= P[aI + (size_t)k * m] * v[k]; v[aI] = t / P[aI + (size_t)aI * m]; }
      for (int aI = m - 1; aI >= 0; --aI) { double t = v[aI]; for (int k = aI + 1; k < m; ++k) t -= P[k + (size_t)aI * m] * v[k]; v[aI] = t / P[aI + (size_t)aI * m]; }
      for (int aI = 0; aI < m; ++aI) Mb[aI + (size_t)o * m] = v[aI];
    }
    for (int aI = 0; aI < m; ++aI)
      for (int bb = 0; bb < m; ++bb) {
        double acc = 0.0;
        for (int o = 0; o < p; ++o) acc += Mb[aI + (size_t)o * m] * Tq[bb + (size_t)o * m];
        Pb[aI + (size_t)bb * m] = -acc;
      }
    double sb = 0.0;       // cotangent of s = 1 / sigma2
    for (int o = 0; o < p; ++o)
      for (int l = 0; l < m; ++l) {
        Hb[o + (size_t)l * p] += s * Mb[l + (size_t)o * m];                      // M = H' s
        sb += Mb[l + (size_t)o * m] * H[o + (size_t)l * p];
        double acc = 0.0;                                                        // P = s H'H: Hb += s H (Pb + Pb')
        for (int bb = 0; bb < m; ++bb) acc += H[o + (size_t)bb * p] * (Pb[bb + (size_t)l * m] + Pb[l + (size_t)bb * m]);
        Hb[o + (size_t)l * p] += s * acc;
      }
    for (int aI = 0; aI < m; ++aI) for (int bb = 0; bb < m; ++bb) sb += Pb[aI + (size_t)bb * m] * HtH[aI + (size_t)bb * m];
    s2g += -sb * s * s;
    G.gs2[b] = s2g;
    if (!(Hblk && Hblk[b])) for (size_t q = 0; q < Hacc.size(); ++q) Hacc[q] += Hb[q];
  }
  G.gH = Hacc;
  if (gy_dev) {
    // dL/dY (n x p) = -((alpha - RH / sigma2_i) T_i) - Rm / sigma2_i     (alpha as the n x m matrix [point][latent])
    Buf<double> Z((size_t)N), ZT((size_t)n * p);
    launch_vec_lin_blocks(alpha.p, RH.p, NB, -1.0, n, (size_t)N, Z.p, st0);
    for (int b = 0; b < nblk; ++b)
      launch_tall_skinny(Z.p + bi0[b], n, bn[b], m, Ttdv[b], p, p, ZT.p + bi0[b], n, nullptr, nullptr, 0, nullptr, 0, st0);
    launch_vec_lin_blocks(ZT.p, Rm.p, NB, 1.0, n, (size_t)n * p, ZT.p, st0);
    launch_vec_axpby(ZT.p, -1.0, ZT.p, 0.0, (size_t)n * p, gy_dev, st0);
    HIPCHK(hipStreamSynchronize(st0));
  }
  return LMM_OK;
}

}  // namespace

extern "C" {

// Value and gradient of logpdf(fx::FiniteGP{<:ILMM}, y) with a dense H (reference src/ilmm.jl:150-181 differentiated; the
// reference's tests take Zygote.gradient(logpdf, ilmmx, y), test/ilmm.jl:31) w.r.t. y, sigma2, H (p x m) and every latent's
// (variance, lengthscale, mean).  The reference's own operation: ONE (mn) x (mn) factorisation of blockdiag(K_l) + SigmaT (x) I;
// here additionally its explicit inverse (triangular solve of identity riders + upper-triangular SYRK on the MFMA kernels),
// per-latent contractions on the diagonal blocks of the inverse, and the chain rule through project(H, sigma2)
// (src/ilmm.jl:61-68) and the regulariser (src/ilmm.jl:171-181) as small host algebra.  Does not shard.
int lmm_ilmm_logpdf_grad(const double* x, int d, int n, const double* y, int p, const double* H, int m, double sigma2,
                         const lmm_gp_t* gps, const lmm_jitters_t* jit, double* out_logpdf, double* grad_y, double* grad_sigma2,
                         double* grad_H, lmm_gp_grad_t* grad_gps) {
  std::lock_guard<std::mutex> lk(g_mu);
  REQUIRE_INIT();
  LMM_TRY
  if (!x || !y || !H || !out_logpdf || d <= 0 || n <= 0 || p <= 0 || m <= 0) return fail(LMM_ERR_ARG, "bad arguments");
  if (int rc = check_gps(gps, m)) return rc;
  if (!(sigma2 > 0.0)) return fail(LMM_ERR_ARG, "sigma2 must be > 0");
  if (!jit) jit = &kDefaultJit;
  hipStream_t st0 = g.streams[0];
  DevIn xd(x, (size_t)d * n, st0), yd(y, (size_t)n * p, st0);
  DevOut gy(grad_y, (size_t)n * p);
  IlmmGrad G;
  if (int rc = ilmm_grad_core(xd.p, d, n, one_noise_block(n, sigma2), yd.p, p, H, m, gps, jit, G, gy.p)) return rc;
  *out_logpdf = G.value;
  if (grad_sigma2) *grad_sigma2 = G.gs2[0];
  if (grad_H) std::copy(G.gH.begin(), G.gH.end(), grad_H);
  if (grad_gps) for (int l = 0; l < m; ++l) grad_gps[l] = G.ggps[l];
  if (grad_y) { gy.finish(st0); HIPCHK(hipStreamSynchronize(st0)); }
  return LMM_OK;
  LMM_CATCH
}

// Value and TOTAL derivatives of logpdf(posterior(f(x, sigma2), y)(xs, sigma2_s), ys) for the dense-H ILMM -- what
// Zygote.gradient(logpdf, pi, y_test) differentiates in reference test/ilmm.jl:32 -- as the joint prior density of (y, ys) under
// per-block noise minus the prior density of y.  Does not shard.  _seq: sequentially conditioned posterior (src/ilmm.jl:184-198
// applied to its own result), one noise block per conditioning batch; x, y as in lmm_oilmm_post_logpdf_grad_seq.
}  // extern "C"

namespace {
// latent_test: the test block observes the LATENT processes (ys is ns x m): get_latent_gp(posterior)(xs, sigma2_s)
int ilmm_post_logpdf_grad_impl(bool latent_test, const double* x, int d, int n, const int* batch_n, const double* batch_sigma2, int nbatch,
                               const double* y, const double* xs, int ns, const double* ys, int p, const double* H, int m,
                               double sigma2_s, const lmm_gp_t* gps, const lmm_jitters_t* jit, double* out_logpdf, double* grad_y,
                               double* grad_ys, double* grad_batch_sigma2, double* grad_sigma2_s, double* grad_H,
                               lmm_gp_grad_t* grad_gps) {
  if (!x || !y || !xs || !ys || !H || !out_logpdf || d <= 0 || n <= 0 || ns <= 0 || p <= 0 || m <= 0) return fail(LMM_ERR_ARG, "bad arguments");
  if (latent_test && m > p) return fail(LMM_ERR_DIM, "out dim of x != out dim of f.");
  if (int rc = check_gps(gps, m)) return rc;
  if (int rc = check_batches(batch_n, batch_sigma2, nbatch, n)) return rc;
  if (!(sigma2_s > 0.0)) return fail(LMM_ERR_ARG, "sigma2 must be > 0");
  if (!jit) jit = &kDefaultJit;
  hipStream_t st0 = g.streams[0];
  const int N = n + ns, pt = latent_test ? m : p;          // columns of ys
  DevIn xd(x, (size_t)d * n, st0), yd(y, (size_t)n * p, st0), xsd(xs, (size_t)d * ns, st0), ysd(ys, (size_t)ns * pt, st0);
  Buf<double> xj((size_t)d * N), yj((size_t)N * p), gj((size_t)N * p), gm((size_t)n * p);
  HIPCHK(hipMemcpyAsync(xj.p, xd.p, (size_t)d * n * sizeof(double), hipMemcpyDeviceToDevice, st0));
  HIPCHK(hipMemcpyAsync(xj.p + (size_t)d * n, xsd.p, (size_t)d * ns * sizeof(double), hipMemcpyDeviceToDevice, st0));
  if (pt < p) HIPCHK(hipMemsetAsync(yj.p, 0, (size_t)N * p * sizeof(double), st0));        // the test block's columns m .. p-1 stay 0
  HIPCHK(hipMemcpy2DAsync(yj.p, (size_t)N * sizeof(double), yd.p, (size_t)n * sizeof(double), (size_t)n * sizeof(double), p, hipMemcpyDeviceToDevice, st0));
  HIPCHK(hipMemcpy2DAsync(yj.p + n, (size_t)N * sizeof(double), ysd.p, (size_t)ns * sizeof(double), (size_t)ns * sizeof(double), pt, hipMemcpyDeviceToDevice, st0));
  // The latent view observes its test points through H* = [I_m; 0] (p x m): z embedded in the first m of p output columns.  project(H*,
  // s) gives T = [I 0], SigmaT = s I (src/ilmm.jl:61-68) -- the latent FiniteGP's own noise -- and a residual of 0; what the p - m
  // padded columns add to the regulariser, -ns (p - m) log(2 pi s) / 2, is taken out again below.
  std::vector<double> Hlat;
  const double* Hblk[LMM_MAX_NOISE_BLOCKS] = {};
  if (latent_test) {
    Hlat.assign((size_t)p * m, 0.0);
    for (int l = 0; l < m; ++l) Hlat[l + (size_t)l * p] = 1.0;
    Hblk[nbatch] = Hlat.data();
  }
  const bool want_gy = grad_y != nullptr || grad_ys != nullptr;
  IlmmGrad GJ, GM;
  if (int rc = ilmm_grad_core(xj.p, d, N, batch_noise_blocks(batch_n, batch_sigma2, nbatch, ns, sigma2_s), yj.p, p, H, m, gps, jit, GJ,
                              want_gy ? gj.p : nullptr, latent_test ? Hblk : nullptr)) return rc;
  if (int rc = ilmm_grad_core(xd.p, d, n, batch_noise_blocks(batch_n, batch_sigma2, nbatch, 0, 0.0), yd.p, p, H, m, gps, jit, GM,
                              grad_y ? gm.p : nullptr)) return rc;
  const double pad = latent_test ? 0.5 * (double)ns * (double)(p - m) : 0.0;
  *out_logpdf = GJ.value - GM.value + pad * (kLog2Pi + std::log(sigma2_s));
  if (grad_batch_sigma2) for (int b = 0; b < nbatch; ++b) grad_batch_sigma2[b] = GJ.gs2[b] - GM.gs2[b];
  if (grad_sigma2_s) *grad_sigma2_s = GJ.gs2[nbatch] + pad / sigma2_s;
  if (grad_H) for (size_t q = 0; q < (size_t)p * m; ++q) grad_H[q] = GJ.gH[q] - GM.gH[q];
  if (grad_gps)
    for (int l = 0; l < m; ++l) {
      grad_gps[l].variance = GJ.ggps[l].variance - GM.ggps[l].variance;
      grad_gps[l].lengthscale = GJ.ggps[l].lengthscale - GM.ggps[l].lengthscale;
      grad_gps[l].mean = GJ.ggps[l].mean - GM.ggps[l].mean;
    }
  if (grad_y) {
    DevOut gy(grad_y, (size_t)n * p);
    Buf<double> top((size_t)n * p);
    HIPCHK(hipMemcpy2DAsync(top.p, (size_t)n * sizeof(double), gj.p, (size_t)N * sizeof(double), (size_t)n * sizeof(double), p, hipMemcpyDeviceToDevice, st0));
    launch_vec_lin(top.p, gm.p, -1.0, n * p, gy.p, st0);
    gy.finish(st0);
    HIPCHK(hipStreamSynchronize(st0));
  }
  if (grad_ys) {
    DevOut gys(grad_ys, (size_t)ns * pt);
    HIPCHK(hipMemcpy2DAsync(gys.p, (size_t)ns * sizeof(double), gj.p + n, (size_t)N * sizeof(double), (size_t)ns * sizeof(double), pt, hipMemcpyDeviceToDevice, st0));
    gys.finish(st0);
    HIPCHK(hipStreamSynchronize(st0));
  }
  return LMM_OK;
}
}  // namespace

extern "C" {

int lmm_ilmm_post_logpdf_grad_seq(const double* x, int d, int n, const int* batch_n, const double* batch_sigma2, int nbatch,
                                  const double* y, const double* xs, int ns, const double* ys, int p, const double* H, int m,
                                  double sigma2_s, const lmm_gp_t* gps, const lmm_jitters_t* jit, double* out_logpdf, double* grad_y,
                                  double* grad_ys, double* grad_batch_sigma2, double* grad_sigma2_s, double* grad_H,
                                  lmm_gp_grad_t* grad_gps) {
  std::lock_guard<std::mutex> lk(g_mu);
  REQUIRE_INIT();
  LMM_TRY
  return ilmm_post_logpdf_grad_impl(false, x, d, n, batch_n, batch_sigma2, nbatch, y, xs, ns, ys, p, H, m, sigma2_s, gps, jit, out_logpdf,
                                    grad_y, grad_ys, grad_batch_sigma2, grad_sigma2_s, grad_H, grad_gps);
  LMM_CATCH
}

// The same for the LATENT view of the posterior: logpdf(get_latent_gp(posterior(...))(xs, sigma2_s), zs) with zs (ns x m, by outputs over
// the m latents) -- reference src/ilmm.jl:39 on the posterior ILMM of :196-197, whose latent GP is the coupled PosteriorGP of the
// IndependentMOGP; Zygote differentiates its logpdf like any other.  grad_ys: ns x m.  grad_H: through the conditioning batches only.
int lmm_ilmm_post_latent_logpdf_grad_seq(const double* x, int d, int n, const int* batch_n, const double* batch_sigma2, int nbatch,
                                         const double* y, const double* xs, int ns, const double* zs, int p, const double* H, int m,
                                         double sigma2_s, const lmm_gp_t* gps, const lmm_jitters_t* jit, double* out_logpdf, double* grad_y,
                                         double* grad_zs, double* grad_batch_sigma2, double* grad_sigma2_s, double* grad_H,
                                         lmm_gp_grad_t* grad_gps) {
  std::lock_guard<std::mutex> lk(g_mu);
  REQUIRE_INIT();
  LMM_TRY
  return ilmm_post_logpdf_grad_impl(true, x, d, n, batch_n, batch_sigma2, nbatch, y, xs, ns, zs, p, H, m, sigma2_s, gps, jit, out_logpdf,
                                    grad_y, grad_zs, grad_batch_sigma2, grad_sigma2_s, grad_H, grad_gps);
  LMM_CATCH
}

// One conditioning batch: posterior(f(x, sigma2), y).
int lmm_ilmm_post_logpdf_grad(const double* x, int d, int n, const double* y, const double* xs, int ns, const double* ys, int p,
                              const double* H, int m, double sigma2, double sigma2_s, const lmm_gp_t* gps, const lmm_jitters_t* jit,
                              double* out_logpdf, double* grad_y, double* grad_ys, double* grad_sigma2, double* grad_sigma2_s,
                              double* grad_H, lmm_gp_grad_t* grad_gps) {
  return lmm_ilmm_post_logpdf_grad_seq(x, d, n, &n, &sigma2, 1, y, xs, ns, ys, p, H, m, sigma2_s, gps, jit, out_logpdf, grad_y, grad_ys,
                                       grad_sigma2, grad_sigma2_s, grad_H, grad_gps);
}

// ------------------------------------------------------------------------------------------------
// posterior
// ------------------------------------------------------------------------------------------------
// noise: per-latent scalar (host, indexed by latent) used when noisevec == NULL; noisevec: device [k][n] per-point noise.
static int posterior_create_common(const double* xd, int d, int n, const lmm_gp_t* gps, int m, const double* noise,
                                   int l0, int l1, const double* delta, lmm_post_t** out,
                                   const double* noisevec = nullptr) {
  const int ms = l1 - l0;
  lmm_post* P = new lmm_post();
  try {
    Dims D(n, 1);
    P->kind = 0; P->f32 = g_f32; P->n = n; P->d = d; P->l0 = l0; P->l1 = l1; P->m = m;
    P->NC = D.NC; P->NR = D.NR; P->ld = D.ld;
    P->gps.assign(gps, gps + m);
    P->x = Buf<double>((size_t)d * n);
    HIPCHK(hipMemcpyAsync(P->x.p, xd, (size_t)d * n * sizeof(double), hipMemcpyDeviceToDevice, g.streams[0]));
    Buf<int> info(std::max(ms, 1));
    HIPCHK(hipMemsetAsync(info.p, 0, std::max(ms, 1) * sizeof(int), g.streams[0]));
    int nb_per = 1, nslots = 1;
    batch_plan(std::max(ms, 1), &nb_per, &nslots, mat_bytes((double)D.elems()));
    for (int k = 0; k < ms; ++k) {
      P->L.emplace_back(mat_count((size_t)D.elems()));
      P->W.emplace_back(mat_count((size_t)(D.NC / 64) * 4096));
      P->alpha.emplace_back((size_t)D.NC);
      P->z.emplace_back((size_t)D.NC);
    }
    P->delta_all = Buf<double>((size_t)n * std::max(ms, 1));
    if (ms > 0) HIPCHK(hipMemcpyAsync(P->delta_all.p, delta, (size_t)n * ms * sizeof(double), hipMemcpyDeviceToDevice, g.streams[0]));
    if (noisevec) {
      P->noise_all = Buf<double>((size_t)n * std::max(ms, 1));
      if (ms > 0) HIPCHK(hipMemcpyAsync(P->noise_all.p, noisevec, (size_t)n * ms * sizeof(double), hipMemcpyDeviceToDevice, g.streams[0]));
    } else {
      P->noise_scalar.assign(noise + l0, noise + l1);
    }
    fork_slots(nslots);
    int bi = 0;
    for (int k0 = 0; k0 < ms; k0 += nb_per, ++bi) {
      hipStream_t st = g.streams[bi % nslots];
      const int nb = std::min(nb_per, ms - k0);
      Batch B;
      GramArgs ga[LMM_MAX_BATCH];
      for (int j = 0; j < nb; ++j) {
        const int k = k0 + j;
        const lmm_gp_t& gp = gps[l0 + k];
        GramArgs a{};
        a.A = P->L[k].p; a.ld = D.ld; a.nrows = D.NR; a.ncols = D.NC; a.x = P->x.p; a.d = d; a.n = n;
        a.kind = gp.kind; a.var = gp.variance; a.inv_ls = 1.0 / gp.lengthscale; a.pad_diag = 1.0;
        a.diag_add = noisevec ? 0.0 : noise[l0 + k];
        a.diag_vec = noisevec ? P->noise_all.p + (size_t)k * n : nullptr;
        a.rider = delta + (size_t)k * n; a.rider_ld = n; a.nrider = 1;
        ga[j] = a;
        B.add(P->L[k].p, P->W[k].p, info.p + k);
      }
      {
        const double gb = (double)n * ((double)n + 1.0) / 2.0 * 8.0;
        ProfScope ps(LMM_PROF_GRAM, nb * gb, st, 0, 0, 0, nb * gb, nb);
        gram_batch_g(ga, nb, st);
      }
      potrf_batch(B, D.ld, D.NR, D.NC, n, st, D.NC + 1);        // one rider row (delta); rows NC + 1 .. NR - 1 are zero padding
      // alpha = L^-T (L^-1 delta): the rider row is z = L^-1 delta (kept as P->z, zero-padded to NC)
      BatchPtr ab{}, zb{};
      for (int j = 0; j < nb; ++j) { ab.p[j] = P->alpha[k0 + j].p; zb.p[j] = P->z[k0 + j].p; }
      launch_extract_rows(B.A, nb, D.ld, D.NC, n, D.NC, ab, zb, st);
      launch_backsolve(B.A, D.ld, B.W, D.NC / 64, ab, nb, st);
    }
    join_slots(nslots);
    std::vector<int> hinfo(std::max(ms, 1), 0);
    HIPCHK(hipMemcpyAsync(hinfo.data(), info.p, std::max(ms, 1) * sizeof(int), hipMemcpyDeviceToHost, g.streams[0]));
    HIPCHK(hipStreamSynchronize(g.streams[0]));
    if (int rc = check_info(hinfo, l0)) { delete P; return rc; }
  } catch (int code) { drain_after_error(); delete P; return code; }
  *out = P;
  return LMM_OK;
}

int lmm_oilmm_posterior_create(const double* x, int d, int n, const double* y, int p, const double* U, const double* S,
                               int m, double sigma2, const lmm_gp_t* gps, int latent_begin, int latent_end,
                               lmm_post_t** out) {
  std::lock_guard<std::mutex> lk(g_mu);
  REQUIRE_INIT();
  LMM_TRY
  if (!x || !y || !U || !S || !out || d <= 0 || n <= 0 || p <= 0 || m <= 0) return fail(LMM_ERR_ARG, "bad arguments");
  if (m > p) return fail(LMM_ERR_DIM, "out dim of x != out dim of f.");
  if (latent_begin < 0 || latent_end > m || latent_begin > latent_end) return fail(LMM_ERR_ARG, "bad latent shard");
  if (int rc = check_gps(gps, m)) return rc;
  hipStream_t st0 = g.streams[0];
  std::vector<double> T, ST, H;
  project_orthogonal(U, S, p, m, sigma2, T, ST, H);
  DevIn xd(x, (size_t)d * n, st0), yd(y, (size_t)n * p, st0);
  Uploaded Td(T, st0);
  std::vector<double> means(m);
  for (int l = 0; l < m; ++l) means[l] = gps[l].mean;
  Uploaded meansd(means, st0);
  const int ms = latent_end - latent_begin;
  Buf<double> delta((size_t)n * std::max(ms, 1));
  if (ms > 0) project_on_device(yd.p, n, p, Td.buf, m, latent_begin, ms, meansd.buf.p + latent_begin, delta.p, st0);
  return posterior_create_common(xd.p, d, n, gps, m, ST.data(), latent_begin, latent_end, delta.p, out);
  LMM_CATCH
}

// posterior(po(x2, sigma2), y2) -- conditioning a posterior OILMM / IndependentMOGP on further observations (exercised by
// AbstractGPs.TestUtils on `po` in reference test/oilmm.jl:34-37; AbstractGPs updates the Cholesky factor).  Latent by
// latent the result is the posterior of the PRIOR given both data sets, each with its own projected noise, so the new state is
// built from the concatenated inputs, the kept residuals and per-point noise.  U, S: the mixing matrix (U = I, S = 1 for a
// bare IndependentMOGP, with p == m).
int lmm_post_condition(const lmm_post_t* post, const double* U, const double* S, int p, int m, double sigma2,
                       const double* x2, int d, int n2, const double* y2, lmm_post_t** out) {
  std::lock_guard<std::mutex> lk(g_mu);
  REQUIRE_INIT();
  LMM_TRY
  if (post && post->f32 != g_f32) return fail(LMM_ERR_ARG, "posterior handle was built in the other compute dtype (lmm_set_compute_dtype)");
  if (!post || !U || !S || !x2 || !y2 || !out || d <= 0 || n2 <= 0) return fail(LMM_ERR_ARG, "bad arguments");
  const lmm_post* P = post;
  if (P->kind != 0) return fail(LMM_ERR_UNSUPPORTED, "sequential conditioning of the dense-H posterior is not built");
  if (P->m != m) return fail(LMM_ERR_DIM, "posterior has %d latents, H has %d", P->m, m);
  if (P->d != d) return fail(LMM_ERR_DIM, "input dimension mismatch");
  if (m > p) return fail(LMM_ERR_DIM, "out dim of x != out dim of f.");
  hipStream_t st0 = g.streams[0];
  const int l0 = P->l0, l1 = P->l1, ms = l1 - l0, n1 = P->n, n = n1 + n2;
  std::vector<double> T, ST, H;
  project_orthogonal(U, S, p, m, sigma2, T, ST, H);
  DevIn x2d(x2, (size_t)d * n2, st0), y2d(y2, (size_t)n2 * p, st0);
  Uploaded Td(T, st0);
  std::vector<double> means(m);
  for (int l = 0; l < m; ++l) means[l] = P->gps[l].mean;
  Uploaded meansd(means, st0);
  Buf<double> xall((size_t)d * n), delta((size_t)n * std::max(ms, 1)), nv((size_t)n * std::max(ms, 1)), d2buf((size_t)n2 * std::max(ms, 1));
  HIPCHK(hipMemcpyAsync(xall.p, P->x.p, (size_t)d * n1 * sizeof(double), hipMemcpyDeviceToDevice, st0));
  HIPCHK(hipMemcpyAsync(xall.p + (size_t)d * n1, x2d.p, (size_t)d * n2 * sizeof(double), hipMemcpyDeviceToDevice, st0));
  if (ms > 0) project_on_device(y2d.p, n2, p, Td.buf, m, l0, ms, meansd.buf.p + l0, d2buf.p, st0);
  for (int k = 0; k < ms; ++k) {
    HIPCHK(hipMemcpyAsync(delta.p + (size_t)k * n, P->delta_all.p + (size_t)k * n1, (size_t)n1 * sizeof(double), hipMemcpyDeviceToDevice, st0));
    HIPCHK(hipMemcpyAsync(delta.p + (size_t)k * n + n1, d2buf.p + (size_t)k * n2, (size_t)n2 * sizeof(double), hipMemcpyDeviceToDevice, st0));
    if (P->noise_all.p != nullptr)
      HIPCHK(hipMemcpyAsync(nv.p + (size_t)k * n, P->noise_all.p + (size_t)k * n1, (size_t)n1 * sizeof(double), hipMemcpyDeviceToDevice, st0));
    else launch_fill(nv.p + (size_t)k * n, n1, P->noise_scalar[k], st0);
    launch_fill(nv.p + (size_t)k * n + n1, n2, ST[l0 + k], st0);
  }
  return posterior_create_common(xall.p, d, n, P->gps.data(), m, ST.data(), l0, l1, delta.p, out, nv.p);
  LMM_CATCH
}

int lmm_mogp_posterior_create(const double* x, int d, int n, const double* y, int m, double sigma2, const lmm_gp_t* gps,
                              int latent_begin, int latent_end, lmm_post_t** out) {
  std::lock_guard<std::mutex> lk(g_mu);
  REQUIRE_INIT();
  LMM_TRY
  if (!x || !y || !out || d <= 0 || n <= 0 || m <= 0) return fail(LMM_ERR_ARG, "bad arguments");
  if (latent_begin < 0 || latent_end > m || latent_begin > latent_end) return fail(LMM_ERR_ARG, "bad latent shard");
  if (int rc = check_gps(gps, m)) return rc;
  hipStream_t st0 = g.streams[0];
  DevIn xd(x, (size_t)d * n, st0), yd(y, (size_t)n * m, st0);
  std::vector<double> T((size_t)m * m, 0.0), means(m), noise(m, sigma2);
  for (int l = 0; l < m; ++l) { T[l + (size_t)l * m] = 1.0; means[l] = gps[l].mean; }
  Uploaded Td(T, st0), meansd(means, st0);
  const int ms = latent_end - latent_begin;
  Buf<double> delta((size_t)n * std::max(ms, 1));
  if (ms > 0) project_on_device(yd.p, n, m, Td.buf, m, latent_begin, ms, meansd.buf.p + latent_begin, delta.p, st0);
  return posterior_create_common(xd.p, d, n, gps, m, noise.data(), latent_begin, latent_end, delta.p, out);
  LMM_CATCH
}

// Dense-H posterior state from the stacked inputs xd (d x n, device), the projected residuals delta ([latent][point], m n,
// device) and the per-batch SigmaT list: assemble blockdiag(K_l) + SigmaT_{batch(i)} (x) e_i e_i', factor, alpha = C \ delta.
static int dense_posterior_build(const double* xd, int d, int n, const double* H, int p, int m, const lmm_gp_t* gps,
                                 const double* delta, const std::vector<double>& sigs, const std::vector<int>& sigidx,
                                 lmm_post_t** out) {
  if ((long long)m * n > 2000000000LL / 64) return fail(LMM_ERR_UNSUPPORTED, "m*n too large for the dense path");
  hipStream_t st0 = g.streams[0];
  std::vector<LatentDev> lat(m);
  for (int l = 0; l < m; ++l) lat[l] = to_dev(gps[l]);
  const int N = m * n;
  Dims D(N, 1);
  lmm_post* P = new lmm_post();
  try {
    P->kind = 1; P->f32 = g_f32; P->n = n; P->d = d; P->l0 = 0; P->l1 = m; P->m = m; P->p = p;
    P->NC = D.NC; P->NR = D.NR; P->ld = D.ld;
    P->gps.assign(gps, gps + m);
    P->H.assign(H, H + (size_t)p * m);
    P->sigs = sigs; P->sigidx = sigidx;
    P->x = Buf<double>((size_t)d * n);
    HIPCHK(hipMemcpyAsync(P->x.p, xd, (size_t)d * n * sizeof(double), hipMemcpyDeviceToDevice, st0));
    P->latd = Buf<LatentDev>(m);
    HIPCHK(hipMemcpyAsync(P->latd.p, lat.data(), m * sizeof(LatentDev), hipMemcpyHostToDevice, st0));
    P->ddelta = Buf<double>((size_t)N);
    HIPCHK(hipMemcpyAsync(P->ddelta.p, delta, (size_t)N * sizeof(double), hipMemcpyDeviceToDevice, st0));
    Uploaded STd(sigs, st0);
    Buf<int> idxd(n);
    HIPCHK(hipMemcpyAsync(idxd.p, sigidx.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice, st0));
    P->L.emplace_back(mat_count(D.elems()));
    P->W.emplace_back(mat_count((size_t)(D.NC / 64) * 4096));
    P->alpha.emplace_back((size_t)D.NC);
    P->z.emplace_back((size_t)D.NC);          // z = L^-1 delta (the rider row): the fp32 mode's means are mu + R' z, not mu + K(x*, x) alpha
    Buf<int> info(1);
    HIPCHK(hipMemsetAsync(info.p, 0, sizeof(int), st0));
    DenseArgs a{};
    a.A = P->L[0].p; a.ld = D.ld; a.nrows = D.NR; a.ncols = D.NC; a.x = P->x.p; a.d = d; a.n = n; a.m = m;
    a.lat = P->latd.p; a.sigmaT = STd.buf.p; a.sig_idx = idxd.p; a.rider = P->ddelta.p; a.rider_ld = N; a.nrider = 1;
    launch_dense_assemble(a, st0);
    potrf_rec(P->L[0].p, D.ld, D.NR, 0, D.NC, P->W[0].p, N, info.p, st0);
    HIPCHK(hipMemsetAsync(P->alpha[0].p, 0, (size_t)D.NC * sizeof(double), st0));
    launch_extract_row(P->L[0].p, D.ld, D.NC, N, P->alpha[0].p, st0);
    HIPCHK(hipMemcpyAsync(P->z[0].p, P->alpha[0].p, (size_t)D.NC * sizeof(double), hipMemcpyDeviceToDevice, st0));
    backsolve1(P->L[0].p, D.ld, P->W[0].p, D.NC / 64, P->alpha[0].p, st0);
    int hinfo = 0;
    HIPCHK(hipMemcpyAsync(&hinfo, info.p, sizeof(int), hipMemcpyDeviceToHost, st0));
    HIPCHK(hipStreamSynchronize(st0));
    if (int rc = check_info(std::vector<int>{hinfo}, 0)) { delete P; return rc; }
  } catch (int code) { drain_after_error(); delete P; return code; }
  *out = P;
  return LMM_OK;
}

// posterior(fx::FiniteGP{<:ILMM}, y), dense H: reference src/ilmm.jl:184-198.  One (mn) x (mn) factorisation kept on the
// device with alpha = C \ (Yproj - mean).
int lmm_ilmm_posterior_create(const double* x, int d, int n, const double* y, int p, const double* H, int m, double sigma2,
                              const lmm_gp_t* gps, const lmm_jitters_t* jit, lmm_post_t** out) {
  std::lock_guard<std::mutex> lk(g_mu);
  REQUIRE_INIT();
  LMM_TRY
  if (!x || !y || !H || !out || d <= 0 || n <= 0 || p <= 0 || m <= 0) return fail(LMM_ERR_ARG, "bad arguments");
  if (int rc = check_gps(gps, m)) return rc;
  if (!jit) jit = &kDefaultJit;
  hipStream_t st0 = g.streams[0];
  std::vector<double> T, ST;
  if (int rc = project_dense(H, p, m, sigma2, jit->project_jitter, T, ST, nullptr)) return rc;
  DevIn xd(x, (size_t)d * n, st0), yd(y, (size_t)n * p, st0);
  Uploaded Td(T, st0);
  std::vector<double> means(m);
  for (int l = 0; l < m; ++l) means[l] = gps[l].mean;
  Uploaded meansd(means, st0);
  Buf<double> delta((size_t)n * m);
  project_on_device(yd.p, n, p, Td.buf, m, 0, m, meansd.buf.p, delta.p, st0);
  return dense_posterior_build(xd.p, d, n, H, p, m, gps, delta.p, ST, std::vector<int>(n, 0), out);
  LMM_CATCH
}

// posterior(pi(x2, sigma2), y2) on the dense-H posterior ILMM (AbstractGPs.TestUtils on `pi`, reference test/ilmm.jl:34-37;
// src/ilmm.jl:184-198 applied to the PosteriorGP latent): the posterior of the PRIOR given both projected data sets, each
// with its own SigmaT (x) I noise.  Returns a NEW handle.
int lmm_ilmm_post_condition(const lmm_post_t* post, double sigma2, const double* x2, int d, int n2, const double* y2,
                            const lmm_jitters_t* jit, lmm_post_t** out) {
  std::lock_guard<std::mutex> lk(g_mu);
  REQUIRE_INIT();
  LMM_TRY
  if (post && post->f32 != g_f32) return fail(LMM_ERR_ARG, "posterior handle was built in the other compute dtype (lmm_set_compute_dtype)");
  if (!post || !x2 || !y2 || !out || d <= 0 || n2 <= 0) return fail(LMM_ERR_ARG, "bad arguments");
  const lmm_post* P = post;
  if (P->kind != 1) return fail(LMM_ERR_ARG, "not a dense-H ILMM posterior");
  const lmm_post* D = dense_state(P);
  if (P->d != d) return fail(LMM_ERR_DIM, "input dimension mismatch");
  if (!jit) jit = &kDefaultJit;
  hipStream_t st0 = g.streams[0];
  const int m = P->m, p = P->p, n1 = P->n, n = n1 + n2;
  std::vector<double> T, ST;
  if (int rc = project_dense(P->H.data(), p, m, sigma2, jit->project_jitter, T, ST, nullptr)) return rc;
  DevIn x2d(x2, (size_t)d * n2, st0), y2d(y2, (size_t)n2 * p, st0);
  Uploaded Td(T, st0);
  std::vector<double> means(m);
  for (int l = 0; l < m; ++l) means[l] = P->gps[l].mean;
  Uploaded meansd(means, st0);
  Buf<double> d2((size_t)n2 * m), delta((size_t)n * m), xall((size_t)d * n);
  project_on_device(y2d.p, n2, p, Td.buf, m, 0, m, meansd.buf.p, d2.p, st0);
  HIPCHK(hipMemcpyAsync(xall.p, D->x.p, (size_t)d * n1 * sizeof(double), hipMemcpyDeviceToDevice, st0));
  HIPCHK(hipMemcpyAsync(xall.p + (size_t)d * n1, x2d.p, (size_t)d * n2 * sizeof(double), hipMemcpyDeviceToDevice, st0));
  for (int l = 0; l < m; ++l) {
    HIPCHK(hipMemcpyAsync(delta.p + (size_t)l * n, D->ddelta.p + (size_t)l * n1, (size_t)n1 * sizeof(double), hipMemcpyDeviceToDevice, st0));
    HIPCHK(hipMemcpyAsync(delta.p + (size_t)l * n + n1, d2.p + (size_t)l * n2, (size_t)n2 * sizeof(double), hipMemcpyDeviceToDevice, st0));
  }
  std::vector<double> sigs = P->sigs;
  sigs.insert(sigs.end(), ST.begin(), ST.end());
  std::vector<int> idx = P->sigidx;
  idx.resize(n, (int)(P->sigs.size() / ((size_t)m * m)));
  return dense_posterior_build(xall.p, d, n, P->H.data(), p, m, P->gps.data(), delta.p, sigs, idx, out);
  LMM_CATCH
}

static void dense_post_cross(const lmm_post* P, const double* xsd, int d, int ns, int nr, double* R, int ldr, hipStream_t st);
static void dense_post_means(const lmm_post* P, const double* xsd, int d, int ns, const double* R, int ldr, double* ml, hipStream_t st);

int lmm_ilmm_post_mean_and_var(const lmm_post_t* post, double sigma2, const double* xs, int d, int ns,
                               const lmm_jitters_t* jit, double* mean_out, double* var_out) {
  std::lock_guard<std::mutex> lk(g_mu);
  REQUIRE_INIT();
  LMM_TRY
  if (post && post->f32 != g_f32) return fail(LMM_ERR_ARG, "posterior handle was built in the other compute dtype (lmm_set_compute_dtype)");
  if (!post || !xs || !mean_out || !var_out || d <= 0 || ns <= 0) return fail(LMM_ERR_ARG, "bad arguments");
  const lmm_post* P = post;
  if (P->kind != 1) return fail(LMM_ERR_ARG, "not a dense-H ILMM posterior");
  const lmm_post* D = dense_state(P);
  if (P->d != d) return fail(LMM_ERR_DIM, "input dimension mismatch");
  if (!jit) jit = &kDefaultJit;
  hipStream_t st0 = g.streams[0];
  const int m = P->m, p = P->p, n = P->n, N = m * n;
  DevIn xsd(xs, (size_t)d * ns, st0);
  Uploaded Hd(P->H, st0);
  Buf<double> ml((size_t)ns * m);
  const int nr = rup(m * ns, 64);
  int ldr = nr; if ((ldr % 512) == 0) ldr += 16;
  Buf<double> R(mat_count((size_t)ldr * P->NC));
  dense_post_cross(P, xsd.p, d, ns, nr, R.p, ldr, st0);
  dense_post_means(P, xsd.p, d, ns, R.p, ldr, ml.p, st0);
  DevOut mo(mean_out, (size_t)ns * p), vo(var_out, (size_t)ns * p);
  launch_mix(ml.p, ns, m, Hd.buf.p, p, 1, 0.0, 0.0, nullptr, 0.0, mo.p, st0);
  Buf<double> dv_part(dense_var_partial_elems(ns, p, N));
  launch_dense_var(R.p, ldr, ns, m, N, Hd.buf.p, p, D->latd.p, jit->default_jitter, sigma2, dv_part.p, vo.p, st0);
  mo.finish(st0); vo.finish(st0);
  HIPCHK(hipStreamSynchronize(st0));
  return LMM_OK;
  LMM_CATCH
}

// Dense-H posterior: R (nr x NC, ldr; rows (l, s) = l ns + s, rows >= m ns zero) = K(xs, x)' L^-T.  Caller holds g_mu.
static void dense_post_cross(const lmm_post* P, const double* xsd, int d, int ns, int nr, double* R, int ldr, hipStream_t st) {
  const lmm_post* D = dense_state(P);
  guard_extent(R, nr, ldr, P->NC, true, "dense-H cross-Gram");
  launch_dense_cross(R, ldr, nr, P->NC, xsd, ns, D->x.p, P->n, d, P->m, D->latd.p, st);
  trsm_rec(R, ldr, nr, D->L[0].p, P->ld, D->W[0].p, 0, P->NC, st);
}
// Latent posterior means at xs, ml[l ns + s].  Float64: mu_l + K(x*, x) alpha_l (no solve needed).  fp32 compute mode: that sum cancels
// over weights alpha = Kt^-1 delta whose Float32-factor error is amplified by cond |alpha| (section 4.3 of DESIGN.md: 0.15 absolute at
// n = 1100 on the per-latent path), so the means take the rider form mu_l + R (L^-1 delta) from the cross-solve block R (which the
// caller has computed: dense_post_cross; rows >= m ns of R are zero).
static void dense_post_means(const lmm_post* P, const double* xsd, int d, int ns, const double* R, int ldr, double* ml, hipStream_t st) {
  const lmm_post* D = dense_state(P);
  const int m = P->m, n = P->n;
  if (!g_f32) {
    Buf<double> pm_part(post_mean_partial_elems(ns, n));
    for (int l = 0; l < m; ++l)
      launch_post_mean(xsd, ns, D->x.p, n, d, D->alpha[0].p + (size_t)l * n, to_dev(P->gps[l]), pm_part.p, ml + (size_t)l * ns, st);
    HIPCHK(hipStreamSynchronize(st));              // pm_part is released on return
    return;
  }
  Buf<double> part(strip_partial_elems(m * ns, m * n, 1)), mu((size_t)m * ns);
  rider_stats_g(R, ldr, m * ns, m * n, D->z[0].p, 0.0, 0.0, part.p, ml, nullptr, st);
  for (int l = 0; l < m; ++l) launch_fill(mu.p + (size_t)l * ns, ns, P->gps[l].mean, st);
  launch_vec_lin(ml, mu.p, 1.0, m * ns, ml, st);
  HIPCHK(hipStreamSynchronize(st));
}
// Latent joint covariance at xs as a factor matrix,  blockdiag(K_l(xs,xs)) + SigAdd (x) I_ns - R R'  (R from dense_post_cross with
// nr = Ds.NC rows), optional rider row, then its Cholesky.
static void dense_post_cov_factor(const lmm_post* P, const double* xsd, int d, int ns, const double* sigadd_dev,
                                  const double* rider, const Dims& Ds, double* A, double* WA, const double* R, int ldr, int* info,
                                  hipStream_t st, bool factor = true) {
  const int m = P->m;
  const lmm_post* D = dense_state(P);
  DenseArgs a{};
  a.A = A; a.ld = Ds.ld; a.nrows = Ds.NR; a.ncols = Ds.NC; a.x = xsd; a.d = d; a.n = ns; a.m = m;
  a.lat = D->latd.p; a.sigmaT = sigadd_dev; a.rider = rider; a.rider_ld = m * ns; a.nrider = rider ? 1 : 0;
  guard_extent(A, Ds.NR, Ds.ld, Ds.NC, true, "dense-H posterior covariance");
  launch_dense_assemble(a, st);
  gemm_nt_g(A, Ds.ld, R, ldr, R, ldr, Ds.NC, Ds.NC, P->NC, 1, false, st, "Schur complement (dense-H posterior covariance)");
  if (factor) potrf_rec(A, Ds.ld, Ds.NR, 0, Ds.NC, WA, m * ns, info, st);
}

// mean_and_cov(pi(xs, sigma2)) / cov on the dense-H posterior ILMM (reference src/ilmm.jl:132-147 with the PosteriorGP latent
// of :196-197; AbstractGPs.TestUtils secondary interface on `pi`, test/ilmm.jl:34-37): C = H_full (Cov_latent + 1e-18 I)
// H_full' + sigma2 I, (p ns) x (p ns) column-major, by-outputs order.
int lmm_ilmm_post_mean_and_cov(const lmm_post_t* post, double sigma2, const double* xs, int d, int ns,
                               const lmm_jitters_t* jit, double* mean_out, double* cov_out) {
  std::lock_guard<std::mutex> lk(g_mu);
  REQUIRE_INIT();
  LMM_TRY
  if (post && post->f32 != g_f32) return fail(LMM_ERR_ARG, "posterior handle was built in the other compute dtype (lmm_set_compute_dtype)");
  if (!post || !xs || !mean_out || !cov_out || d <= 0 || ns <= 0) return fail(LMM_ERR_ARG, "bad arguments");
  const lmm_post* P = post;
  if (P->kind != 1) return fail(LMM_ERR_ARG, "not a dense-H ILMM posterior");
  const lmm_post* D = dense_state(P);
  if (P->d != d) return fail(LMM_ERR_DIM, "input dimension mismatch");
  if (!jit) jit = &kDefaultJit;
  const int m = P->m, p = P->p, n = P->n, Ns = m * ns;
  if ((double)p * ns * (double)p * ns > 4e8) return fail(LMM_ERR_UNSUPPORTED, "full covariance (p*ns)^2 too large");
  hipStream_t st0 = g.streams[0];
  DevIn xsd(xs, (size_t)d * ns, st0);
  Uploaded Hd(P->H, st0), Zd(std::vector<double>((size_t)m * m, 0.0), st0);
  Buf<double> ml((size_t)Ns);
  Dims Ds(Ns, 0);
  int ldr = Ds.NC; if ((ldr % 512) == 0) ldr += 16;
  Buf<double> A(mat_count(Ds.elems())), R(mat_count((size_t)ldr * P->NC)), T((size_t)p * ns * Ns);
  dense_post_cross(P, xsd.p, d, ns, Ds.NC, R.p, ldr, st0);
  dense_post_means(P, xsd.p, d, ns, R.p, ldr, ml.p, st0);
  dense_post_cov_factor(P, xsd.p, d, ns, Zd.buf.p, nullptr, Ds, A.p, nullptr, R.p, ldr, nullptr, st0, false);
  DevOut mo(mean_out, (size_t)ns * p), co(cov_out, (size_t)ns * p * ns * p);
  launch_mix(ml.p, ns, m, Hd.buf.p, p, 1, 0.0, 0.0, nullptr, 0.0, mo.p, st0);
  launch_dense_cov(A.p, Ds.ld, ns, m, Hd.buf.p, p, jit->default_jitter, sigma2, T.p, co.p, st0);
  mo.finish(st0); co.finish(st0);
  HIPCHK(hipStreamSynchronize(st0));
  return LMM_OK;
  LMM_CATCH
}

// logpdf(pi(xs, sigma2), ys) on the dense-H posterior ILMM (reference test/ilmm.jl:25; src/ilmm.jl:150-163 with the
// PosteriorGP latent of :196-197): project ys, one (m ns) x (m ns) factorisation of latent posterior cov + SigmaT (x) I.
int lmm_ilmm_post_logpdf(const lmm_post_t* post, double sigma2, const double* xs, int d, int ns, const double* ys,
                         const lmm_jitters_t* jit, double* out) {
  std::lock_guard<std::mutex> lk(g_mu);
  REQUIRE_INIT();
  LMM_TRY
  if (post && post->f32 != g_f32) return fail(LMM_ERR_ARG, "posterior handle was built in the other compute dtype (lmm_set_compute_dtype)");
  if (!post || !xs || !ys || !out || d <= 0 || ns <= 0) return fail(LMM_ERR_ARG, "bad arguments");
  const lmm_post* P = post;
  if (P->kind != 1) return fail(LMM_ERR_ARG, "not a dense-H ILMM posterior");
  const lmm_post* D = dense_state(P);
  if (P->d != d) return fail(LMM_ERR_DIM, "input dimension mismatch");
  if (!jit) jit = &kDefaultJit;
  hipStream_t st0 = g.streams[0];
  const int m = P->m, p = P->p, n = P->n, Ns = m * ns;
  std::vector<double> T, ST;
  double logdetST = 0.0;
  if (int rc = project_dense(P->H.data(), p, m, sigma2, jit->project_jitter, T, ST, &logdetST)) return rc;
  DevIn xsd(xs, (size_t)d * ns, st0), ysd(ys, (size_t)ns * p, st0);
  Uploaded Td(T, st0), STd(ST, st0), Hd(P->H, st0);
  Buf<double> Ty((size_t)ns * m), ml((size_t)ns * m), delta((size_t)ns * m), partial(tall_skinny_partials(ns, p)), resid_dev(1);
  project_on_device(ysd.p, ns, p, Td.buf, m, 0, m, nullptr, Ty.p, st0);
  residual_on_device(ysd.p, ns, p, Ty.p, m, Hd.buf, partial.p, resid_dev.p, st0);
  Dims Ds(Ns, 1);
  int ldr = Ds.NC; if ((ldr % 512) == 0) ldr += 16;
  Buf<double> A(mat_count(Ds.elems())), WA(mat_count((size_t)(Ds.NC / 64) * 4096)), R(mat_count((size_t)ldr * P->NC)), lml_dev(1);
  dense_post_cross(P, xsd.p, d, ns, Ds.NC, R.p, ldr, st0);
  dense_post_means(P, xsd.p, d, ns, R.p, ldr, ml.p, st0);
  launch_vec_lin(Ty.p, ml.p, -1.0, Ns, delta.p, st0);
  Buf<int> info(1);
  HIPCHK(hipMemsetAsync(info.p, 0, sizeof(int), st0));
  dense_post_cov_factor(P, xsd.p, d, ns, STd.buf.p, delta.p, Ds, A.p, WA.p, R.p, ldr, info.p, st0);
  launch_lml_reduce(A.p, Ds.ld, Ns, Ds.NC, 1, lml_dev.p, st0);
  double lml = 0.0, resid = 0.0;
  int hinfo = 0;
  HIPCHK(hipMemcpyAsync(&lml, lml_dev.p, sizeof(double), hipMemcpyDeviceToHost, st0));
  HIPCHK(hipMemcpyAsync(&resid, resid_dev.p, sizeof(double), hipMemcpyDeviceToHost, st0));
  HIPCHK(hipMemcpyAsync(&hinfo, info.p, sizeof(int), hipMemcpyDeviceToHost, st0));
  HIPCHK(hipStreamSynchronize(st0));
  if (int rc = check_info(std::vector<int>{hinfo}, 0)) return rc;
  *out = lml - ((double)ns * ((double)(p - m) * kLog2Pi + ((double)p * std::log(sigma2) - logdetST)) + resid / sigma2) / 2.0;
  return LMM_OK;
  LMM_CATCH
}

// rand(rng, pi(xs, sigma2)) on the dense-H posterior ILMM (reference src/ilmm.jl:78-87 with the PosteriorGP latent): the
// latent joint sample mean + chol(Cov + 1e-12 I).U' z  (z: m*ns normals), mixed by H, plus sqrt(sigma2) eps.
int lmm_ilmm_post_rand(const lmm_post_t* post, double sigma2, int add_noise, const double* xs, int d, int ns,
                       const double* z_lat, const double* eps, const lmm_jitters_t* jit, double* out) {
  std::lock_guard<std::mutex> lk(g_mu);
  REQUIRE_INIT();
  LMM_TRY
  if (post && post->f32 != g_f32) return fail(LMM_ERR_ARG, "posterior handle was built in the other compute dtype (lmm_set_compute_dtype)");
  if (!post || !xs || !z_lat || !out || d <= 0 || ns <= 0 || (add_noise && !eps)) return fail(LMM_ERR_ARG, "bad arguments");
  const lmm_post* P = post;
  if (P->kind != 1) return fail(LMM_ERR_ARG, "not a dense-H ILMM posterior");
  const lmm_post* D = dense_state(P);
  if (P->d != d) return fail(LMM_ERR_DIM, "input dimension mismatch");
  if (!jit) jit = &kDefaultJit;
  hipStream_t st0 = g.streams[0];
  const int m = P->m, p = P->p, n = P->n, Ns = m * ns;
  std::vector<double> J((size_t)m * m, 0.0);
  for (int l = 0; l < m; ++l) J[l + (size_t)l * m] = jit->ilmm_rand_jitter;
  DevIn xsd(xs, (size_t)d * ns, st0), zd(z_lat, (size_t)Ns, st0), epsd(add_noise ? eps : nullptr, (size_t)ns * p, st0);
  Uploaded Jd(J, st0), Hd(P->H, st0);
  Buf<double> ml((size_t)Ns), X((size_t)Ns);
  Dims Ds(Ns, 0);
  int ldr = Ds.NC; if ((ldr % 512) == 0) ldr += 16;
  Buf<double> A(mat_count(Ds.elems())), WA(mat_count((size_t)(Ds.NC / 64) * 4096)), R(mat_count((size_t)ldr * P->NC)), part(strip_partial_elems(Ns, Ns, 1));
  dense_post_cross(P, xsd.p, d, ns, Ds.NC, R.p, ldr, st0);
  dense_post_means(P, xsd.p, d, ns, R.p, ldr, ml.p, st0);
  Buf<int> info(1);
  HIPCHK(hipMemsetAsync(info.p, 0, sizeof(int), st0));
  dense_post_cov_factor(P, xsd.p, d, ns, Jd.buf.p, nullptr, Ds, A.p, WA.p, R.p, ldr, info.p, st0);
  launch_trmv_lower(A.p, Ds.ld, Ns, zd.p, 0.0, part.p, X.p, st0);
  launch_vec_lin(X.p, ml.p, 1.0, Ns, X.p, st0);
  DevOut od(out, (size_t)ns * p);
  launch_mix(X.p, ns, m, Hd.buf.p, p, 1, 0.0, 0.0, add_noise ? epsd.p : nullptr, std::sqrt(sigma2), od.p, st0);
  od.finish(st0);
  int hinfo = 0;
  HIPCHK(hipMemcpyAsync(&hinfo, info.p, sizeof(int), hipMemcpyDeviceToHost, st0));
  HIPCHK(hipStreamSynchronize(st0));
  return check_info(std::vector<int>{hinfo}, 0);
  LMM_CATCH
}

// get_latent_gp(posterior(fx::FiniteGP{<:ILMM}, y)) for a dense H (reference src/ilmm.jl:39 on the ILMM of :196-197): the latent
// PosteriorGP{IndependentMOGP} as a handle of its own.  It SHARES the device state of `post` (the (mn) x (mn) factor, alpha, x)
// and differs only in H = I_m, p = m, so every lmm_ilmm_post_* entry point answers for the m latent outputs at
// MOInputIsotopicByOutputs(xs, m): with project_jitter = 0 the projection is the identity, SigmaT = sigma2 I and the regulariser
// vanishes, i.e. logpdf is the generic Gaussian of the latent posterior + sigma2 I; rand with ilmm_rand_jitter = sigma2 and
// add_noise = 0 is AbstractGPs' mean + chol(cov + sigma2 I).U' z.  Destroy it with lmm_post_destroy (either order w.r.t. `post`).
int lmm_ilmm_post_latent_view(const lmm_post_t* post, lmm_post_t** out) {
  std::lock_guard<std::mutex> lk(g_mu);
  REQUIRE_INIT();
  LMM_TRY
  if (!post || !out) return fail(LMM_ERR_ARG, "bad arguments");
  if (post->kind != 1) return fail(LMM_ERR_ARG, "not a dense-H ILMM posterior");
  lmm_post* B = const_cast<lmm_post*>(dense_state(post));
  lmm_post* V = new lmm_post();
  V->f32 = B->f32; V->kind = 1; V->n = B->n; V->d = B->d; V->l0 = 0; V->l1 = B->m; V->m = B->m; V->p = B->m;
  V->NC = B->NC; V->NR = B->NR; V->ld = B->ld;
  V->gps = B->gps; V->sigs = B->sigs; V->sigidx = B->sigidx;
  V->H.assign((size_t)B->m * B->m, 0.0);
  for (int l = 0; l < B->m; ++l) V->H[l + (size_t)l * B->m] = 1.0;
  V->base = B;
  ++B->views;
  *out = V;
  return LMM_OK;
  LMM_CATCH
}

int lmm_post_destroy(lmm_post_t* post) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (post) {
    if (g.init) (void)hipDeviceSynchronize();
    if (post->base) {                                   // a latent view: release the base if it was waiting for its last view
      lmm_post* B = post->base;
      if (--B->views == 0 && B->zombie) delete B;
      delete post;
    } else if (post->views > 0) {
      post->zombie = true;                              // views still use this state: freed with the last of them
    } else {
      delete post;
    }
  }
  return LMM_OK;
}

// Rk (nsr x NC, ldr) = K(xs, x): the cross-Gram of a posterior latent's training inputs as rider rows (rows beyond ns zero).
static GramArgs cross_gram_args(const lmm_post* P, const lmm_gp_t& gp, const double* xsd, int d, int ns, double* Rk, int ldr,
                                int nsr) {
  GramArgs r{};
  r.A = Rk; r.ld = ldr; r.nrows = P->NC + nsr; r.ncols = P->NC; r.row_tile0 = P->NC / 64; r.row_shift = P->NC; r.full = 1;
  r.x = P->x.p; r.d = d; r.n = P->n; r.kind = gp.kind; r.var = gp.variance; r.inv_ls = 1.0 / gp.lengthscale;
  r.xs = xsd; r.ns = ns;
  return r;
}
static void cross_gram(const lmm_post* P, const lmm_gp_t& gp, const double* xsd, int d, int ns, double* Rk, int ldr, int nsr,
                       hipStream_t st) {
  gram_g(cross_gram_args(P, gp, xsd, d, ns, Rk, ldr, nsr), st);
}

// Latent marginals (mean, var) of latents [l0, l1) at xs into device arrays (ns per latent).
// post != NULL: posterior latents; else prior latents gps[l0..l1).  Caller holds g_mu.
static int latent_marginals_dev(const lmm_post* P, const lmm_gp_t* gps_shard, int ms, const double* xsd, int d, int ns,
                                double* mean_lat, double* var_lat) {
  if (ms == 0) return LMM_OK;
  if (P == nullptr) {
    fork_slots(1);
    for (int k = 0; k < ms; ++k) {
      LatentDev gd = to_dev(gps_shard[k]);
      // prior: constant mean, variance kappa(0)
      rider_stats_g(nullptr, 0, ns, 0, nullptr, gd.mean, gps_shard[k].variance, nullptr, mean_lat + (size_t)k * ns,
                         var_lat + (size_t)k * ns, g.streams[0]);
    }
    return LMM_OK;
  }
  if (P->d != d) return fail(LMM_ERR_DIM, "input dimension mismatch: posterior has d=%d, xs has d=%d", P->d, d);
  const int nsr = rup(ns, 64);
  int ldr = nsr; if ((ldr % 512) == 0) ldr += 16;
  int nb_per = 1, nslots = 1;
  batch_plan(ms, &nb_per, &nslots, mat_bytes((double)ldr * P->NC));
  std::vector<std::vector<Buf<double>>> R(nslots);
  std::vector<Buf<double>> part;
  for (int s = 0; s < nslots; ++s) {
    for (int j = 0; j < nb_per; ++j) R[s].emplace_back(mat_count((size_t)ldr * P->NC));
    part.emplace_back(strip_partial_elems(nsr, P->NC, 2));
  }
  fork_slots(nslots);
  int bi = 0;
  for (int k0 = 0; k0 < ms; k0 += nb_per, ++bi) {
    const int s = bi % nslots, nb = std::min(nb_per, ms - k0);
    hipStream_t st = g.streams[s];
    BatchPtr Rb{}, Lb{}, Wb{};
    GramArgs ga[LMM_MAX_BATCH];
    for (int j = 0; j < nb; ++j) {
      const int k = k0 + j;
      ga[j] = cross_gram_args(P, P->gps[P->l0 + k], xsd, d, ns, R[s][j].p, ldr, nsr);
      Rb.p[j] = R[s][j].p; Lb.p[j] = P->L[k].p; Wb.p[j] = P->W[k].p;
    }
    {
      const double gb = (double)ns * P->n * 8.0;                   // cross-Gram K(x*, x): a full n* x n rectangle written once
      ProfScope ps(LMM_PROF_GRAM, nb * gb, st, 0, 0, 0, nb * gb, nb);
      gram_batch_g(ga, nb, st);
    }
    trsm_rec(Rb, ldr, nsr, Lb, P->ld, Wb, nb, 0, P->NC, st);       // R_j <- K(x*, x) L_j^-T for the whole batch
    for (int j = 0; j < nb; ++j) {
      const int k = k0 + j;
      const lmm_gp_t& gp = P->gps[P->l0 + k];
      // mean = mu + K(x*,x) alpha = mu + R' (L^-1 delta);  var = kappa(0) - colsumsq(R)   (one pass over R)
      const double rb = (double)ns * P->n * 8.0;                   // R read once
      ProfScope ps(LMM_PROF_STRIP, rb, st, ns, P->n, 0, rb);
      rider_stats_g(R[s][j].p, ldr, ns, P->n, P->z[k].p, gp.mean, gp.variance, part[s].p, mean_lat + (size_t)k * ns,
                         var_lat + (size_t)k * ns, st);
    }
  }
  join_slots(nslots);
  HIPCHK(hipStreamSynchronize(g.streams[0]));   // R buffers are released on return
  return LMM_OK;
}

int lmm_latent_marginals(const lmm_post_t* post, const lmm_gp_t* gps, int m_shard, const double* xs, int d, int ns,
                         double* mean_lat, double* var_lat) {
  std::lock_guard<std::mutex> lk(g_mu);
  REQUIRE_INIT();
  LMM_TRY
  if (post && post->f32 != g_f32) return fail(LMM_ERR_ARG, "posterior handle was built in the other compute dtype (lmm_set_compute_dtype)");
  if (!xs || !mean_lat || !var_lat || d <= 0 || ns <= 0) return fail(LMM_ERR_ARG, "bad arguments");
  if (post && post->kind != 0) return fail(LMM_ERR_UNSUPPORTED, "per-latent marginals of the dense-H posterior (coupled latents): use lmm_ilmm_post_mean_and_var");
  const int ms = post ? (post->l1 - post->l0) : m_shard;
  if (!post) { if (int rc = check_gps(gps, m_shard)) return rc; }
  hipStream_t st0 = g.streams[0];
  DevIn xsd(xs, (size_t)d * ns, st0);
  DevOut mo(mean_lat, (size_t)ns * ms), vo(var_lat, (size_t)ns * ms);
  if (int rc = latent_marginals_dev(post, gps, ms, xsd.p, d, ns, mo.p, vo.p)) return rc;
  mo.finish(st0); vo.finish(st0);
  HIPCHK(hipStreamSynchronize(st0));
  return LMM_OK;
  LMM_CATCH
}

int lmm_oilmm_mean_and_var(const lmm_post_t* post, const lmm_gp_t* gps, const double* U, const double* S, int p, int m,
                           int latent_begin, int latent_end, double sigma2, int add_noise, const double* xs, int d,
                           int ns, const lmm_jitters_t* jit, double* mean_out, double* var_out) {
  std::lock_guard<std::mutex> lk(g_mu);
  REQUIRE_INIT();
  LMM_TRY
  if (post && post->f32 != g_f32) return fail(LMM_ERR_ARG, "posterior handle was built in the other compute dtype (lmm_set_compute_dtype)");
  if (!U || !xs || !mean_out || d <= 0 || ns <= 0 || p <= 0 || m <= 0) return fail(LMM_ERR_ARG, "bad arguments");
  if (post && post->kind != 0) return fail(LMM_ERR_UNSUPPORTED, "dense-H posterior handle: use lmm_ilmm_post_mean_and_var");
  if (!jit) jit = &kDefaultJit;
  int l0 = latent_begin, l1 = latent_end;
  if (post) { l0 = post->l0; l1 = post->l1; if (post->m != m) return fail(LMM_ERR_DIM, "posterior has %d latents, H has %d", post->m, m); }
  else if (int rc = check_gps(gps, m)) return rc;
  if (l0 < 0 || l1 > m || l0 > l1) return fail(LMM_ERR_ARG, "bad latent shard");
  const int ms = l1 - l0;
  hipStream_t st0 = g.streams[0];
  // H columns of the shard (p x ms)
  std::vector<double> Hs((size_t)p * std::max(ms, 1), 0.0);
  for (int k = 0; k < ms; ++k)
    for (int o = 0; o < p; ++o) Hs[o + (size_t)k * p] = U[o + (size_t)(l0 + k) * p] * (S ? std::sqrt(S[l0 + k]) : 1.0);
  Uploaded Hd(Hs, st0);
  DevIn xsd(xs, (size_t)d * ns, st0);
  Buf<double> ml((size_t)ns * std::max(ms, 1)), vl((size_t)ns * std::max(ms, 1));
  // fp32 mode: mu + K(x*, x) alpha is a cancelling sum over weights alpha = Kt^-1 delta whose Float32-factor error is amplified
  // (cond x eps x |alpha|); the rider form mu + R' (L^-1 delta) of the full path is stable, so posterior means take that path
  const bool mean_only = var_out == nullptr && !(g_f32 && post);
  if (mean_only) {
    // mean only (AbstractGPs.mean(fx), reference src/ilmm.jl:142 -> mean_and_var(fx)[1]): the posterior latent means are
    // mu + K(x*, x) alpha -- n n* kernel evaluations, no triangular solve (the reference pays for the variances it discards)
    if (post && post->d != d) return fail(LMM_ERR_DIM, "input dimension mismatch: posterior has d=%d, xs has d=%d", post->d, d);
    Buf<double> pm_part(post ? post_mean_partial_elems(ns, post->n) : 1);
    for (int k = 0; k < ms; ++k) {
      const lmm_gp_t& gp = post ? post->gps[l0 + k] : gps[l0 + k];
      launch_post_mean(xsd.p, ns, post ? post->x.p : nullptr, post ? post->n : 0, d, post ? post->alpha[k].p : nullptr, to_dev(gp),
                       pm_part.p, ml.p + (size_t)k * ns, st0);
    }
    DevOut mo(mean_out, (size_t)ns * p);
    mix_marginals(ml.p, ns, ms, Hd.buf.p, p, 1, 0.0, 0.0, mo.p, st0);
    mo.finish(st0);
    HIPCHK(hipStreamSynchronize(st0));
    return LMM_OK;
  }
  if (int rc = latent_marginals_dev(post, post ? nullptr : gps + l0, ms, xsd.p, d, ns, ml.p, vl.p)) return rc;
  DevOut mo(mean_out, (size_t)ns * p), vo(var_out, (size_t)ns * p);
  // reference src/oilmm.jl:69,72: M = H M_latent;  V = abs2.(H) V_latent .+ sigma2   (V_latent carries the 1e-18 jitter);
  // Float64 VALU by default, v_mfma_f32_16x16x32_bf16 under lmm_set_projection_dtype(LMM_PROJ_BF16 / _BF16X2)
  mix_marginals(ml.p, ns, ms, Hd.buf.p, p, 1, 0.0, 0.0, mo.p, st0);
  if (var_out) mix_marginals(vl.p, ns, ms, Hd.buf.p, p, 2, jit->default_jitter, add_noise ? sigma2 : 0.0, vo.p, st0);
  mo.finish(st0); vo.finish(st0);
  HIPCHK(hipStreamSynchronize(st0));
  return LMM_OK;
  LMM_CATCH
}

// R (nsr x NC, ldr) = K(xs, x) L^-T for latent k of the posterior: the riders of the cross-Gram solved against the factor.
static void cross_solve(const lmm_post* P, int k, const lmm_gp_t& gp, const double* xsd, int d, int ns, double* Rk, int ldr,
                        int nsr, hipStream_t st) {
  cross_gram(P, gp, xsd, d, ns, Rk, ldr, nsr, st);
  trsm_rec(Rk, ldr, nsr, P->L[k].p, P->ld, P->W[k].p, 0, P->NC, st);
}

// Per-latent posterior (or prior) covariance at xs as a factor matrix B (NRs x NCs): gram(xs) + diag_add - R R' (posterior,
// R from cross_solve), rider row = rider_vec.  Not factorised here (the caller batches potrf_rec).  Caller holds g_mu.
static GramArgs cov_args(const lmm_gp_t& gp, const double* xsd, int d, int ns, double diag_add, const double* rider_vec,
                         const Dims& Ds, double* B) {
  GramArgs a{};
  a.A = B; a.ld = Ds.ld; a.nrows = Ds.NR; a.ncols = Ds.NC; a.x = xsd; a.d = d; a.n = ns;
  a.kind = gp.kind; a.var = gp.variance; a.inv_ls = 1.0 / gp.lengthscale; a.diag_add = diag_add; a.pad_diag = 1.0;
  a.rider = rider_vec; a.rider_ld = ns; a.nrider = rider_vec ? 1 : 0;
  return a;
}
static void cov_at_xs(const lmm_post* P, const lmm_gp_t& gp, const double* xsd, int d, int ns, double diag_add,
                      const double* rider_vec, const Dims& Ds, double* B, const double* Rk, int ldr, hipStream_t st) {
  gram_g(cov_args(gp, xsd, d, ns, diag_add, rider_vec, Ds, B), st);
  // Schur complement on the leading NCs x NCs block (rows of R beyond ns are zero)
  if (P != nullptr) gemm_nt_g(B, Ds.ld, Rk, ldr, Rk, ldr, Ds.NC, Ds.NC, P->NC, 1, false, st, "Schur complement (posterior covariance at xs)");
}
// The same for the nb latents of a batch: one Gram launch per run of equal kinds, ONE batched Schur-complement GEMM.
static void cov_at_xs_batch(const lmm_post* P, const GramArgs* ga, int nb, const Dims& Ds, const BatchPtr& Bb, const BatchPtr& Rb,
                            int ldr, hipStream_t st) {
  gram_batch_g(ga, nb, st);
  if (P != nullptr) gemm_nt_g(Bb, 0, Ds.ld, Rb, 0, ldr, Rb, 0, ldr, Ds.NC, Ds.NC, P->NC, 1, false, nb, st, "batched Schur complement (posterior covariance at xs)");
}

// Working buffers of the batched "covariance at xs" loops (rand, posterior logpdf): per stream slot nb_per factor matrices
// with their inverse blocks, means, riders and cross-solve blocks R, and one reduction scratch (reused latent after latent in
// stream order).
struct XsSlots {
  int nb_per = 1, nslots = 1, ldr = 0, nsr = 0;
  std::vector<std::vector<Buf<double>>> B, WB, mu, rid, R;
  std::vector<Buf<double>> part;
  XsSlots(const lmm_post* P, int ms, int ns, const Dims& Ds) {
    nsr = Ds.NC;                     // the Schur complement reads Ds.NC rows of R (rows beyond ns are zero)
    ldr = nsr; if ((ldr % 512) == 0) ldr += 16;
    batch_plan(std::max(ms, 1), &nb_per, &nslots, mat_bytes((double)Ds.elems() + (P ? (double)ldr * P->NC : 0.0)));
    B.resize(nslots); WB.resize(nslots); mu.resize(nslots); rid.resize(nslots); R.resize(nslots);
    for (int s = 0; s < nslots; ++s) {
      for (int j = 0; j < nb_per; ++j) {
        B[s].emplace_back(mat_count(Ds.elems())); WB[s].emplace_back(mat_count((size_t)(Ds.NC / 64) * 4096));
        mu[s].emplace_back((size_t)ns); rid[s].emplace_back((size_t)ns);
        R[s].emplace_back(P ? mat_count((size_t)ldr * P->NC) : 1);
      }
      part.emplace_back(std::max(strip_partial_elems(nsr, P ? P->NC : 1, 1), strip_partial_elems(ns, ns, 1)));
    }
  }
  // R[s][j] <- K(xs, x) L_k^-T and mu[s][j] <- mean_k(xs) for the latents k0..k0+nb-1 of the posterior's shard (one batched solve)
  void cross_solve_batch(const lmm_post* P, int s, int k0, int nb, const double* xsd, int d, int ns, hipStream_t st) {
    BatchPtr Rb{}, Lb{}, Wb{};
    GramArgs ga[LMM_MAX_BATCH];
    for (int j = 0; j < nb; ++j) {
      ga[j] = cross_gram_args(P, P->gps[P->l0 + k0 + j], xsd, d, ns, R[s][j].p, ldr, nsr);
      Rb.p[j] = R[s][j].p; Lb.p[j] = P->L[k0 + j].p; Wb.p[j] = P->W[k0 + j].p;
    }
    gram_batch_g(ga, nb, st);
    trsm_rec(Rb, ldr, nsr, Lb, P->ld, Wb, nb, 0, P->NC, st);
    for (int j = 0; j < nb; ++j)
      rider_stats_g(R[s][j].p, ldr, ns, P->n, P->z[k0 + j].p, P->gps[P->l0 + k0 + j].mean, 0.0, part[s].p, mu[s][j].p, nullptr, st);
  }
};

extern "C" int lmm_lmm_mean_and_cov(const lmm_post_t* post, const lmm_gp_t* gps, const double* U, const double* S, int p, int m,
                                    int latent_begin, int latent_end, double sigma2, int add_noise, const double* xs, int d,
                                    int ns, const lmm_jitters_t* jit, double* mean_out, double* cov_out) {
  std::lock_guard<std::mutex> lk(g_mu);
  REQUIRE_INIT();
  LMM_TRY
  if (post && post->f32 != g_f32) return fail(LMM_ERR_ARG, "posterior handle was built in the other compute dtype (lmm_set_compute_dtype)");
  if (!U || !xs || !mean_out || !cov_out || d <= 0 || ns <= 0 || p <= 0 || m <= 0) return fail(LMM_ERR_ARG, "bad arguments");
  if ((double)p * ns * (double)p * ns > 4e8) return fail(LMM_ERR_UNSUPPORTED, "full covariance (p*ns)^2 too large");
  if (!jit) jit = &kDefaultJit;
  const lmm_post* P = post;
  int l0 = latent_begin, l1 = latent_end;
  if (P) {
    if (P->kind != 0) return fail(LMM_ERR_UNSUPPORTED, "full covariance of the dense-H posterior is not built");
    l0 = P->l0; l1 = P->l1;
    if (P->m != m) return fail(LMM_ERR_DIM, "posterior has %d latents, H has %d", P->m, m);
    if (P->d != d) return fail(LMM_ERR_DIM, "input dimension mismatch");
  } else if (int rc = check_gps(gps, m)) return rc;
  if (l0 < 0 || l1 > m || l0 > l1) return fail(LMM_ERR_ARG, "bad latent shard");
  const int ms = l1 - l0;
  hipStream_t st0 = g.streams[0];
  std::vector<double> Hs((size_t)p * std::max(ms, 1), 0.0);
  for (int k = 0; k < ms; ++k)
    for (int o = 0; o < p; ++o) Hs[o + (size_t)k * p] = U[o + (size_t)(l0 + k) * p] * (S ? std::sqrt(S[l0 + k]) : 1.0);
  Uploaded Hd(Hs, st0);
  DevIn xsd(xs, (size_t)d * ns, st0);
  DevOut mo(mean_out, (size_t)ns * p), co(cov_out, (size_t)ns * p * ns * p);
  Buf<double> ml((size_t)ns * std::max(ms, 1));
  Dims Ds(ns, 0);
  const int nsr = Ds.NC;           // the Schur complement reads Ds.NC rows of R
  int ldr = nsr; if ((ldr % 512) == 0) ldr += 16;
  const int CH = LMM_MAX_BATCH;
  std::vector<Buf<double>> Cm;
  for (int c = 0; c < std::min(CH, std::max(ms, 1)); ++c) Cm.emplace_back(mat_count(Ds.elems()));
  Buf<double> R(P ? mat_count((size_t)ldr * P->NC) : 1), part(strip_partial_elems(nsr, P ? P->NC : 1, 1));
  if (ms == 0) {
    HIPCHK(hipMemsetAsync(mo.p, 0, (size_t)ns * p * sizeof(double), st0));
    BatchPtr none{};
    launch_cov_mix(none, Ds.ld, 0, Hd.buf.p, p, ns, 0.0, add_noise ? sigma2 : 0.0, 1, co.p, st0);
  }
  for (int k0 = 0; k0 < ms; k0 += CH) {
    const int nl = std::min(CH, ms - k0);
    BatchPtr cl{};
    for (int j = 0; j < nl; ++j) {
      const int k = k0 + j;
      const lmm_gp_t& gp = P ? P->gps[l0 + k] : gps[l0 + k];
      if (P) cross_solve(P, k, gp, xsd.p, d, ns, R.p, ldr, nsr, st0);
      rider_stats_g(P ? R.p : nullptr, ldr, ns, P ? P->n : 0, P ? P->z[k].p : nullptr, gp.mean, 0.0, part.p,
                         ml.p + (size_t)k * ns, nullptr, st0);
      cov_at_xs(P, gp, xsd.p, d, ns, 0.0, nullptr, Ds, Cm[j].p, R.p, ldr, st0);
      cl.p[j] = Cm[j].p;
    }
    launch_cov_mix(cl, Ds.ld, nl, Hd.buf.p + (size_t)k0 * p, p, ns, jit->default_jitter, add_noise ? sigma2 : 0.0,
                   k0 == 0 ? 1 : 0, co.p, st0);
  }
  if (ms > 0) launch_mix(ml.p, ns, ms, Hd.buf.p, p, 1, 0.0, 0.0, nullptr, 0.0, mo.p, st0);
  mo.finish(st0); co.finish(st0);
  HIPCHK(hipStreamSynchronize(st0));
  return LMM_OK;
  LMM_CATCH
}

// cov(f::IndependentMOGP, x, y): reference src/independent_mogp.jl:66-71 (both inputs by outputs) and :184-215 (by features / mixed:
// the same blocks at permuted rows / columns).  Block l = cov(f_l, x.x, y.x): kernelmatrix(k_l, x, y) for a prior latent,
// K(x, y) - A_x' A_y with A_z = C.U' \ K(x_train, z) for a PosteriorGP latent (AbstractGPs).
extern "C" int lmm_mogp_cross_cov(const lmm_post_t* post, const lmm_gp_t* gps, int m, int latent_begin, int latent_end,
                                  const double* x, int d, int n, int x_by_features, const double* y, int n2, int y_by_features,
                                  double* cov_out) {
  std::lock_guard<std::mutex> lk(g_mu);
  REQUIRE_INIT();
  LMM_TRY
  if (post && post->f32 != g_f32) return fail(LMM_ERR_ARG, "posterior handle was built in the other compute dtype (lmm_set_compute_dtype)");
  if (!x || !y || !cov_out || d <= 0 || n <= 0 || n2 <= 0 || m <= 0) return fail(LMM_ERR_ARG, "bad arguments");
  if ((double)m * n * (double)m * n2 > 4e8) return fail(LMM_ERR_UNSUPPORTED, "cross-covariance (m n) x (m n2) too large");
  const lmm_post* P = post;
  int l0 = latent_begin, l1 = latent_end;
  if (P) {
    if (P->kind != 0) return fail(LMM_ERR_UNSUPPORTED, "cross-covariance of the coupled latents of a dense-H posterior is not built");
    l0 = P->l0; l1 = P->l1;
    if (P->m != m) return fail(LMM_ERR_DIM, "posterior has %d latents, m = %d", P->m, m);
    if (P->d != d) return fail(LMM_ERR_DIM, "input dimension mismatch: posterior has d=%d, x has d=%d", P->d, d);
  } else if (int rc = check_gps(gps, m)) return rc;
  if (l0 < 0 || l1 > m || l0 > l1) return fail(LMM_ERR_ARG, "bad latent shard");
  hipStream_t st0 = g.streams[0];
  DevIn xd(x, (size_t)d * n, st0), yd(y, (size_t)d * n2, st0);
  const size_t R = (size_t)m * n, Ccols = (size_t)m * n2;
  DevOut co(cov_out, R * Ccols);
  HIPCHK(hipMemsetAsync(co.p, 0, R * Ccols * sizeof(double), st0));        // the off-diagonal blocks (and latents outside the shard)
  // K(x, y) as the "rider rows" of a Gram launch whose column points are y: rows [NCy, NCy + nxr) of a (NCy + nxr) x NCy layout,
  // stored from buffer row 0 (the cross-Gram form of the predictive paths)
  const int nxr = rup(n, 128), NCy = rup(n2, 128);
  int ldk = nxr; if ((ldk % 512) == 0) ldk += 16;
  Buf<double> Kb(mat_count((size_t)ldk * NCy));
  int ldr = std::max(nxr, NCy); if ((ldr % 512) == 0) ldr += 16;
  Buf<double> Rx(P ? mat_count((size_t)ldr * P->NC) : 1), Ry(P ? mat_count((size_t)ldr * P->NC) : 1);
  for (int l = l0; l < l1; ++l) {
    const lmm_gp_t& gp = P ? P->gps[l] : gps[l];
    GramArgs a{};
    a.A = Kb.p; a.ld = ldk; a.nrows = NCy + nxr; a.ncols = NCy; a.row_tile0 = NCy / 64; a.row_shift = NCy; a.full = 1;
    a.x = yd.p; a.d = d; a.n = n2; a.kind = gp.kind; a.var = gp.variance; a.inv_ls = 1.0 / gp.lengthscale;
    a.xs = xd.p; a.ns = n;
    gram_g(a, st0, "cross-covariance K(x, y)");
    if (P) {
      const int k = l - l0;
      cross_solve(P, k, gp, xd.p, d, n, Rx.p, ldr, nxr, st0);              // R_x = K(x, X) L^-T
      cross_solve(P, k, gp, yd.p, d, n2, Ry.p, ldr, NCy, st0);             // R_y = K(y, X) L^-T
      gemm_nt_g(Kb.p, ldk, Rx.p, ldr, Ry.p, ldr, nxr, NCy, P->NC, 0, false, st0, "cross-covariance Schur complement");
    }
    launch_block_scatter(Kb.p, ldk, n, n2, co.p, R, x_by_features ? (size_t)l : (size_t)l * n, x_by_features ? m : 1,
                         y_by_features ? (size_t)l : (size_t)l * n2, y_by_features ? m : 1, st0);
  }
  co.finish(st0);
  HIPCHK(hipStreamSynchronize(st0));
  return LMM_OK;
  LMM_CATCH
}

int lmm_oilmm_post_logpdf(const lmm_post_t* post, const double* U, const double* S, int p, int m, double sigma2,
                          const double* xs, int d, int ns, const double* ys, int with_regulariser, double* out) {
  std::lock_guard<std::mutex> lk(g_mu);
  REQUIRE_INIT();
  LMM_TRY
  if (post && post->f32 != g_f32) return fail(LMM_ERR_ARG, "posterior handle was built in the other compute dtype (lmm_set_compute_dtype)");
  if (!post || !U || !S || !xs || !ys || !out || d <= 0 || ns <= 0) return fail(LMM_ERR_ARG, "bad arguments");
  const lmm_post* P = post;
  if (P->kind != 0) return fail(LMM_ERR_UNSUPPORTED, "dense-H posterior handle: use lmm_ilmm_post_logpdf");
  if (P->m != m) return fail(LMM_ERR_DIM, "posterior has %d latents, H has %d", P->m, m);
  if (P->d != d) return fail(LMM_ERR_DIM, "input dimension mismatch");
  if (m > p) return fail(LMM_ERR_DIM, "out dim of x != out dim of f.");
  if (!(sigma2 > 0.0)) return fail(LMM_ERR_ARG, "sigma2 must be > 0");
  hipStream_t st0 = g.streams[0];
  const int l0 = P->l0, l1 = P->l1, ms = l1 - l0;
  std::vector<double> T, ST, H;
  project_orthogonal(U, S, p, m, sigma2, T, ST, H);
  DevIn xsd(xs, (size_t)d * ns, st0), ysd(ys, (size_t)ns * p, st0);
  Uploaded Td(T, st0);
  const int C = with_regulariser ? m : ms, c0 = with_regulariser ? 0 : l0;
  Buf<double> Ty((size_t)ns * std::max(C, 1)), resid_dev(1);
  double resid = 0.0;
  if (C > 0) project_on_device(ysd.p, ns, p, Td.buf, m, c0, C, nullptr, Ty.p, st0);
  if (with_regulariser) {
    Uploaded Hd(H, st0);
    Buf<double> partial(tall_skinny_partials(ns, p));
    residual_on_device(ysd.p, ns, p, Ty.p, m, Hd.buf, partial.p, resid_dev.p, st0);
    HIPCHK(hipMemcpyAsync(&resid, resid_dev.p, sizeof(double), hipMemcpyDeviceToHost, st0));
    HIPCHK(hipStreamSynchronize(st0));
  }
  const double* Ty_shard = Ty.p + (size_t)(l0 - c0) * ns;
  Dims Ds(ns, 1);
  XsSlots X(P, ms, ns, Ds);
  const int nslots = X.nslots;
  Buf<double> outd(std::max(ms, 1));
  Buf<int> info(std::max(ms, 1));
  HIPCHK(hipMemsetAsync(info.p, 0, std::max(ms, 1) * sizeof(int), st0));
  fork_slots(nslots);
  int bi = 0;
  for (int k0 = 0; k0 < ms; k0 += X.nb_per, ++bi) {
    const int s = bi % nslots, nb = std::min(X.nb_per, ms - k0);
    hipStream_t st = g.streams[s];
    Batch Bt;
    X.cross_solve_batch(P, s, k0, nb, xsd.p, d, ns, st);
    GramArgs ga[LMM_MAX_BATCH];
    BatchPtr Bb{}, Rb{};
    for (int j = 0; j < nb; ++j) {
      const int k = k0 + j;
      const lmm_gp_t& gp = P->gps[l0 + k];
      launch_vec_lin(Ty_shard + (size_t)k * ns, X.mu[s][j].p, -1.0, ns, X.rid[s][j].p, st);
      ga[j] = cov_args(gp, xsd.p, d, ns, ST[l0 + k], X.rid[s][j].p, Ds, X.B[s][j].p);
      Bb.p[j] = X.B[s][j].p; Rb.p[j] = X.R[s][j].p;
      Bt.add(X.B[s][j].p, X.WB[s][j].p, info.p + k);
    }
    cov_at_xs_batch(P, ga, nb, Ds, Bb, Rb, X.ldr, st);
    potrf_batch(Bt, Ds.ld, Ds.NR, Ds.NC, ns, st);
    launch_lml_reduce(Bt.A, nb, Ds.ld, ns, Ds.NC, 1, outd.p + k0, st);
  }
  join_slots(nslots);
  std::vector<double> lml(std::max(ms, 1), 0.0);
  std::vector<int> hinfo(std::max(ms, 1), 0);
  HIPCHK(hipMemcpyAsync(lml.data(), outd.p, std::max(ms, 1) * sizeof(double), hipMemcpyDeviceToHost, st0));
  HIPCHK(hipMemcpyAsync(hinfo.data(), info.p, std::max(ms, 1) * sizeof(int), hipMemcpyDeviceToHost, st0));
  HIPCHK(hipStreamSynchronize(st0));
  if (int rc = check_info(hinfo, l0)) return rc;
  double total = 0.0;
  for (int k = 0; k < ms; ++k) total += lml[k];
  if (with_regulariser) {
    double logdetS = 0.0;
    for (int l = 0; l < m; ++l) logdetS += std::log(S[l]);
    total += -((double)ns * (logdetS + (double)(p - m) * std::log(2.0 * M_PI * sigma2)) + resid / sigma2) / 2.0;
  }
  *out = total;
  return LMM_OK;
  LMM_CATCH
}

int lmm_lmm_rand_multi(const lmm_post_t* post, const lmm_gp_t* gps, const double* U, const double* S, int p, int m,
                       int latent_begin, int latent_end, double sigma2, int add_noise, const double* xs, int d, int ns,
                       int nsamples, const double* z_lat, const double* eps, const lmm_jitters_t* jit, double* out) {
  std::lock_guard<std::mutex> lk(g_mu);
  REQUIRE_INIT();
  LMM_TRY
  if (post && post->f32 != g_f32) return fail(LMM_ERR_ARG, "posterior handle was built in the other compute dtype (lmm_set_compute_dtype)");
  if (!U || !xs || !z_lat || !out || d <= 0 || ns <= 0 || p <= 0 || m <= 0 || nsamples <= 0) return fail(LMM_ERR_ARG, "bad arguments");
  if (add_noise && !eps) return fail(LMM_ERR_ARG, "eps is NULL");
  if (!jit) jit = &kDefaultJit;
  const lmm_post* P = post;
  int l0 = latent_begin, l1 = latent_end;
  if (P) {
    if (P->kind != 0) return fail(LMM_ERR_UNSUPPORTED, "dense-H posterior handle: use lmm_ilmm_post_rand");
    l0 = P->l0; l1 = P->l1;
    if (P->m != m) return fail(LMM_ERR_DIM, "posterior has %d latents, H has %d", P->m, m);
    if (P->d != d) return fail(LMM_ERR_DIM, "input dimension mismatch");
  } else if (int rc = check_gps(gps, m)) return rc;
  if (l0 < 0 || l1 > m || l0 > l1) return fail(LMM_ERR_ARG, "bad latent shard");
  const int ms = l1 - l0;
  // OILMM: f(x) default jitter 1e-18 (reference src/oilmm.jl:47); dense-H ILMM: 1e-12 (src/ilmm.jl:84)
  const double jitter = S ? jit->default_jitter : jit->ilmm_rand_jitter;
  hipStream_t st0 = g.streams[0];
  std::vector<double> Hs((size_t)p * std::max(ms, 1), 0.0);
  for (int k = 0; k < ms; ++k)
    for (int o = 0; o < p; ++o) Hs[o + (size_t)k * p] = U[o + (size_t)(l0 + k) * p] * (S ? std::sqrt(S[l0 + k]) : 1.0);
  Uploaded Hd(Hs, st0);
  // z_lat: [sample][m][ns], eps: [sample][p][ns], out: [sample][p][ns]
  DevIn xsd(xs, (size_t)d * ns, st0), zd(z_lat, (size_t)ns * m * nsamples, st0);
  DevIn epsd(add_noise ? eps : nullptr, (size_t)ns * p * nsamples, st0);
  Dims Ds(ns, 0);
  XsSlots Xs(P, ms, ns, Ds);
  const int nslots = Xs.nslots;
  Buf<double> X((size_t)ns * std::max(ms, 1) * nsamples);     // [sample][latent of the shard][ns]
  Buf<int> info(std::max(ms, 1));
  HIPCHK(hipMemsetAsync(info.p, 0, std::max(ms, 1) * sizeof(int), st0));
  fork_slots(nslots);
  int bi = 0;
  for (int k0 = 0; k0 < ms; k0 += Xs.nb_per, ++bi) {
    const int s = bi % nslots, nb = std::min(Xs.nb_per, ms - k0);
    hipStream_t st = g.streams[s];
    Batch Bt;
    if (P) Xs.cross_solve_batch(P, s, k0, nb, xsd.p, d, ns, st);      // posterior: R_k and the mean vectors (sample = mean + L z)
    GramArgs ga[LMM_MAX_BATCH];
    BatchPtr Bb{}, Rb{};
    for (int j = 0; j < nb; ++j) {
      const int k = k0 + j;
      const lmm_gp_t& gp = P ? P->gps[l0 + k] : gps[l0 + k];
      ga[j] = cov_args(gp, xsd.p, d, ns, jitter, nullptr, Ds, Xs.B[s][j].p);
      Bb.p[j] = Xs.B[s][j].p; Rb.p[j] = Xs.R[s][j].p;
      Bt.add(Xs.B[s][j].p, Xs.WB[s][j].p, info.p + k);
    }
    cov_at_xs_batch(P, ga, nb, Ds, Bb, Rb, Xs.ldr, st);
    potrf_batch(Bt, Ds.ld, Ds.NR, Ds.NC, ns, st);      // ONE factorisation per latent, nsamples triangular products
    for (int j = 0; j < nb; ++j) {
      const int k = k0 + j;
      const double mu_const = P ? 0.0 : gps[l0 + k].mean;
      for (int q = 0; q < nsamples; ++q) {
        double* Xq = X.p + ((size_t)q * ms + k) * ns;
        launch_trmv_lower(Xs.B[s][j].p, Ds.ld, ns, zd.p + ((size_t)q * m + l0 + k) * ns, mu_const, Xs.part[s].p, Xq, st);
        if (P) launch_vec_lin(Xq, Xs.mu[s][j].p, 1.0, ns, Xq, st);
      }
    }
  }
  join_slots(nslots);
  DevOut od(out, (size_t)ns * p * nsamples);
  // reference src/oilmm.jl:50-53 / src/ilmm.jl:86: F = vec((H X')') + sqrt(sigma2) eps
  for (int q = 0; q < nsamples; ++q)
    launch_mix(X.p + (size_t)q * ms * ns, ns, ms, Hd.buf.p, p, 1, 0.0, 0.0, add_noise ? epsd.p + (size_t)q * ns * p : nullptr,
               std::sqrt(sigma2), od.p + (size_t)q * ns * p, st0);
  od.finish(st0);
  std::vector<int> hinfo(std::max(ms, 1), 0);
  HIPCHK(hipMemcpyAsync(hinfo.data(), info.p, std::max(ms, 1) * sizeof(int), hipMemcpyDeviceToHost, st0));
  HIPCHK(hipStreamSynchronize(st0));
  return check_info(hinfo, l0);
  LMM_CATCH
}

int lmm_lmm_rand(const lmm_post_t* post, const lmm_gp_t* gps, const double* U, const double* S, int p, int m,
                 int latent_begin, int latent_end, double sigma2, int add_noise, const double* xs, int d, int ns,
                 const double* z_lat, const double* eps, const lmm_jitters_t* jit, double* out) {
  return lmm_lmm_rand_multi(post, gps, U, S, p, m, latent_begin, latent_end, sigma2, add_noise, xs, d, ns, 1, z_lat, eps, jit, out);
}

// Standard normals on the device (Philox4x32-10 + Box-Muller, Float64): out[j], j < count, reproducible for (seed, stream).
// out may be a host or a device pointer.  Optional companion of lmm_lmm_rand / lmm_lmm_rand_multi, whose normals are
// caller-supplied: the Julia shim draws them from the reference's rng on the host; this generator serves callers that do not
// need that stream (SURVEY.md section 8a, K7 "optional Philox").
int lmm_normals(unsigned long long seed, unsigned long long stream, size_t count, double* out) {
  std::lock_guard<std::mutex> lk(g_mu);
  REQUIRE_INIT();
  LMM_TRY
  if (!out && count > 0) return fail(LMM_ERR_ARG, "out is NULL");
  hipStream_t st0 = g.streams[0];
  DevOut od(out, count);
  launch_normals(seed, stream, count, od.p, st0);
  od.finish(st0);
  HIPCHK(hipStreamSynchronize(st0));
  return LMM_OK;
  LMM_CATCH
}

// ------------------------------------------------------------------------------------------------
// building blocks (device pointers) for tests / profiling
// ------------------------------------------------------------------------------------------------
int lmm_dev_potrf(double* A, int nrows, int ncols, int ld, double* Winv, int n_real, int* info_dev) {
  std::lock_guard<std::mutex> lk(g_mu);
  REQUIRE_INIT();
  LMM_TRY
  if (!A || !Winv || !info_dev || nrows % 64 || ncols % 64 || nrows < ncols || ld < nrows || (ld & 1))
    return fail(LMM_ERR_ARG, "bad arguments");
  potrf_rec(A, ld, nrows, 0, ncols, Winv, n_real, info_dev, g.streams[0]);
  int hinfo = 0;
  HIPCHK(hipMemcpyAsync(&hinfo, info_dev, sizeof(int), hipMemcpyDeviceToHost, g.streams[0]));
  HIPCHK(hipStreamSynchronize(g.streams[0]));
  // a dependency-wait timeout is an error of the launch (LMM_ERR_HIP); a non-positive pivot stays in *info_dev for the caller, as before
  if (hinfo == LMM_INFO_SYNC_TIMEOUT) return check_info(&hinfo, 1, 0);
  return LMM_OK;
  LMM_CATCH
}

// The region kernel's row-task plan for one block column (host arithmetic only; no device, no lmm_init needed).
int lmm_dev_region_plan(int P, int nb, int rows_below, int rows_real, int cus, int assistants, int out[3]) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (!out || P < 1 || P > LMM_REGION_MAX_PANELS || nb < 1 || nb > LMM_MAX_BATCH || rows_below < 0 || rows_real > rows_below || cus < 1 || assistants < 0) {
    return fail(LMM_ERR_ARG, "lmm_dev_region_plan: bad arguments");
  }
  region_plan_probe(P, nb, rows_below, rows_real, cus, assistants, out);
  return LMM_OK;
}
// Test hook of the dataflow kernels' launch-epoch counter: *old_epoch (may be NULL) = the current value; set_to >= 0 replaces it (set
// it to 2^26 - 2 and the next launches execute the wrap-around clear of the persistent flag words).
int lmm_dev_flag_epoch(int set_to, int* old_epoch) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (set_to >= (1 << 26)) return fail(LMM_ERR_ARG, "lmm_dev_flag_epoch: the epoch has 26 bits");
  const int old = region_flag_epoch(set_to);
  if (old_epoch) *old_epoch = old;
  return LMM_OK;
}
// The allocation-extent guard on a freshly pooled block of alloc_bytes: LMM_OK when a rows x cols block of doubles with leading
// dimension ld fits, LMM_ERR_ARG (and nothing launched) when it does not.  Exists so that the guard itself has a test.
int lmm_dev_extent_check(size_t alloc_bytes, size_t rows, size_t ld, size_t cols) {
  std::lock_guard<std::mutex> lk(g_mu);
  REQUIRE_INIT();
  LMM_TRY
  Buf<double> blk((alloc_bytes + 7) / 8);
  guard_extent(blk.p, rows, ld, cols, false, "lmm_dev_extent_check");
  return LMM_OK;
  LMM_CATCH
}

// Host-only (no GPU, no lmm_init needed): the status the library derives from `count` pivot-info words -- LMM_OK, LMM_ERR_NOT_PD
// (first non-zero word; lmm_last_error_detail gives latent_begin + index and the pivot) or LMM_ERR_HIP when ANY word carries the
// region kernel's dependency-timeout marker (-7777), whichever position it is in.  Exists so that this translation has a test.
int lmm_dev_check_info(const int* info, int count, int latent_begin) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (!info || count < 0) return fail(LMM_ERR_ARG, "bad arguments");
  return check_info(info, (size_t)count, latent_begin);
}

int lmm_dev_gemm_nt_sub(double* C, int ldc, const double* A, int lda, const double* B, int ldb, int M, int N, int K,
                        int lower) {
  std::lock_guard<std::mutex> lk(g_mu);
  REQUIRE_INIT();
  LMM_TRY
  if (!C || !A || !B || M <= 0 || N <= 0 || K <= 0 || M % 64 || N % 64 || K % 16 || (ldc & 1) || (lda & 1) || (ldb & 1) || ldc < M || lda < M || ldb < N)
    return fail(LMM_ERR_ARG, "bad arguments");
  // lower: the tile enumeration (MT - tj row tiles under column tile tj of a common-origin region) assumes a trapezoid at least as
  // tall as it is wide
  if (lower && M < N) return fail(LMM_ERR_ARG, "lower != 0 needs M >= N (lower trapezoid of a common-origin region)");
  launch_gemm_nt(C, ldc, A, lda, B, ldb, M, N, K, lower, false, g.streams[0]);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(g.streams[0]));
  return LMM_OK;
  LMM_CATCH
}

int lmm_dev_gram(double* A, int ld, int nrows, int ncols, const double* x, int d, int n, const lmm_gp_t* gp,
                 double diag_add) {
  std::lock_guard<std::mutex> lk(g_mu);
  REQUIRE_INIT();
  LMM_TRY
  if (!A || !x || !gp || nrows % 64 || ncols % 64 || (ld & 1) || ld < nrows) return fail(LMM_ERR_ARG, "bad arguments");
  GramArgs a{};
  a.A = A; a.ld = ld; a.nrows = nrows; a.ncols = ncols; a.x = x; a.d = d; a.n = n;
  a.kind = gp->kind; a.var = gp->variance; a.inv_ls = 1.0 / gp->lengthscale; a.diag_add = diag_add; a.pad_diag = 1.0;
  gram_g(a, g.streams[0]);
  HIPCHK(hipStreamSynchronize(g.streams[0]));
  return LMM_OK;
  LMM_CATCH
}

int lmm_profile_begin(int serial) {
  std::lock_guard<std::mutex> lk(g_mu);
  REQUIRE_INIT();
  g.prof = true; g.prof_serial = serial != 0;
  g.prof_recs.clear();
  return LMM_OK;
}

int lmm_profile_end(lmm_prof_entry_t* out) {
  std::lock_guard<std::mutex> lk(g_mu);
  REQUIRE_INIT();
  LMM_TRY
  if (!out) return fail(LMM_ERR_ARG, "out is NULL");
  HIPCHK(hipDeviceSynchronize());
  for (int c = 0; c < LMM_PROF_COUNT; ++c) { out[c].launches = 0; out[c].ms = 0.0; out[c].work = 0.0; out[c].bytes = 0.0; }
  for (auto& r : g.prof_recs) {
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, r.e0, r.e1));
    out[r.cls].launches += r.count; out[r.cls].ms += ms; out[r.cls].work += r.work; out[r.cls].bytes += r.bytes;
    if (getenv("LMM_PROF_DUMP") && r.M > 0)
      fprintf(stderr, "[prof] cls=%d M=%d N=%d K=%d ms=%.4f tflops=%.2f\n", r.cls, r.M, r.N, r.K, ms, r.work / (ms * 1e-3) / 1e12);
    g.ev_pool.push_back(r.e0); g.ev_pool.push_back(r.e1);
  }
  g.prof_recs.clear();
  g.prof = false; g.prof_serial = false;
  return LMM_OK;
  LMM_CATCH
}

// Write-only yardstick for the Gram assembly's roofline line: GB/s of hipMemsetAsync into a pooled block of `bytes` (median-free mean
// of `reps` back-to-back fills between two events on the main stream, after one untimed fill that touches the block).
int lmm_dev_write_rate(size_t bytes, int reps, double* gbs) {
  std::lock_guard<std::mutex> lk(g_mu);
  REQUIRE_INIT();
  LMM_TRY
  if (!gbs || bytes < (1u << 20) || reps < 1) return fail(LMM_ERR_ARG, "bad arguments");
  Buf<double> blk(bytes / 8);
  hipStream_t st = g.streams[0];
  hipEvent_t e0, e1;
  HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
  HIPCHK(hipMemsetAsync(blk.p, 0, bytes, st));
  HIPCHK(hipEventRecord(e0, st));
  for (int r = 0; r < reps; ++r) HIPCHK(hipMemsetAsync(blk.p, 0, bytes, st));
  HIPCHK(hipEventRecord(e1, st));
  HIPCHK(hipStreamSynchronize(st));
  float ms = 0.f;
  HIPCHK(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  *gbs = (double)bytes * reps / (ms * 1e-3) / 1e9;
  return LMM_OK;
  LMM_CATCH
}

int lmm_dev_mfma_f64_peak(double* tflops) {
  std::lock_guard<std::mutex> lk(g_mu);
  REQUIRE_INIT();
  LMM_TRY
  const int blocks = 256 * 2, iters = 5000;
  Buf<double> out((size_t)blocks * 256);
  hipEvent_t e0, e1;
  HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
  launch_mfma_peak(out.p, blocks, iters, g.streams[0]);   // warm-up of the same length (clock ramp)
  HIPCHK(hipEventRecord(e0, g.streams[0]));
  launch_mfma_peak(out.p, blocks, iters, g.streams[0]);
  HIPCHK(hipEventRecord(e1, g.streams[0]));
  HIPCHK(hipEventSynchronize(e1));
  float ms = 0.f;
  HIPCHK(hipEventElapsedTime(&ms, e0, e1));
  const double flops = (double)blocks * 4 /*waves*/ * iters * 16.0 * 2048.0;      // 16 MFMAs of 16 x 16 x 4 x 2 flops per iteration
  *tflops = flops / (ms * 1e-3) / 1e12;
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return LMM_OK;
  LMM_CATCH
}

}  // extern "C"
