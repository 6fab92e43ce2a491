// Host-side state shared by the translation units of libsthenomi.so: the context (one GPU, its
// streams, scratch and a grow-only device-memory cache), device-resident specs, RAII buffers.
#pragma once
#include <functional>
#include "common.h"
#include "../../include/sthenomi.h"
#include "../../include/sthenomi_bench.h"

#include <mutex>
#include <vector>

// Grow-only cache of device allocations of one context.  The host-buffer entry points
// (sgp_logpdf, sgp_rand, sgp_posterior_predict, ...) used to hipMalloc / hipFree their N x N
// workspace on every call; now a released block goes back to the cache and the next call of the
// same shape gets it without touching the driver.  Blocks are only handed out again after the
// context's streams have been drained (sgp::CtxScope), so reuse is safe across calls.
struct sgp_pool_block {
  void* p = nullptr;
  size_t bytes = 0;
  bool used = false;
};

struct sgp_multi;   // multi.hip: the ranks of a multi-GPU context
void sgp_multi_destroy(sgp_multi* m);
struct sgp_mpost;   // multi.hip: a kept sharded factor (posterior on a multi-GPU context)
// the operators of a multi-GPU context (multi.hip), called from the C-ABI entry points with the primary context held
int sgp_multi_logpdf(struct sgp_ctx* ctx, const sgp_cov_spec* spec, const double* mean, int noise_kind,
                     const double* noise, const double* Y, int64_t ldy, int64_t ncols, double* out);
int sgp_multi_rand(struct sgp_ctx* ctx, const sgp_cov_spec* spec, const double* mean, int noise_kind,
                   const double* noise, const double* Z, int64_t ldz, int64_t S, double* out, int64_t ldo);
int sgp_multi_posterior_create(struct sgp_ctx* ctx, const sgp_cov_spec* spec, const double* mean, int noise_kind,
                               const double* noise, const double* y, double* alpha_out, sgp_mpost** out);
int sgp_multi_posterior_predict(sgp_mpost* mp, const sgp_cov_spec* cross, const sgp_cov_spec* prior_ss,
                                const double* mean_s, double* mean_out, double* var_out, double* cov_out,
                                int64_t ldcov);
void sgp_multi_posterior_destroy(sgp_mpost* mp);
// covariance entry points on a multi-GPU context (round 6): column chunks / point slices per rank, no communication
int sgp_multi_kernelmatrix(struct sgp_ctx* ctx, const sgp_cov_spec* spec, double* K, int64_t ldk);
int sgp_multi_kernelmatrix_diag(struct sgp_ctx* ctx, const sgp_cov_spec* spec, double* out);
// state behind the test / diagnosis hooks of include/sthenomi_bench.h (libsthenomi_bench.so: bench_hooks.hip)
int sgp_multi_set_fault(struct sgp_multi* m, int rank, long step, double stall_s);
int sgp_multi_is_broken(struct sgp_multi* m);
const std::vector<double>& sgp_multi_profile_pieces(struct sgp_multi* m);
// executed / dense tile products of the last sharded factorisation (sgp_ctx_factor_work on a multi-GPU context)
int sgp_multi_factor_work(struct sgp_multi* m, double* executed, double* dense);
int sgp_multi_logpdf_grad(struct sgp_ctx* ctx, const sgp_cov_spec* spec, const double* mean, int noise_kind,
                          const double* noise, const double* y, double* logpdf_out, double* grad_y, double* grad_mean,
                          double* grad_noise, double* grad_coef, double* grad_inscale, double* const* grad_inputs = nullptr,
                          double* const* grad_rowscale = nullptr);
// sgp_elbo_grad / _x / _xs: the arguments of the three entry points as one record (NULL = not asked for), and one rank's
// view of a data-sharded call (round 6)
namespace sgp {
struct ElboGradArgs {
  const sgp_cov_spec *zz = nullptr, *xz = nullptr;
  const double *var_x = nullptr, *mean_x = nullptr;
  int noise_kind = 0;
  const double* noise_x = nullptr;
  int z_noise_kind = 0;
  const double *z_noise = nullptr, *y = nullptr;
  double *elbo_out = nullptr, *grad_y = nullptr, *grad_mean = nullptr, *grad_noise = nullptr, *grad_var_x = nullptr,
         *grad_z_noise = nullptr, *grad_coef_zz = nullptr, *grad_inscale_zz = nullptr, *grad_coef_xz = nullptr,
         *grad_inscale_xz = nullptr;
  double* const* grad_inputs_zz = nullptr;
  double* const* grad_inputs_xz = nullptr;
  double* const* grad_rowscale_zz = nullptr;
  double* const* grad_rowscale_xz = nullptr;
  double* const* grad_colscale_xz = nullptr;
};
struct ElboGradShard {
  bool primary = false;   // the rank that also produces the zz-side results (G_zz and its contractions)
  long n_total = 0;       // data points of the whole call (the bound's N log 2 pi)
  // collective over the ranks of the call: d_part (device memory of this rank, len doubles, complete) <- the sum over ranks
  std::function<int(double* d_part, long len)> reduce;
};
}  // namespace sgp
// data points sharded over the ranks: one reduction of M^2 + M + 4 doubles between the two factorisations (multi.hip)
int sgp_multi_elbo_grad(struct sgp_ctx* ctx, const sgp::ElboGradArgs& a);
// h6: the six terms of the bound (capi.hip: vfe_pipeline).  dLz / d_wz / d_part0 / d_wg non-NULL (device 0
// buffers): the M x M factors a sparse posterior keeps (d_part0 doubles as rank 0's part and ends up holding chol(A A' + I))
int sgp_multi_vfe(struct sgp_ctx* ctx, const sgp_cov_spec* zz, const sgp_cov_spec* xz, const double* var_x,
                  const double* mean_x, int noise_kind, const double* noise_x, int z_noise_kind,
                  const double* z_noise, const double* y, double* h6, double* dLz, double* d_wz, double* d_part0,
                  double* d_wg);

struct sgp_ctx {
  int device = 0;
  long serial = 0;                // unique per created context (capi.hip: ctx_is_live)
  sgp_multi* multi = nullptr;     // non-null: the operators shard over several GPUs (sgp_ctx_create_multi)
  int multi_nranks = 0;
  hipStream_t stream = nullptr;   // panel / critical-path stream (high priority)
  hipStream_t stream2 = nullptr;  // trailing-update stream (look-ahead overlap)
  hipEvent_t ev_panel = nullptr, ev_rest = nullptr;
  int lookahead = 1;
  long la_max_n = 65536;   // SGP_LA_MAX_N: look-ahead only for factorisations of fewer columns (serial + fused from there on)
  long wout = 0;  // 0 = automatic
  long wmid = 0;  // middle blocking level of an outer panel (capi.hip: panel_factor_mid); 0 = automatic
  double* d_invd = nullptr;    // 8 x 256: micro-block inverses of the current diagonal block
  int fuse_potrf = 11;         // SGP_FUSE_POTRF: bit 0 = inner K = 128 updates, bit 1 = outer trailing updates also factor the next
                               // diagonal block (tile (0, 0) of their C) in the workgroup that updates it, bit 3 = that tile goes
                               // from the accumulators straight into the factorisation's LDS layout; bit 2 = at every size
  long fuse_max_n = 32768;     // SGP_FUSE_MAX_N: fused launches only while n_pad is below this (bit 2 lifts the limit)
  int fuse_now = 0;            // what the factorisation under way uses (set by chol_bordered / sgp_dev_panel_factor)
  double* d_slots = nullptr;   // per-128-block logdet contributions
  long n_slots = 0;
  double* d_scal = nullptr;    // [0] logdet, [1] misc, [16 ..] per-rhs sums
  long n_scal = 0;
  int* d_info = nullptr;
  // dataflow (single-launch) factorisation, chol_df.hip: SGP_DATAFLOW = 0 never, 1 whenever it applies, unset: by size
  int dataflow = -1;
  long df_min_n = 3072, df_max_n = 65536;   // automatic mode: n_pad range it is used for (SGP_DF_MIN_N / SGP_DF_MAX_N)
  int df_wgs = 0;                   // persistent workgroups (2 per CU)
  double df_timeout_s = 10.0;       // bound of a single dependency wait inside the kernel
  int* d_df_state = nullptr;        // task counter, abort word, per-tile-row progress (grown on demand)
  long n_df_state = 0;
  long df_fat_max_n = 24576;        // SGP_DF_FAT_MAX_N: below this the one-workgroup-per-CU (256 VGPR) instantiation
  long long* d_df_stats = nullptr;  // SGP_DF_STATS=1: per-workgroup tick counters, summarised on stderr after every launch
  double* d_df_inv = nullptr;       // inverse diagonal blocks of every 128-block when the caller keeps none
  long n_df_inv = 0;
  // hybrid schedule (round 5; capi.hip: use_hybrid): launches with look-ahead whose PANEL factorisation is one dataflow
  // launch on the panel (tile-level dependencies for the chain and its row solves), lock-step launches for the trailing updates
  // SGP_HYBRID = 0 never, 1 whenever it applies (n_pad >= 4096), unset (-1): from SGP_HYBRID_MIN_N columns on (measured:
  // profiles/r05_experiments/hybrid.md); SGP_HYBRID_W / _WGS / _FAT: panel width, persistent workgroups of a panel launch, the
  // one-workgroup-per-CU instantiation
  int hybrid = -1, hybrid_wgs = 256, hybrid_fat = 1;
  long hybrid_w = 2048, hybrid_min_n = 24576;
  long hybrid_grow_min_n = 16384;   // the gradient path's factorisations (identity border): SGP_HYBRID_GROW_MIN_N
  // sgp_logpdf_batch (round 6): equally sized members up to SGP_BATCH_MAX_N padded columns are factored as ONE task pool of the
  // dataflow kernel (0: never); SGP_BATCH_FAT: its one-workgroup-per-CU instantiation (1) or the lean one (0)
  long batch_max_n = 12288;
  int batch_fat = 1;
  int hybrid_serial = 0;       // SGP_HYBRID_SERIAL = 1: one stream (bench.py: the update launches' rate with the chip to themselves)
  bool df_timed_out = false;   // the last dataflow launch ran into its wait bound (fetch_info)
  int df_fallback = 1;         // SGP_DF_FALLBACK=0: report the timeout instead (the kernel's own error path, tests)
  // structural zeros (common.h; capi.hip: sz_build): SGP_STRUCT_ZEROS = 0 switches the skipping off (A/B)
  int struct_zeros = 1;
  sgp::sz_word* d_sz = nullptr;   // device copy of the factor's tile pattern (grow-only)
  size_t n_sz = 0;
  std::vector<sgp::sz_word> h_sz;
  sgp::sz_word* h_sz_pin = nullptr;   // pinned staging copy of the pattern (capi.hip: sz_upload), reused behind ev_sz
  size_t n_sz_pin = 0;
  hipEvent_t ev_sz = nullptr;
  int* d_szmap = nullptr;         // scratch of the compacted live-tile id maps (gemm_nt.hip: tile_compact_kernel), grow-only
  long n_szmap = 0;
  double sz_executed = 0, sz_dense = 0;   // k-block products of the last factorisation: run / of the dense schedule
  const double* sz_base = nullptr;        // the matrix the pattern in h_sz belongs to while chol_bordered runs (else nullptr)
  long sz_ld = 0;
  int sz_words = 0;
  // fraction of the lower tiles of the update C -= P P' that are live under the pattern (1 without one): the timed
  // (instrumented) step prices a launch by the tile products it runs
  double sz_live_fraction(const double* P, long ld, const double* C, long M, long Nc, long K) const {
    if (!sz_base || ld != sz_ld) return 1.0;
    const long oc = C - sz_base, op = P - sz_base;
    if (oc < 0 || op < 0) return 1.0;
    const long cr = oc % ld, cc = oc / ld, pc = op / ld;
    if (cr != cc || cr % 128 || pc % 128 || K % 128) return 1.0;
    const long t0 = cr / 128, kt0 = pc / 128, kt1 = kt0 + K / 128, n_tr = M / 128, n_tc = Nc / 128;
    double live = 0, all = 0;
    for (long tc = 0; tc < n_tc; ++tc)
      for (long tr = tc; tr < n_tr; ++tr) {
        const sgp::sz_word* ra = &h_sz[(size_t)(t0 + tr) * sz_words];
        const sgp::sz_word* rb = &h_sz[(size_t)(t0 + tc) * sz_words];
        long kmin = -1, kmax = -1;
        for (long k = kt0; k < kt1; ++k)
          if (((ra[k >> 6] & rb[k >> 6]) >> (k & 63)) & 1) {
            if (kmin < 0) kmin = k;
            kmax = k;
          }
        all += (double)(kt1 - kt0);
        if (kmin >= 0) live += (double)(kmax - kmin + 1);   // the tile contracts first .. last needed k tile
      }
    return all > 0 ? live / all : 1.0;
  }
  long df_fallbacks = 0;       // operators rerun on the launch-based schedule because of that (capi.hip: with_df_fallback)
  std::recursive_mutex mu;   // recursive: with_df_fallback (capi.hip) holds it across an operator and its rerun
  // optional per-launch timing of the trailing updates (roofline evidence for bench.py)
  bool time_updates = false;
  std::vector<hipEvent_t> ev;
  std::vector<double> ev_flops;
  // optional per-stage HIP-event timing of the pipelines that bench.py cannot time from outside
  // (sgp_ctx_stage_timing / sgp_ctx_stage_ms; stage ids are documented in sthenomi.h)
  int stage_timing = 0;
  double stage_ms[16] = {0};
  // device-memory cache (see sgp_pool_block); guarded by mu
  std::vector<sgp_pool_block> pool;
  size_t pool_bytes = 0;
  int pool_enabled = 1;
  // pinned staging area for small host -> device uploads (spec inputs, y, mean): one async copy
  // per call instead of one blocking hipMemcpy per array
  char* h_stage = nullptr;
  char* d_stage = nullptr;
  size_t stage_cap = 0;
};

namespace sgp {

extern thread_local sgp_ctx* tl_ctx;  // the context whose entry point this thread is inside

void* pool_alloc(sgp_ctx* ctx, size_t bytes);   // nullptr on failure (error text set)
void pool_free(sgp_ctx* ctx, void* p);
void pool_trim(sgp_ctx* ctx);                   // release every unused block to the driver

// RAII: lock the context, make its device current, route DevBuf through its cache; on exit drain
// the context's streams so that cached blocks can be reused by the next call.
struct CtxScope {
  sgp_ctx* ctx;
  sgp_ctx* prev;
  std::unique_lock<std::recursive_mutex> lk;
  explicit CtxScope(sgp_ctx* c) : ctx(c), prev(tl_ctx), lk(c->mu) {
    hipSetDevice(c->device);
    tl_ctx = c;
  }
  ~CtxScope() {
    hipStreamSynchronize(ctx->stream);
    hipStreamSynchronize(ctx->stream2);
    tl_ctx = prev;
  }
};

// Attributes the time between successive mark() calls on one stream to the stage id of the earlier
// mark (HIP events); finish() drains the stream and adds the intervals to ctx->stage_ms.  A no-op
// unless the context has stage timing switched on.
struct StageTimer {
  sgp_ctx* ctx;
  hipStream_t s;
  std::vector<hipEvent_t> ev;
  std::vector<int> id;
  StageTimer(sgp_ctx* c, hipStream_t st) : ctx(c), s(st) {}
  ~StageTimer() {
    for (auto e : ev) hipEventDestroy(e);
  }
  void mark(int stage) {
    if (!ctx->stage_timing) return;
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return;
    ev.push_back(e);
    id.push_back(stage);
    hipEventRecord(e, s);
  }
  void finish() {
    if (!ctx->stage_timing || ev.empty()) return;
    mark(-1);
    hipEventSynchronize(ev.back());
    for (size_t i = 0; i + 1 < ev.size(); ++i) {
      float ms = 0;
      if (id[i] >= 0 && id[i] < 16 && hipEventElapsedTime(&ms, ev[i], ev[i + 1]) == hipSuccess) ctx->stage_ms[id[i]] += ms;
    }
  }
};

struct DevBuf {
  double* p = nullptr;
  sgp_ctx* owner = nullptr;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  DevBuf(DevBuf&& o) noexcept : p(o.p), owner(o.owner) { o.p = nullptr; }
  ~DevBuf() { release(); }
  void release() {
    if (!p) return;
    if (owner)
      pool_free(owner, p);
    else
      hipFree(p);
    p = nullptr;
  }
  int alloc(size_t n) {
    release();
    size_t bytes = sizeof(double) * (n ? n : 1);
    owner = tl_ctx;
    if (owner) {
      p = (double*)pool_alloc(owner, bytes);
      return p ? 0 : -2;
    }
    if (hipMalloc(&p, bytes) != hipSuccess) {
      set_error("hipMalloc failed (" + std::to_string(bytes) + " bytes)");
      p = nullptr;
      return -2;
    }
    return 0;
  }
  int upload(const double* h, size_t n) {
    int rc = alloc(n);
    if (rc) return rc;
    if (n && hipMemcpy(p, h, sizeof(double) * n, hipMemcpyHostToDevice) != hipSuccess) {
      set_error("hipMemcpy H2D failed");
      return -2;
    }
    return 0;
  }
};

}  // namespace sgp

struct sgp_dspec {
  sgp_ctx* ctx = nullptr;
  int nrb = 0, ncb = 0, symmetric = 0;
  std::vector<long> row_len, col_len, row_off, col_off;
  long N = 0, M = 0;
  std::vector<double*> d_bufs;          // everything to free (cache blocks of ctx)
  std::vector<int> term_ptr;            // CSR over pairs
  std::vector<sgp::DevTerm> h_terms;    // host copy (device pointers inside)
  sgp::DevTerm* d_terms = nullptr;
  std::vector<int> pair_dmax;
  std::vector<int> term_row_input;     // spec input index each term reads its row / column points from
  std::vector<int> term_col_input;
  std::vector<int> in_dim;             // per spec input
  std::vector<long> in_n;
};
