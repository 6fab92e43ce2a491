// libsthenomi.so host driver + C ABI (include/sthenomi.h).  gfx950 only; no CPU fallback:
// every numerical result is produced by the HIP kernels in this directory.
#include "ctx.h"
#include "driver.h"
#include "tilemap.h"
#include "sz_pattern.h"
#include "df_tasks.h"
#include <limits>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <cstdlib>
#include <mutex>
#include <set>
#include <vector>

namespace sgp {

static thread_local std::string g_err;
void set_error(const std::string& s) { g_err = s; }

static inline long rup(long x, long m) { return (x + m - 1) / m * m; }
static inline long spec_rows_host(const sgp_cov_spec* sp) {
  long n = 0;
  for (int i = 0; i < sp->n_row_blocks; ++i) n += sp->row_len[i];
  return n;
}
constexpr long NOMASK = -(1L << 40);
constexpr long INVD_STRIDE = 8 * 256;  // eight 16x16 inverse diagonal blocks per 128-block
constexpr long WOUT_SMALL = 512;   // outer panel width of the two-level blocked Cholesky
constexpr long WOUT_LARGE = 1024;  // ... for n_pad >= 32768 (halves the C-tile traffic per flop)

}  // namespace sgp

using namespace sgp;

#define CHECK_ARG(cond, msg)       \
  do {                             \
    if (!(cond)) {                 \
      sgp::set_error(msg);         \
      return -1;                   \
    }                              \
  } while (0)
#define CHECK_RC(expr)        \
  do {                        \
    int _rc = (expr);         \
    if (_rc != 0) return _rc; \
  } while (0)

extern "C" int sgp_abi_version(void) { return SGP_ABI_VERSION; }
extern "C" const char* sgp_last_error(void) { return g_err.c_str(); }

extern "C" int sgp_geometry(int64_t N, int64_t S, int64_t* n_pad, int64_t* m_tot) {
  long np = rup(N, TILE);
  if (np == 0) np = TILE;
  long mt = np + (S > 0 ? rup(S, TILE) : 0);
  if (n_pad) *n_pad = np;
  if (m_tot) *m_tot = mt;
  return 0;
}

// ---------------------------------------------------------------------------------------
// device-memory cache of a context
// ---------------------------------------------------------------------------------------
namespace sgp {
thread_local sgp_ctx* tl_ctx = nullptr;

void* pool_alloc(sgp_ctx* ctx, size_t bytes) {
  bytes = (bytes + 255) / 256 * 256;
  if (ctx->pool_enabled) {
    // best fit among the unused blocks that are not wastefully large for the request
    const size_t cap = bytes < (1u << 20) ? (4u << 20) : bytes + bytes / 4;
    long best = -1;
    for (size_t i = 0; i < ctx->pool.size(); ++i) {
      const sgp_pool_block& b = ctx->pool[i];
      if (b.used || b.bytes < bytes || b.bytes > cap) continue;
      if (best < 0 || b.bytes < ctx->pool[best].bytes) best = (long)i;
    }
    if (best >= 0) {
      ctx->pool[best].used = true;
      return ctx->pool[best].p;
    }
  }
  void* p = nullptr;
  hipError_t e = hipMalloc(&p, bytes);
  if (e != hipSuccess && ctx->pool_enabled) {  // give cached blocks back to the driver and retry once
    (void)hipGetLastError();
    pool_trim(ctx);
    e = hipMalloc(&p, bytes);
  }
  if (e != hipSuccess) {
    (void)hipGetLastError();
    set_error("hipMalloc failed (" + std::to_string(bytes) + " bytes)");
    return nullptr;
  }
  if (ctx->pool_enabled) {
    sgp_pool_block b;
    b.p = p;
    b.bytes = bytes;
    b.used = true;
    ctx->pool.push_back(b);
    ctx->pool_bytes += bytes;
  }
  return p;
}

void pool_free(sgp_ctx* ctx, void* p) {
  if (!p) return;
  for (auto& b : ctx->pool)
    if (b.p == p) {
      b.used = false;
      return;
    }
  hipFree(p);  // not a cached block (cache disabled)
}

void pool_trim(sgp_ctx* ctx) {
  hipStreamSynchronize(ctx->stream);
  hipStreamSynchronize(ctx->stream2);
  std::vector<sgp_pool_block> keep;
  for (auto& b : ctx->pool) {
    if (b.used) {
      keep.push_back(b);
    } else {
      hipFree(b.p);
      ctx->pool_bytes -= b.bytes;
    }
  }
  ctx->pool.swap(keep);
}
}  // namespace sgp

extern "C" int sgp_ctx_trim(sgp_ctx* ctx) {
  CHECK_ARG(ctx != nullptr, "sgp_ctx_trim: NULL argument");
  CtxScope scope(ctx);
  pool_trim(ctx);
  return 0;
}

extern "C" int sgp_ctx_stage_timing(sgp_ctx* ctx, int enable) {
  CHECK_ARG(ctx != nullptr, "sgp_ctx_stage_timing: NULL argument");
  CtxScope scope(ctx);
  ctx->stage_timing = enable ? 1 : 0;
  for (double& v : ctx->stage_ms) v = 0.0;
  return 0;
}

extern "C" int sgp_ctx_stage_ms(sgp_ctx* ctx, double* out16) {
  CHECK_ARG(ctx && out16, "sgp_ctx_stage_ms: NULL argument");
  CtxScope scope(ctx);
  for (int i = 0; i < 16; ++i) out16[i] = ctx->stage_ms[i];
  return 0;
}

// Live contexts.  A posterior handle (sgp_post / sgp_sparse_post) may outlive the context it was created on -- it is freed
// without it -- but predicting against it needs the context's streams, scratch and (multi-GPU) ranks: the predict entry
// points look the context up here and fail cleanly when it is gone (advisor, round 3: that was a use-after-free).
// (a destroyed context's address may be handed out again by the allocator: handles remember the context's SERIAL, too)
static std::mutex g_live_mu;
static std::set<const sgp_ctx*> g_live_ctx;
static long g_ctx_serial = 0;
static bool ctx_is_live(const sgp_ctx* c, long serial) {
  std::lock_guard<std::mutex> lk(g_live_mu);
  return g_live_ctx.count(c) != 0 && c->serial == serial;
}

// ---------------------------------------------------------------------------------------
// Run-time switches of a context: ONE table, read once when the context is created (docs/03_kernels.md section 3.5 lists
// every entry with the test that exercises it).  Round 6 pruned the switches whose A/B was closed -- the XCD task queues of
// the dataflow kernel, the explicit-inverse panel solves (SGP_REFINE), the stamp-kernel experiments, the fp32 driver's
// panel widths, the VFE two-chunk overlap, SGP_POOL, SGP_FUSE_MAX_N, SGP_LA_MAX_N, SGP_SPLITK_SUB, SGP_HYBRID_GROW.
// What is left selects between schedules that all give the same bits (the bit-identity tests flip them) or bounds a wait.
// ---------------------------------------------------------------------------------------
namespace {
struct CtxKnob {
  const char* name;
  void (*set)(sgp_ctx*, const char*);
};
const CtxKnob kCtxKnobs[] = {
    // ---- schedule of the blocked Cholesky (capi.hip: chol_bordered; every choice gives the same factor)
    {"SGP_LOOKAHEAD", [](sgp_ctx* c, const char* v) { c->lookahead = atoi(v); }},          // 0: no second stream, 2: at every size
    {"SGP_WOUT", [](sgp_ctx* c, const char* v) { c->wout = atol(v) / TILE * TILE; }},       // outer panel width of the launches
    {"SGP_WMID", [](sgp_ctx* c, const char* v) { c->wmid = atol(v) / TILE * TILE; }},       // middle blocking level
    {"SGP_FUSE_POTRF", [](sgp_ctx* c, const char* v) { c->fuse_potrf = atoi(v); }},        // fused update + diagonal block
    {"SGP_DATAFLOW", [](sgp_ctx* c, const char* v) { c->dataflow = atoi(v); }},            // 0 never / 1 always / unset by size
    {"SGP_DF_MIN_N", [](sgp_ctx* c, const char* v) { c->df_min_n = atol(v); }},
    {"SGP_DF_MAX_N", [](sgp_ctx* c, const char* v) { c->df_max_n = atol(v); }},
    {"SGP_DF_FAT_MAX_N", [](sgp_ctx* c, const char* v) { c->df_fat_max_n = atol(v); }},    // one workgroup per CU below this
    {"SGP_DF_WGS", [](sgp_ctx* c, const char* v) { if (atoi(v) > 0) c->df_wgs = atoi(v); }},
    {"SGP_DF_TIMEOUT_S", [](sgp_ctx* c, const char* v) { c->df_timeout_s = atof(v); }},    // bound of one wait inside the kernel
    {"SGP_DF_FALLBACK", [](sgp_ctx* c, const char* v) { c->df_fallback = atoi(v); }},      // 0: report the time-out (tests)
    {"SGP_HYBRID", [](sgp_ctx* c, const char* v) { c->hybrid = atoi(v); }},                // 0 never / 1 from 4096 columns on
    {"SGP_HYBRID_W", [](sgp_ctx* c, const char* v) { c->hybrid_w = std::max<long>(TILE, atol(v) / TILE * TILE); }},
    {"SGP_HYBRID_WGS", [](sgp_ctx* c, const char* v) { c->hybrid_wgs = std::max(8, atoi(v)); }},
    {"SGP_HYBRID_FAT", [](sgp_ctx* c, const char* v) { c->hybrid_fat = atoi(v); }},
    // (the gradient path's crossover -- 16384 by default -- follows a limit the user sets: advisor, round 5: it used to stay at
    // min(limit, 16384), so raising the limit could not move it; SGP_HYBRID_GROW_MIN_N, read after it, names it separately)
    {"SGP_HYBRID_MIN_N", [](sgp_ctx* c, const char* v) { c->hybrid_min_n = c->hybrid_grow_min_n = atol(v); }},
    {"SGP_HYBRID_GROW_MIN_N", [](sgp_ctx* c, const char* v) { c->hybrid_grow_min_n = atol(v); }},
    {"SGP_HYBRID_SERIAL", [](sgp_ctx* c, const char* v) { c->hybrid_serial = atoi(v); }},  // one stream (bench.py: uncontended leg)
    {"SGP_BATCH_MAX_N", [](sgp_ctx* c, const char* v) { c->batch_max_n = atol(v); }},      // sgp_logpdf_batch: pooled up to this size
    // ---- structural zeros
    {"SGP_STRUCT_ZEROS", [](sgp_ctx* c, const char* v) { c->struct_zeros = atoi(v); }},    // 0: the dense schedule (A/B, same bits)
};
void apply_env_knobs(sgp_ctx* c) {
  for (const CtxKnob& k : kCtxKnobs)
    if (const char* v = getenv(k.name)) k.set(c, v);
}
}  // namespace

extern "C" int sgp_ctx_create(int device, sgp_ctx** out) {
  CHECK_ARG(out != nullptr, "sgp_ctx_create: out is NULL");
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0) {
    set_error("sgp_ctx_create: no HIP device visible (libsthenomi has no CPU path)");
    return -3;
  }
  CHECK_ARG(device >= 0 && device < ndev, "sgp_ctx_create: bad device ordinal");
  SGP_HIP(hipSetDevice(device));
  hipDeviceProp_t prop;
  SGP_HIP(hipGetDeviceProperties(&prop, device));
  if (std::string(prop.gcnArchName).find("gfx950") == std::string::npos) {
    set_error(std::string("sgp_ctx_create: device is ") + prop.gcnArchName +
              ", libsthenomi is built for gfx950 only");
    return -3;
  }
  sgp_ctx* c = new sgp_ctx();
  c->device = device;
  auto init = [&]() -> int {
    int lo = 0, hi = 0;
    SGP_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
    // the panel (critical-path) stream at high priority, the trailing-update stream below it
    SGP_HIP(hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, hi));
    SGP_HIP(hipStreamCreateWithPriority(&c->stream2, hipStreamNonBlocking, lo));
    SGP_HIP(hipEventCreateWithFlags(&c->ev_panel, hipEventDisableTiming));
    SGP_HIP(hipEventCreateWithFlags(&c->ev_rest, hipEventDisableTiming));
    {
      hipDeviceProp_t prop;
      SGP_HIP(hipGetDeviceProperties(&prop, device));
      c->df_wgs = 2 * prop.multiProcessorCount;
    }
    apply_env_knobs(c);
    SGP_HIP(hipMalloc(&c->d_invd, sizeof(double) * 8 * 256));
    c->n_slots = 1 << 15;
    SGP_HIP(hipMalloc(&c->d_slots, sizeof(double) * c->n_slots));
    c->n_scal = 16 + (1 << 16);
    SGP_HIP(hipMalloc(&c->d_scal, sizeof(double) * c->n_scal));
    SGP_HIP(hipMalloc(&c->d_info, sizeof(int)));
    return 0;
  };
  int rc = init();
  if (rc) {
    sgp_ctx_destroy(c);
    return rc;
  }
  {
    std::lock_guard<std::mutex> lk(g_live_mu);
    c->serial = ++g_ctx_serial;
    g_live_ctx.insert(c);
  }
  *out = c;
  return 0;
}

extern "C" int sgp_ctx_destroy(sgp_ctx* c) {
  if (!c) return 0;
  {
    std::lock_guard<std::mutex> lk(g_live_mu);
    g_live_ctx.erase(c);
  }
  if (c->multi) sgp_multi_destroy(c->multi);
  c->multi = nullptr;
  hipSetDevice(c->device);
  if (c->stream) hipStreamSynchronize(c->stream);
  if (c->stream2) hipStreamSynchronize(c->stream2);
  for (auto e : c->ev) hipEventDestroy(e);
  for (auto& b : c->pool) hipFree(b.p);
  if (c->h_stage) hipHostFree(c->h_stage);
  if (c->d_invd) hipFree(c->d_invd);
  if (c->d_slots) hipFree(c->d_slots);
  if (c->d_scal) hipFree(c->d_scal);
  if (c->d_info) hipFree(c->d_info);
  if (c->d_df_state) hipFree(c->d_df_state);
  if (c->d_sz) hipFree(c->d_sz);
  if (c->d_szmap) hipFree(c->d_szmap);
  if (c->h_sz_pin) hipHostFree(c->h_sz_pin);
  if (c->ev_sz) hipEventDestroy(c->ev_sz);
  if (c->d_df_inv) hipFree(c->d_df_inv);
  if (c->d_df_stats) hipFree(c->d_df_stats);
  if (c->ev_panel) hipEventDestroy(c->ev_panel);
  if (c->ev_rest) hipEventDestroy(c->ev_rest);
  if (c->stream2) hipStreamDestroy(c->stream2);
  if (c->stream) hipStreamDestroy(c->stream);
  delete c;
  return 0;
}

// ---------------------------------------------------------------------------------------
// device-resident specs
// ---------------------------------------------------------------------------------------
static int pow2ceil(int d) {
  int p = 1;
  while (p < d) p <<= 1;
  return p;
}

// Uploads a spec with ONE host -> device copy: inputs (packed to ld == dim), row / column scale
// vectors and the term table are laid out in a pinned staging buffer of the context and moved in
// one piece into one cached device block.  Caller holds the context (CtxScope / ctx->mu).
static void dspec_free(sgp_dspec* ds);
static int dspec_create(sgp_ctx* ctx, const sgp_cov_spec* sp, sgp_dspec** out) {
  CHECK_ARG(ctx && sp && out, "sgp_dspec_create: NULL argument");
  CHECK_ARG(sp->n_row_blocks >= 1 && sp->n_col_blocks >= 1, "spec: need >= 1 block");
  sgp_dspec* ds = new sgp_dspec();
  ds->ctx = ctx;
  ds->nrb = sp->n_row_blocks;
  ds->ncb = sp->n_col_blocks;
  ds->symmetric = sp->symmetric;
  long off = 0;
  for (int i = 0; i < ds->nrb; ++i) {
    ds->row_len.push_back(sp->row_len[i]);
    ds->row_off.push_back(off);
    off += sp->row_len[i];
  }
  ds->N = off;
  off = 0;
  for (int j = 0; j < ds->ncb; ++j) {
    ds->col_len.push_back(sp->col_len[j]);
    ds->col_off.push_back(off);
    off += sp->col_len[j];
  }
  ds->M = off;
  auto fail = [&](const char* msg) {
    set_error(msg);
    dspec_free(ds);
    return -1;
  };
  if (ds->symmetric && (ds->nrb != ds->ncb || ds->row_len != ds->col_len))
    return fail("spec: symmetric spec needs identical row / col blocks");
  // ---- pass 1: validate, lay everything out (byte offsets into one block, 256-byte aligned)
  size_t total = 0;
  auto place = [&](size_t bytes) {
    size_t at = total;
    total += (std::max<size_t>(bytes, 8) + 255) / 256 * 256;
    return at;
  };
  std::vector<size_t> in_off(sp->n_inputs);
  for (int k = 0; k < sp->n_inputs; ++k) {
    const sgp_input& in = sp->inputs[k];
    if (in.dim < 1 || in.dim > (1 << 20)) return fail("spec: input dimension must be >= 1");
    if (in.n < 0 || in.ld < in.dim) return fail("spec: bad input n / ld");
    in_off[k] = place(sizeof(double) * (size_t)(in.dim * in.n));
    ds->in_dim.push_back((int)in.dim);
    ds->in_n.push_back((long)in.n);
  }
  int npairs = ds->nrb * ds->ncb;
  ds->term_ptr.assign(sp->term_ptr, sp->term_ptr + npairs + 1);
  ds->pair_dmax.assign(npairs, 1);
  const int nterms = sp->term_ptr[npairs];
  std::vector<size_t> rs_off(nterms, (size_t)-1), cs_off(nterms, (size_t)-1);
  std::vector<long> rs_len(nterms, 0), cs_len(nterms, 0);
  for (int I = 0; I < ds->nrb; ++I)
    for (int J = 0; J < ds->ncb; ++J) {
      int p = I * ds->ncb + J;
      for (int t = sp->term_ptr[p]; t < sp->term_ptr[p + 1]; ++t) {
        const sgp_term& T = sp->terms[t];
        if (T.kind < 0 || T.kind > SGP_CONST) return fail("spec: unknown kernel kind");
        if (T.row_input < 0 || T.row_input >= sp->n_inputs || T.col_input < 0 ||
            T.col_input >= sp->n_inputs)
          return fail("spec: term input index out of range");
        const sgp_input& ri = sp->inputs[T.row_input];
        const sgp_input& ci = sp->inputs[T.col_input];
        if (ri.dim != ci.dim) return fail("spec: row / col input dimension mismatch");
        if (ri.n != ds->row_len[I] || ci.n != ds->col_len[J])
          return fail("spec: input length does not match block length");
        if (T.row_scale && ds->row_len[I] > 0) {
          rs_len[t] = ds->row_len[I];
          rs_off[t] = place(sizeof(double) * (size_t)rs_len[t]);
        }
        if (T.col_scale && ds->col_len[J] > 0) {
          cs_len[t] = ds->col_len[J];
          cs_off[t] = place(sizeof(double) * (size_t)cs_len[t]);
        }
        ds->pair_dmax[p] = std::max(ds->pair_dmax[p], pow2ceil((int)ri.dim));
      }
    }
  const size_t terms_off = place(sizeof(DevTerm) * (size_t)std::max(1, nterms));
  // ---- one cached device block, one pinned staging buffer
  char* d_base = (char*)pool_alloc(ctx, total);
  if (!d_base) {
    dspec_free(ds);
    return -2;
  }
  ds->d_bufs.push_back((double*)d_base);
  if (total > ctx->stage_cap) {
    if (ctx->h_stage) hipHostFree(ctx->h_stage);
    ctx->h_stage = nullptr;
    ctx->stage_cap = 0;
    size_t cap = std::max<size_t>(total + total / 2, 1u << 20);
    if (hipHostMalloc((void**)&ctx->h_stage, cap, hipHostMallocDefault) != hipSuccess) {
      (void)hipGetLastError();
      return fail("spec: hipHostMalloc failed (staging buffer)");
    }
    ctx->stage_cap = cap;
  }
  char* h_base = ctx->h_stage;
  // ---- pass 2: fill the staging buffer
  for (int k = 0; k < sp->n_inputs; ++k) {
    const sgp_input& in = sp->inputs[k];
    double* dst = (double*)(h_base + in_off[k]);
    if (in.ld == in.dim) {
      if (in.n > 0) memcpy(dst, in.x, sizeof(double) * (size_t)(in.dim * in.n));
    } else {
      for (long j = 0; j < in.n; ++j) memcpy(dst + j * in.dim, in.x + j * in.ld, sizeof(double) * (size_t)in.dim);
    }
  }
  DevTerm* h_terms = (DevTerm*)(h_base + terms_off);
  for (int I = 0; I < ds->nrb; ++I)
    for (int J = 0; J < ds->ncb; ++J) {
      int p = I * ds->ncb + J;
      for (int t = sp->term_ptr[p]; t < sp->term_ptr[p + 1]; ++t) {
        const sgp_term& T = sp->terms[t];
        const sgp_input& ri = sp->inputs[T.row_input];
        const sgp_input& ci = sp->inputs[T.col_input];
        DevTerm D;
        D.kind = T.kind;
        D.dim = (int)ri.dim;
        D.coef = T.coef;
        D.param = T.param;
        D.xr = (const double*)(d_base + in_off[T.row_input]);
        D.ldr = ri.dim;
        D.xc = (const double*)(d_base + in_off[T.col_input]);
        D.ldc = ci.dim;
        D.rs = nullptr;
        D.cs = nullptr;
        if (rs_len[t]) {
          memcpy(h_base + rs_off[t], T.row_scale, sizeof(double) * (size_t)rs_len[t]);
          D.rs = (const double*)(d_base + rs_off[t]);
        }
        if (cs_len[t]) {
          memcpy(h_base + cs_off[t], T.col_scale, sizeof(double) * (size_t)cs_len[t]);
          D.cs = (const double*)(d_base + cs_off[t]);
        }
        h_terms[t] = D;
        ds->h_terms.push_back(D);
        ds->term_row_input.push_back(T.row_input);
        ds->term_col_input.push_back(T.col_input);
      }
    }
  ds->d_terms = (DevTerm*)(d_base + terms_off);
  if (hipMemcpy(d_base, h_base, total, hipMemcpyHostToDevice) != hipSuccess) {
    (void)hipGetLastError();
    return fail("spec: upload failed");
  }
  *out = ds;
  return 0;
}

extern "C" int sgp_dspec_create(sgp_ctx* ctx, const sgp_cov_spec* sp, sgp_dspec** out) {
  CHECK_ARG(ctx && sp && out, "sgp_dspec_create: NULL argument");
  CtxScope scope(ctx);
  return dspec_create(ctx, sp, out);
}

// Internal: caller already holds ds->ctx (or is tearing it down from a failed create).
static void dspec_free(sgp_dspec* ds) {
  if (!ds) return;
  for (double* p : ds->d_bufs) pool_free(ds->ctx, p);
  delete ds;
}

// for the other translation units (f32.hip): the caller holds the context
int sgp_dspec_create_nolock(sgp_ctx* ctx, const sgp_cov_spec* sp, sgp_dspec** out) { return dspec_create(ctx, sp, out); }
void sgp_dspec_free_nolock(sgp_dspec* ds) { dspec_free(ds); }

extern "C" int sgp_dspec_destroy(sgp_dspec* ds) {
  if (!ds) return 0;
  if (tl_ctx == ds->ctx) {  // called from inside an entry point of the same context
    dspec_free(ds);
    return 0;
  }
  CtxScope scope(ds->ctx);
  dspec_free(ds);
  return 0;
}

// Assemble the block pairs of `ds` into the matrix whose element (r, c) (global indices,
// optionally shifted by row_shift) lives at Kv[r + row_shift + c*ld], restricted to the
// global tile window [tile_r_lo, tile_r_hi) x [tile_c_lo, tile_c_hi).
static int assemble(const sgp_dspec* ds, double* Kv, long ld, long tile_r_lo, long tile_r_hi,
                    long tile_c_lo, long tile_c_hi, int lower_only, int noise_kind, double sigma2,
                    const double* d_noise_diag, hipStream_t s) {
  for (int I = 0; I < ds->nrb; ++I) {
    if (ds->row_len[I] == 0) continue;
    for (int J = 0; J < ds->ncb; ++J) {
      if (ds->col_len[J] == 0) continue;
      if (lower_only && I < J) continue;
      long r0 = ds->row_off[I], nr = ds->row_len[I], c0 = ds->col_off[J], nc = ds->col_len[J];
      long trf = std::max(r0 / TILE, tile_r_lo), trl = std::min((r0 + nr - 1) / TILE + 1, tile_r_hi);
      long tcf = std::max(c0 / TILE, tile_c_lo), tcl = std::min((c0 + nc - 1) / TILE + 1, tile_c_hi);
      if (trf >= trl || tcf >= tcl) continue;
      int p = I * ds->ncb + J;
      int t0 = ds->term_ptr[p], t1 = ds->term_ptr[p + 1];
      int dmax = ds->pair_dmax[p];
      int per = assemble_terms_per_launch(dmax);  // terms per launch: <= 64 KiB of LDS
      int nk = (ds->symmetric && I == J) ? noise_kind : -1;
      if (t0 == t1) {
        CHECK_RC(launch_assemble_block(Kv, ld, r0, nr, c0, nc, ds->d_terms, 0, 1, lower_only, 0, nk,
                                       sigma2, d_noise_diag, trf, tcf, trl - trf, tcl - tcf, s));
        continue;
      }
      for (int t = t0; t < t1; t += per) {
        int cnt = std::min(per, t1 - t);
        CHECK_RC(launch_assemble_block(Kv, ld, r0, nr, c0, nc, ds->d_terms + t, cnt, dmax,
                                       lower_only, t > t0 ? 1 : 0, t == t0 ? nk : -1, sigma2,
                                       d_noise_diag, trf, tcf, trl - trf, tcl - tcf, s));
      }
    }
  }
  return 0;
}

// ---------------------------------------------------------------------------------------
// blocked Cholesky on the bordered matrix
// ---------------------------------------------------------------------------------------
// X <- X inv(Lkk)' for `rows` (multiple of 128) rows at X: the panel TRSM.
// A product with the explicit inverse W = inv(Lkk) alone is not backward stable -- on the ill-conditioned covariances smooth
// kernels produce it loses 2+ digits against LAPACK dtrsm and can push the Schur complement indefinite -- so the solve is
// panel_solve_kernel: blocked substitution over 16-column blocks, each diagonal solve = inverse product + one refinement
// step; one launch, no inv(Lkk) needed.  (Rounds 2 - 5 kept the explicit-inverse product and a 128-level refinement as
// SGP_REFINE = 0 / 2 A/B modes; removed in round 6.)
// `inv` points at the eight 16x16 inverse diagonal blocks (block c at inv + c * inv_cstride, element [m][k] at
// + k * inv_kstride + m): what potrf_diag just wrote (ctx scratch or a kept copy).
static int solve_rows(sgp_ctx* ctx, double* X, long ldx, long rows, const double* Lkk, long ldl, const double* inv,
                      long inv_cstride, long inv_kstride, hipStream_t s) {
  (void)ctx;
  if (rows <= 0) return 0;
  // (structured models: the 128-row tiles of this block column that are structurally zero are left alone -- the record
  // chol_bordered / the sharded driver set for the matrix or packed panel being factored, gemm_nt.hip: strip_skip_for)
  const StripSkip sk = strip_skip_for(X, ldx);
  return launch_panel_solve(X, ldx, rows, Lkk, ldl, inv, inv_cstride, inv_kstride, s, sk.nz ? &sk : nullptr);
}

// Factor one column panel in place (inner right-looking loop, nb = 128).
// P: m x w, top w x w block is the diagonal block.  d_invstore: optional array that keeps the
// eight 16x16 inverse diagonal blocks of every 128-block (INVD_STRIDE doubles per block, the
// layout potrf_diag writes) for later solves against the factor; else ctx scratch.
// Which fused update + potrf_diag launches a factorisation of `n` columns (or a panel of `n` rows) uses: the
// context's SGP_FUSE_POTRF below SGP_FUSE_MAX_N, nothing from there on unless bit 2 asks for every size.
// `serial`: the factorisation runs without the two-stream look-ahead (every kernel has the chip to itself) -- there
// the fused launches pay at every size: the diagonal block's pivot chain runs under the other tiles of its own
// launch (N = 65536: 1541 -> 1525 ms).
static int fuse_mode(const sgp_ctx* ctx, long n, bool serial = false) {
  if (n < ctx->fuse_max_n || serial) return ctx->fuse_potrf;
  return (ctx->fuse_potrf & 4) ? (ctx->fuse_potrf | 3) : 0;
}
struct FuseScope {   // panel_factor reads ctx->fuse_now; restored on every exit path
  sgp_ctx* c;
  int saved;
  FuseScope(sgp_ctx* c_, int v) : c(c_), saved(c_->fuse_now) { c->fuse_now = v; }
  ~FuseScope() { c->fuse_now = saved; }
};

// first_done: the panel's first diagonal block is already factored -- the trailing update that produced it had
// potrf_diag fused into the workgroup of that tile (ctx->fuse_potrf, gemm_nt.hip: gemm_nt_dma_potrf_kernel).
static int panel_factor(sgp_ctx* ctx, double* P, long ld, long m, long w, long g0, double* d_slots,
                        int* d_info, double* d_invstore, hipStream_t s, bool first_done = false) {
  // Fused inner updates (SGP_FUSE_POTRF bit 0): the K = 128 update with block column j also factors the diagonal
  // block of block column j + 1 in the workgroup that updates it, so the 128-pivot chain runs under the rest of
  // the update instead of after it and one launch per block column disappears.
  const bool fuse = (ctx->fuse_now & 1) != 0;
  bool diag_done = first_done;
  for (long j = 0; j < w; j += TILE) {
    double* D = P + j + j * ld;
    double* invd = d_invstore ? d_invstore + (j / TILE) * INVD_STRIDE : ctx->d_invd;
    if (!diag_done) CHECK_RC(launch_potrf_diag(D, ld, invd, d_slots + j / TILE, d_info, g0 + j, s));
    diag_done = false;
    long mrest = m - j - TILE;
    if (mrest > 0) {
      double* A21 = P + (j + TILE) + j * ld;
      CHECK_RC(solve_rows(ctx, A21, ld, mrest, D, ld, invd, 256, 16, s));  // L21 = A21 * L11^-T
      long wrest = w - j - TILE;
      if (wrest > 0) {
        if (fuse) {
          // the scratch inverse blocks (no d_invstore) are free again: the row solve that read block column j's
          // ran before this launch on the same stream
          double* invn = d_invstore ? d_invstore + (j / TILE + 1) * INVD_STRIDE : ctx->d_invd;
          CHECK_RC(launch_gemm_nt_potrf(A21, ld, P + (j + TILE) + (j + TILE) * ld, ld, mrest, wrest, TILE, 0,
                                        (ctx->fuse_now & 8) ? 1 : 0, invn, d_slots + j / TILE + 1, d_info,
                                        g0 + j + TILE, s));
          diag_done = true;
        } else {
          CHECK_RC(launch_gemm_nt(A21, ld, A21, ld, P + (j + TILE) + (j + TILE) * ld, ld, mrest, wrest, TILE, -1.0, 1.0,
                                  0, 0, 0, s));
        }
      }
    }
  }
  return 0;
}

static double update_flops(long m, long nc, long k) {
  // algorithmic flops of C[lower, m x nc] -= P P': triangle of the square part + rows below
  double sq = (double)k * (double)nc * (double)(nc + 1);
  double below = 2.0 * (double)k * (double)(m - nc) * (double)nc;
  return sq + below;
}

// timed (optional) trailing-update launch
struct FusedDiag {   // the diagonal block a fused trailing update factors: tile (0, 0) of its C
  double* invd;
  double* slot;
  int* info;
  long gcol0;
  int handoff;   // accumulators -> potrf_diag's LDS layout directly (SGP_FUSE_POTRF bit 3)
};
static int launch_update_kernel(const double* P, long ld, double* C, long M, long Nc, long K, hipStream_t s,
                                const FusedDiag* fz) {
  if (fz)
    return launch_gemm_nt_potrf(P, ld, C, ld, M, Nc, K, 1, fz->handoff, fz->invd, fz->slot, fz->info, fz->gcol0, s);
  return launch_gemm_nt_update(P, ld, C, ld, M, Nc, K, s);
}
static int launch_update(sgp_ctx* ctx, const double* P, long ld, double* C, long M, long Nc, long K,
                         hipStream_t s, const FusedDiag* fz = nullptr) {
  if (M <= 0 || Nc <= 0) return 0;
  if (ctx->time_updates) {
    // the context owns both events from the moment they exist (destroyed with the next timed call or the
    // context), whatever happens below
    hipEvent_t e0 = nullptr, e1 = nullptr;
    SGP_HIP(hipEventCreate(&e0));
    ctx->ev.push_back(e0);
    if (hipEventCreate(&e1) != hipSuccess) {
      ctx->ev.pop_back();
      hipEventDestroy(e0);
      set_error("hipEventCreate failed");
      return -2;
    }
    ctx->ev.push_back(e1);
    ctx->ev_flops.push_back(update_flops(M, Nc, K) * ctx->sz_live_fraction(P, ld, C, M, Nc, K));
    SGP_HIP(hipEventRecord(e0, s));
    CHECK_RC(launch_update_kernel(P, ld, C, M, Nc, K, s, fz));
    SGP_HIP(hipEventRecord(e1, s));
    return 0;
  }
  return launch_update_kernel(P, ld, C, M, Nc, K, s, fz);
}

// Middle blocking level: an outer panel of w columns wider than `wmid` is factored as sub-panels of wmid columns, each
// followed by ONE K = wmid update of the panel's remaining columns (lower trapezoid; the next sub-panel's first diagonal
// block fused into it).  The outer trailing updates then run with K = w -- half the C-tile passes per flop of K = w / 2,
// i.e. half the per-tile prologue / epilogue share -- while the shallow K = 128 work stays that of a wmid-wide panel.
static int panel_factor_mid(sgp_ctx* ctx, double* P, long ld, long m, long w, long g0, double* d_slots, int* d_info,
                            double* d_invstore, hipStream_t s, bool first_done, long wmid) {
  if (wmid < TILE || wmid >= w) return panel_factor(ctx, P, ld, m, w, g0, d_slots, d_info, d_invstore, s, first_done);
  // recursive halving down to wmid: the left half, ONE update of the right half with it (K = w / 2), the right half
  const bool fuse_mid = (ctx->fuse_now & 2) != 0;
  const long wl = std::max(wmid, (w / 2 + wmid - 1) / wmid * wmid), c = wl, rest = w - wl;
  CHECK_RC(panel_factor_mid(ctx, P, ld, m, wl, g0, d_slots, d_info, d_invstore, s, first_done, wmid));
  if (rest <= 0) return 0;
  const FusedDiag fz = {d_invstore ? d_invstore + (c / TILE) * INVD_STRIDE : ctx->d_invd, d_slots + c / TILE, d_info,
                        g0 + c, (ctx->fuse_now & 8) ? 1 : 0};
  CHECK_RC(launch_update(ctx, P + c, ld, P + c + c * ld, m - c, rest, wl, s, fuse_mid ? &fz : nullptr));
  return panel_factor_mid(ctx, P + c + c * ld, ld, m - c, rest, g0 + c, d_slots + c / TILE, d_info,
                          d_invstore ? d_invstore + (c / TILE) * INVD_STRIDE : nullptr, s, fuse_mid, wmid);
}

// Which schedule factors n_pad columns (measured on MI355X, profiles/archive/r03_dataflow.md; Matern-5/2, D = 8, whole logpdf):
//   n_pad <  3072            launches (one outer panel; the dataflow chain of 68 us per 128 columns loses to the fused
//                            launches' 62: N = 2048 1.18 vs 1.25 ms)
//   3072 <= n_pad < 24576    dataflow, one workgroup per CU (256 VGPRs: no spills in the chain tasks, and a chain task never
//                            shares its CU with a contraction): N = 4096 2.51 -> 2.41 ms, 8192 7.7 -> 5.1, 16384 32.7 -> 27.8
//   24576 <= n_pad < 65536   dataflow, two workgroups per CU (the contractions are the work): 32768 218.7 -> 204.8 ms
//   n_pad >= 65536           launches, serial schedule with deep outer blocking: the lock-step trailing updates share
//                            every operand k slice through L2, the desynchronised contractions of the dataflow kernel
//                            stream theirs from HBM and the clock pays for it (-11 %, profiles/archive/r03_experiments/)
// SGP_DATAFLOW = 0 / 1 forces never / always; SGP_DF_MIN_N, SGP_DF_MAX_N, SGP_DF_FAT_MAX_N move the limits.
// (Round 4 also measured XCD-affine task queues for the dataflow kernel -- profiles/r04_experiments/dataflow_xcd_queues.md: handing
// a queue's patches to the free workgroups of its XCD does not make them share operand panels, they start their tiles whenever
// they become free; +3.6 % at 32768 columns.  The code was removed in round 6.)
// Hybrid schedule (round 5; the round-4 verdict's item 4): the look-ahead schedule of the launches with every outer PANEL (2048
// columns) factored by ONE launch of the dataflow kernel on the panel -- its diagonal chain and the row solves below it as
// tile tasks with tile-level dependencies (the rows below the panel's diagonal block are that kernel's "bordered rows") --
// while the trailing updates stay lock-step launches of K = 2048 (operand sharing through L2, 0.85 in situ).  The panel
// kernel runs beside the previous panel's trailing update: the chain that the serial-deep schedule pays in full (57 ms at
// N = 65536) and that the launch-based look-ahead ran 2.5 - 3 x slower beside an update (many small dependent launches) is
// hidden.  Same arithmetic order per tile (k ascending): bit-identical.  Measured (profiles/r05_experiments/hybrid.md):
// N = 65536 1454 -> 1395 ms, 32768 197 -> 182 ms, 16384 26.7 -> 28.6 (stays on the dataflow kernel).
// The gradient path's factorisations (`grow`: an identity border rides along, 3 x the flops per column, no whole-matrix
// dataflow form) take it from 16384 columns on: N = 16384 82.2 -> 81.1 ms, 24576 260.8 -> 253.5, 32768 605.8 -> 582.0
// (profiles/r05_experiments/hybrid_grad.txt).
static bool use_hybrid(const sgp_ctx* ctx, long n_pad, bool grow = false) {
  if (ctx->hybrid == 0 || n_pad < 4096) return false;
  if (ctx->hybrid == 1) return true;
  // by size -- unless the caller pinned another schedule (SGP_DATAFLOW = 0 / 1, SGP_LOOKAHEAD = 0)
  return ctx->dataflow < 0 && ctx->lookahead != 0 && n_pad >= (grow ? ctx->hybrid_grow_min_n : ctx->hybrid_min_n);
}
static bool use_dataflow(const sgp_ctx* ctx, long n_pad) {
  if (ctx->dataflow == 0) return false;
  return ctx->dataflow == 1 || (n_pad >= ctx->df_min_n && n_pad < ctx->df_max_n);
}
extern "C" int sgp_ctx_factor_work(sgp_ctx* ctx, double* executed, double* dense) {
  CHECK_ARG(ctx && executed && dense, "sgp_ctx_factor_work: NULL argument");
  if (ctx->multi) return sgp_multi_factor_work(ctx->multi, executed, dense);
  *executed = ctx->sz_executed;
  *dense = ctx->sz_dense;
  return 0;
}
// A block order that keeps the factor sparse (sthenomi.h): greedy minimum fill on the block graph, host only.
extern "C" int sgp_cov_spec_suggest_order(const sgp_cov_spec* sp, int32_t* perm_out, int32_t* changes) {
  CHECK_ARG(sp && perm_out, "sgp_cov_spec_suggest_order: NULL argument");
  CHECK_ARG(sp->symmetric && sp->n_row_blocks == sp->n_col_blocks && sp->n_row_blocks >= 1,
            "sgp_cov_spec_suggest_order: the spec must be symmetric (cov(f, x))");
  const int nb = sp->n_row_blocks;
  std::vector<std::vector<char>> adj((size_t)nb, std::vector<char>((size_t)nb, 0));
  for (int I = 0; I < nb; ++I)
    for (int J = 0; J < nb; ++J) {
      const int p = I * nb + J, q = J * nb + I;
      if (I != J && (sp->term_ptr[p + 1] > sp->term_ptr[p] || sp->term_ptr[q + 1] > sp->term_ptr[q])) adj[(size_t)I][(size_t)J] = 1;
    }
  // weighted fill of eliminating the blocks in a given order (what the order costs: lengths of the block pairs that fill in)
  auto total_fill = [&](const std::vector<int>& order) {
    std::vector<std::vector<char>> a = adj;
    std::vector<char> gone((size_t)nb, 0);
    double fill = 0;
    for (int v : order) {
      for (int x = 0; x < nb; ++x)
        for (int y = x + 1; y < nb; ++y)
          if (!gone[(size_t)x] && !gone[(size_t)y] && x != v && y != v && a[(size_t)v][(size_t)x] && a[(size_t)v][(size_t)y] &&
              !a[(size_t)x][(size_t)y]) {
            a[(size_t)x][(size_t)y] = a[(size_t)y][(size_t)x] = 1;
            fill += (double)sp->row_len[x] * (double)sp->row_len[y];
          }
      gone[(size_t)v] = 1;
    }
    return fill;
  };
  std::vector<std::vector<char>> a = adj;
  std::vector<char> left((size_t)nb, 1);
  std::vector<int> order;
  for (int step = 0; step < nb; ++step) {
    int best = -1;
    double best_fill = 0;
    for (int v = 0; v < nb; ++v) {
      if (!left[(size_t)v]) continue;
      double fill = 0;
      for (int x = 0; x < nb; ++x)
        for (int y = x + 1; y < nb; ++y)
          if (left[(size_t)x] && left[(size_t)y] && x != v && y != v && a[(size_t)v][(size_t)x] && a[(size_t)v][(size_t)y] &&
              !a[(size_t)x][(size_t)y])
            fill += (double)sp->row_len[x] * (double)sp->row_len[y];
      const bool better = best < 0 || fill < best_fill || (fill == best_fill && sp->row_len[v] < sp->row_len[best]);
      if (better) {
        best = v;
        best_fill = fill;
      }
    }
    for (int x = 0; x < nb; ++x)
      for (int y = x + 1; y < nb; ++y)
        if (left[(size_t)x] && left[(size_t)y] && x != best && y != best && a[(size_t)best][(size_t)x] && a[(size_t)best][(size_t)y])
          a[(size_t)x][(size_t)y] = a[(size_t)y][(size_t)x] = 1;
    left[(size_t)best] = 0;
    order.push_back(best);
  }
  for (int k = 0; k < nb; ++k) perm_out[k] = order[(size_t)k];
  if (changes) {
    std::vector<int> ident((size_t)nb);
    for (int k = 0; k < nb; ++k) ident[(size_t)k] = k;
    *changes = total_fill(order) < total_fill(ident) ? 1 : 0;
  }
  return 0;
}
extern "C" const char* sgp_ctx_factor_schedule(sgp_ctx* ctx, int64_t N) {
  if (!ctx || N < 1) return "";
  const long n_pad = rup(N, TILE);
  if (use_hybrid(ctx, n_pad)) return "hybrid";
  if (use_dataflow(ctx, n_pad)) return n_pad < ctx->df_fat_max_n ? "dataflow-fat" : "dataflow";
  const bool la = ctx->lookahead && (n_pad < ctx->la_max_n || ctx->lookahead == 2);
  if (la) return n_pad <= 4096 && ctx->wout <= 0 ? "launches-one-panel" : "launches-lookahead";
  return n_pad >= 65536 ? "launches-serial-deep" : "launches-serial";
}

// Two-level right-looking Cholesky of the bordered matrix with one-panel look-ahead:
// the outer panel J+1 is updated and factored on the (high-priority) panel stream while the
// rest of the trailing matrix is still being updated with panel J on the update stream.
// grow > 0: bordered rows >= n_pad hold a matrix that is upper triangular by tile from row `grow` on
// (the identity rows of the gradient path: row grow + i stays zero left of column i), so panel
// J0..J0+wj only touches rows < grow + J0 + wj.
// The tile-level pattern of the factor of the matrix a caller is about to factor (structural zeros, common.h).
struct SzMask {
  const sz_word* d_nz = nullptr;   // device rows (ctx->d_sz), nullptr: dense
  int words = 0;
};
struct SzScope {   // the launch-based updates read the pattern through gemm_nt.hip's per-thread record
  sgp_ctx* c;
  bool on;
  SzScope(sgp_ctx* ctx, const double* A, long ld, long ncols, const SzMask* sz) : c(ctx), on(sz && sz->d_nz) {
    if (on) {
      gemm_set_structure(A, ld, sz->d_nz, sz->words, ncols);
      c->sz_base = A;
      c->sz_ld = ld;
      c->sz_words = sz->words;
    }
  }
  ~SzScope() {
    if (on) {
      gemm_set_structure(nullptr, 0, nullptr, 0);
      c->sz_base = nullptr;
    }
  }
};

// scratch of a dataflow launch on this context (grow-only): progress words for nb matrices of m_tot rows and -- when the caller
// keeps no inverse diagonal blocks itself -- room for those of inv_cols columns (0: not needed)
static int df_scratch(sgp_ctx* ctx, long m_tot, int nb, long inv_cols, hipStream_t s) {
  const long need_state = df_state_words(m_tot, nb), need_inv = (inv_cols / TILE) * INVD_STRIDE;
  if (need_state > ctx->n_df_state) {
    SGP_HIP(hipStreamSynchronize(s));
    if (ctx->d_df_state) hipFree(ctx->d_df_state);
    ctx->d_df_state = nullptr;
    ctx->n_df_state = 0;
    SGP_HIP(hipMalloc(&ctx->d_df_state, sizeof(int) * need_state));
    ctx->n_df_state = need_state;
  }
  if (need_inv > ctx->n_df_inv) {
    SGP_HIP(hipStreamSynchronize(s));
    if (ctx->d_df_inv) hipFree(ctx->d_df_inv);
    ctx->d_df_inv = nullptr;
    ctx->n_df_inv = 0;
    SGP_HIP(hipMalloc(&ctx->d_df_inv, sizeof(double) * need_inv));
    ctx->n_df_inv = need_inv;
  }
  return 0;
}

// ---- one function per schedule of the blocked Cholesky (round 6: rounds 3 - 5 had grown ONE loop that threaded six booleans --
// hybrid, la, deep, fuse_outer, grow, sz -- through itself); chol_bordered below only chooses.  Every schedule performs the
// same arithmetic per tile (k ascending): the factor is bit-identical whichever runs (tests/test_gpu_dataflow.py).

// (1) the whole bordered factorisation as ONE launch of the dataflow kernel (chol_df.hip): persistent workgroups, tile-level
// dependencies instead of launches, streams and events.  3072 <= n_pad < 24576 by default.
static int chol_dataflow_whole(sgp_ctx* ctx, double* A, long ld, long n_pad, long m_tot, double* d_wall, hipStream_t s,
                               const SzMask* sz) {
  CHECK_RC(df_scratch(ctx, m_tot, 1, d_wall ? 0 : n_pad, s));
  {
    const int fat = n_pad < ctx->df_fat_max_n ? 1 : 0;
    if (getenv("SGP_DF_STATS") && !ctx->d_df_stats)   // + 8 stamps for each of up to 4096 tile columns
      SGP_HIP(hipMalloc(&ctx->d_df_stats, sizeof(long long) * 8 * ((size_t)ctx->df_wgs + 4096)));
    long long* d_cols = ctx->d_df_stats && n_pad / TILE <= 4096 ? ctx->d_df_stats + 8 * (size_t)ctx->df_wgs : nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (ctx->d_df_stats) {
      SGP_HIP(hipEventCreate(&e0));
      SGP_HIP(hipEventCreate(&e1));
      SGP_HIP(hipEventRecord(e0, s));
    }
    CHECK_RC(launch_chol_dataflow(A, ld, n_pad, m_tot, ctx->d_df_state, d_wall ? d_wall : ctx->d_df_inv, ctx->d_slots,
                                  ctx->d_info, ctx->df_wgs, ctx->df_timeout_s, s, ctx->d_df_stats, d_cols, fat,
                                  sz ? sz->d_nz : nullptr, sz ? sz->words : 0));
    if (ctx->d_df_stats) {   // diagnosis only: drains the stream
      hipEventRecord(e1, s);
      hipStreamSynchronize(s);
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      hipEventDestroy(e0);
      hipEventDestroy(e1);
      std::vector<long long> h((size_t)8 * ctx->df_wgs);
      SGP_HIP(hipMemcpy(h.data(), ctx->d_df_stats, sizeof(long long) * h.size(), hipMemcpyDeviceToHost));
      double sum[8] = {0};
      long nw = 0;
      long off_xcd = 0;
      for (int w = 0; w < ctx->df_wgs; ++w) {
        if (h[(size_t)8 * w + 1] == 0) continue;
        ++nw;
        if ((int)((h[(size_t)8 * w] >> 40) & 15) != (w & 7)) ++off_xcd;   // not on XCD (id % 8)
        h[(size_t)8 * w] &= (1LL << 40) - 1;
        for (int q = 0; q < 8; ++q) sum[q] += (double)h[(size_t)8 * w + q];
      }
      fprintf(stderr, "dataflow: %ld of %ld workgroups NOT on XCD (id %% 8)\n", off_xcd, nw);
      const double us = 0.01 / (double)std::max<long>(nw, 1);   // ticks (100 MHz) -> us, averaged over the workgroups
      fprintf(stderr,
              "dataflow n_pad=%ld m_tot=%ld: %.3f ms, %ld workgroups, %.0f tasks; per workgroup (us): in kernel %.1f | "
              "contraction %.1f (of which waiting %.1f) | wait diag %.1f | potrf %.1f | solve %.1f | publish+dequeue %.1f\n",
              n_pad, m_tot, ms, nw, sum[0], sum[1] * us, sum[2] * us, sum[7] * us, sum[3] * us, sum[4] * us, sum[5] * us,
              sum[6] * us);
      {   // spread over the workgroups: time in the kernel and tasks taken (min / quartiles / max)
        std::vector<double> tk, nt;
        for (int w = 0; w < ctx->df_wgs; ++w)
          if (h[(size_t)8 * w + 1]) {
            tk.push_back((double)h[(size_t)8 * w + 1] * 0.01);
            nt.push_back((double)h[(size_t)8 * w]);
          }
        std::sort(tk.begin(), tk.end());
        std::sort(nt.begin(), nt.end());
        auto q = [](const std::vector<double>& v, double f) { return v.empty() ? 0.0 : v[(size_t)(f * (double)(v.size() - 1))]; };
        fprintf(stderr, "  per workgroup: in kernel (us) min %.0f q1 %.0f median %.0f q3 %.0f max %.0f | tasks min %.0f median %.0f max %.0f\n",
                q(tk, 0), q(tk, 0.25), q(tk, 0.5), q(tk, 0.75), q(tk, 1), q(nt, 0), q(nt, 0.5), q(nt, 1));
      }
      if (d_cols && n_pad / TILE >= 3) {
        const long T = n_pad / TILE;
        std::vector<long long> c((size_t)8 * T);
        SGP_HIP(hipMemcpy(c.data(), d_cols, sizeof(long long) * c.size(), hipMemcpyDeviceToHost));
        // the chain, column j -> j + 1 (averages over the columns, us): diagonal tile published [4] -> the task below
        // sees it [5] -> solved [6] -> published [7] -> next diagonal task sees it [1'] -> last k block done [2'] ->
        // factored [3'] -> published [4']
        double seg[7] = {0};
        long cnt = 0;
        for (long j = 1; j + 2 < T; ++j) {
          const long long* a0 = &c[(size_t)8 * j];
          const long long* a1 = &c[(size_t)8 * (j + 1)];
          if (!a0[4] || !a0[5] || !a0[6] || !a0[7] || !a1[1] || !a1[2] || !a1[3] || !a1[4]) continue;
          seg[0] += (double)(a0[5] - a0[4]);
          seg[1] += (double)(a0[6] - a0[5]);
          seg[2] += (double)(a0[7] - a0[6]);
          seg[3] += (double)(a1[1] - a0[7]);
          seg[4] += (double)(a1[2] - a1[1]);
          seg[5] += (double)(a1[3] - a1[2]);
          seg[6] += (double)(a1[4] - a1[3]);
          ++cnt;
        }
        if (cnt) {
          const double u = 0.01 / (double)cnt;
          fprintf(stderr,
                  "  chain per column (us, %ld columns): diag published -> seen below %.1f | solve %.1f | publish %.1f | -> "
                  "seen by next diag %.1f | last k block %.1f | potrf %.1f | publish %.1f | sum %.1f\n",
                  cnt, seg[0] * u, seg[1] * u, seg[2] * u, seg[3] * u, seg[4] * u, seg[5] * u, seg[6] * u,
                  (seg[0] + seg[1] + seg[2] + seg[3] + seg[4] + seg[5] + seg[6]) * u);
        }
      }
    }
  }
  return 0;
}

// what the two launch-based schedules share: the trailing updates of outer panel [J0, J0 + wj) -- with look-ahead (`la`) the
// next panel's columns on the panel stream s (optionally factoring that panel's first diagonal block in the same launch, fz)
// and the rest on the update stream sB; without it one launch on s
struct OuterSweep {
  sgp_ctx* ctx;
  double* A;
  long ld, n_pad, m_tot, grow;
  hipStream_t s, sB;
  bool la;
  bool rest_pending = false;
  long rows(long J0, long wj) const { return grow > 0 ? std::min(m_tot, grow + J0 + wj) : m_tot; }   // rows panel J0 touches
  int begin() {
    if (!la) return 0;
    // the update stream must see everything enqueued on s so far (assembly)
    SGP_HIP(hipEventRecord(ctx->ev_panel, s));
    SGP_HIP(hipStreamWaitEvent(sB, ctx->ev_panel, 0));
    return 0;
  }
  int updates(long J0, long wj, long WOUT, const FusedDiag* fz) {
    const long c0 = J0 + wj, m_eff = rows(J0, wj);
    const long w1 = std::min(WOUT, n_pad - c0), c1 = c0 + w1;   // the next panel
    if (la) {
      SGP_HIP(hipEventRecord(ctx->ev_panel, s));
      // look-ahead: next panel's columns on the panel stream (after the previous rest update)
      if (rest_pending) SGP_HIP(hipStreamWaitEvent(s, ctx->ev_rest, 0));
      CHECK_RC(launch_update(ctx, A + c0 + J0 * ld, ld, A + c0 + c0 * ld, m_eff - c0, w1, wj, s, fz));
      if (c1 < n_pad) {
        SGP_HIP(hipStreamWaitEvent(sB, ctx->ev_panel, 0));
        CHECK_RC(launch_update(ctx, A + c1 + J0 * ld, ld, A + c1 + c1 * ld, m_eff - c1, n_pad - c1, wj, sB));
        SGP_HIP(hipEventRecord(ctx->ev_rest, sB));
        rest_pending = true;
      }
    } else {
      if (rest_pending) {
        SGP_HIP(hipStreamWaitEvent(s, ctx->ev_rest, 0));
        rest_pending = false;
      }
      CHECK_RC(launch_update(ctx, A + c0 + J0 * ld, ld, A + c0 + c0 * ld, m_eff - c0, n_pad - c0, wj, s, fz));
    }
    return 0;
  }
  int end() {
    if (la && rest_pending) SGP_HIP(hipStreamWaitEvent(s, ctx->ev_rest, 0));
    return 0;
  }
};

// (2) hybrid (round 5; use_hybrid above): the look-ahead schedule of the launches with every outer PANEL (2048 columns) factored
// by ONE launch of the dataflow kernel on the panel -- also for the gradient path's border (`grow`: a panel launch takes the
// rows its panel touches, the identity rows among them as bordered rows; only the few tiles above the identity's diagonal
// inside one panel are multiplied out, or skipped when the caller's pattern covers them).  From 24576 columns on.
static int chol_hybrid(sgp_ctx* ctx, double* A, long ld, long n_pad, long m_tot, double* d_wall, hipStream_t s, long grow,
                       const SzMask* sz) {
  CHECK_RC(df_scratch(ctx, m_tot, 1, d_wall ? 0 : n_pad, s));
  const long WOUT = std::min(ctx->hybrid_w, n_pad);
  // (Round 6 measured panel GROUPS here -- far updates once per G panels with K = G x 2048, near classes every step, the sharded
  // sweep's scheme and the round-5 verdict's item 3: c5 1373 -> 1466 / 1476 / 1472 ms at G = 2 / 3 / 4, bit-identical; the near
  // launches cost more than the deeper far launches save.  profiles/r06_experiments/hybrid_groups.md; removed.)
  // (SGP_HYBRID_SERIAL=1, measurement only: the far updates on the panel stream too -- every kernel has the chip alone)
  OuterSweep sw{ctx, A, ld, n_pad, m_tot, grow, s, ctx->hybrid_serial ? s : ctx->stream2, true};
  CHECK_RC(sw.begin());
  for (long J0 = 0; J0 < n_pad; J0 += WOUT) {
    const long wj = std::min(WOUT, n_pad - J0);
    // the panel as ONE launch of persistent workgroups: its diagonal chain and the row solves below it with tile-level
    // dependencies (the rows below the panel's diagonal block are the kernel's "bordered rows")
    CHECK_RC(launch_chol_dataflow(A + J0 + J0 * ld, ld, wj, sw.rows(J0, wj) - J0, ctx->d_df_state,
                                  (d_wall ? d_wall : ctx->d_df_inv) + (J0 / TILE) * INVD_STRIDE, ctx->d_slots + J0 / TILE,
                                  ctx->d_info, ctx->hybrid_wgs, ctx->df_timeout_s, s, nullptr, nullptr, ctx->hybrid_fat,
                                  sz ? sz->d_nz : nullptr, sz ? sz->words : 0, J0));
    if (J0 + wj >= n_pad) break;
    CHECK_RC(sw.updates(J0, wj, WOUT, nullptr));   // (the dataflow panel factors its own first block: nothing fused)
  }
  return sw.end();
}

// (3) launches: the two-level right-looking schedule of rounds 1 - 3 -- n_pad < 3072, and whenever the other two are switched
// off (SGP_DATAFLOW=0 SGP_HYBRID=0: the dataflow time-out fallback, the A/B tests).
//   outer panel width, measured (profiles/archive/r02_summary.md): one panel for n_pad <= 4096 (the outer level only adds
//   launches there: 1.52 -> 1.30 ms at N = 2048), 1024 up to 8192, 512 in the mid range where the panel stream is the
//   critical path (N = 16384: 34.8 vs 35.4 ms), 1024 from 32768 on (halves the C-tile traffic per flop of the big updates);
//   look-ahead below 65536 columns only (la_max_n): at N = 65536 the overlap hides ~70 ms of panel chain but the sharing costs
//   the trailing updates as much; SGP_LOOKAHEAD=2: look-ahead at every size;
//   serial-deep at n_pad >= 65536 (round 3): outer panels of 4096 columns factored by recursive halving down to 1024
//   (panel_factor_mid) -- the big trailing updates run with K = 4096, the mid updates with K = 2048 / 1024;
//   SGP_FUSE_POTRF bit 1: the trailing update that finishes the next panel's first diagonal block factors that block in the
//   same launch, below 32768 columns where the panel chain is the critical path (fuse_max_n; bit 2: at every size).
static int chol_launches(sgp_ctx* ctx, double* A, long ld, long n_pad, long m_tot, double* d_wall, hipStream_t s, long grow) {
  const bool la = ctx->lookahead && s == ctx->stream && (n_pad < ctx->la_max_n || ctx->lookahead == 2);
  const bool deep = !la && n_pad >= 65536;
  const long WOUT = ctx->wout > 0 ? ctx->wout
                    : n_pad <= 4096 ? n_pad
                    : n_pad <= 8192 ? WOUT_LARGE
                    : deep ? 4 * WOUT_LARGE
                    : n_pad >= 32768 ? WOUT_LARGE
                                     : WOUT_SMALL;
  const long WMID = ctx->wmid > 0 ? ctx->wmid : (deep && ctx->wout <= 0 ? WOUT_LARGE : 0);
  FuseScope fuse_scope(ctx, fuse_mode(ctx, n_pad, !la));
  const bool fuse_outer = (ctx->fuse_now & 2) != 0;
  OuterSweep sw{ctx, A, ld, n_pad, m_tot, grow, s, la ? ctx->stream2 : s, la};
  CHECK_RC(sw.begin());
  bool first_done = false;
  for (long J0 = 0; J0 < n_pad; J0 += WOUT) {
    const long wj = std::min(WOUT, n_pad - J0);
    CHECK_RC(panel_factor_mid(ctx, A + J0 + J0 * ld, ld, sw.rows(J0, wj) - J0, wj, J0, ctx->d_slots + J0 / TILE, ctx->d_info,
                              d_wall ? d_wall + (J0 / TILE) * INVD_STRIDE : nullptr, s, first_done, WMID));
    const long c0 = J0 + wj;
    if (c0 >= n_pad) break;
    const FusedDiag fz_next = {d_wall ? d_wall + (c0 / TILE) * INVD_STRIDE : ctx->d_invd, ctx->d_slots + c0 / TILE,
                               ctx->d_info, c0, (ctx->fuse_now & 8) ? 1 : 0};
    first_done = fuse_outer;
    CHECK_RC(sw.updates(J0, wj, WOUT, fuse_outer ? &fz_next : nullptr));
  }
  return sw.end();
}

// The dispatcher.  grow > 0 (the gradient path): bordered rows >= n_pad hold a matrix that is upper triangular by tile from row
// `grow` on (the identity rows: row grow + i stays zero left of column i), so panel J0 .. J0 + wj only touches rows
// < grow + J0 + wj; its pattern (sz) covers the identity rows (sz_pattern's grad_border form).
static int chol_bordered(sgp_ctx* ctx, double* A, long ld, long n_pad, long m_tot, double* d_wall,
                         hipStream_t s, long grow = 0, const SzMask* sz = nullptr) {
  CHECK_ARG(n_pad / TILE <= ctx->n_slots, "matrix too large for the logdet slot buffer");
  SzScope sz_scope(ctx, A, ld, n_pad, sz);
  const bool hybrid = s == ctx->stream && use_hybrid(ctx, n_pad, grow != 0);
  if (!hybrid && grow == 0 && use_dataflow(ctx, n_pad)) return chol_dataflow_whole(ctx, A, ld, n_pad, m_tot, d_wall, s, sz);
  // the launch-based updates read the pattern through gemm_nt.hip's per-thread record; the compacted id maps of consecutive
  // structured launches of ONE stream may share a scratch buffer, the update stream of a look-ahead gets a map of its own
  if (sz_scope.on && s == ctx->stream) {
    const bool two = hybrid || (ctx->lookahead && (n_pad < ctx->la_max_n || ctx->lookahead == 2));
    gemm_set_structure(A, ld, sz->d_nz, sz->words, n_pad, ctx->d_szmap, ctx->n_szmap / 2, 0, two ? ctx->stream2 : nullptr,
                       two ? ctx->d_szmap + ctx->n_szmap / 2 : nullptr);
  }
  if (hybrid) return chol_hybrid(ctx, A, ld, n_pad, m_tot, d_wall, s, grow, sz);
  return chol_launches(ctx, A, ld, n_pad, m_tot, d_wall, s, grow);
}

// K + Sigma_y (lower tiles), identity padding, bordered rows
// Structural zeros: the tile pattern of the factor of K + Sigma_y for a symmetric spec.
//   block level : block pair (I, J) has terms, or not (flatten.py emits none for independent processes: exact zeros)
//   tile level  : tile (ti, tj) of the matrix is non-zero when some block pair it overlaps has terms; diagonal tiles always
//                 (noise, identity padding); the bordered rows (y - m, extra right-hand sides) are dense
//   factor      : symbolic factorisation, column by column -- tile (i, j) fills in when rows i and j share a non-zero
//                 tile in an earlier column
// Nothing is returned (dense) for one block, dense noise, or a pattern without zeros.  Also counts the k-block products of
// the contractions with and without the skipping (ctx->sz_executed / sz_dense: what the bench lines report).
// sz_pattern: the host part (ctx->h_sz, ctx->sz_executed / sz_dense); returns the words per row through *words, 0 = dense.
// grad_border (round 5): the matrix is the gradient path's [K ; (y - m)' ; I] (m_tot = 2 n_pad + 128: logpdf_grad_core) -- the
// identity rows get their own pattern (sz_symbolic: border_identity), and T_c more rows follow the T_r factor rows: row
// T_r + t = the tiles (t, .) of K itself that overlap a block pair with terms (+ the diagonal), i.e. the tiles of
// G = (alpha alpha' - C^-1) / 2 the contractions with the kernel derivatives read (TileSkip::need0).
static int sz_pattern(sgp_ctx* ctx, const sgp_dspec* ds, int noise_kind, long n_pad, long m_tot, int* words,
                      bool grad_border = false) {
  *words = 0;
  const long T_c = n_pad / TILE, T_r = m_tot / TILE;
  double dense = 0;
  for (long j = 0; j < T_c; ++j) dense += (double)j * (double)(T_r - j);
  if (grad_border) {
    // the dense count of the gradient's border: an identity row q only ever holds tiles at columns >= q, so column j sees
    // j - q of its k tiles (q < j) -- j (T_c - j) for the rows of K, j for the observation row, j (j + 1) / 2 for the identity rows
    if (T_r != 2 * T_c + 1) return 0;
    dense = 0;
    for (long j = 0; j < T_c; ++j) dense += (double)j * (double)(T_c - j) + (double)j + 0.5 * (double)j * (double)(j + 1);
  }
  ctx->sz_dense = ctx->sz_executed = dense;
  if (!ctx->struct_zeros || !ds->symmetric || noise_kind == SGP_NOISE_DENSE || ds->nrb < 2 || ds->nrb != ds->ncb) return 0;
  const int nb = ds->nrb;
  std::vector<char> bnz((size_t)nb * nb, 0);
  bool any_zero = false;
  for (int I = 0; I < nb; ++I)
    for (int J = 0; J < nb; ++J) {
      const int p = I * nb + J, q = J * nb + I;
      const bool has = ds->term_ptr[p + 1] > ds->term_ptr[p] || ds->term_ptr[q + 1] > ds->term_ptr[q];
      bnz[(size_t)I * nb + J] = has ? 1 : 0;
      if (!has && ds->row_len[I] > 0 && ds->row_len[J] > 0) any_zero = true;
    }
  if (!any_zero) return 0;
  SzPattern pat;   // sz_pattern.h: host-only, checked on the CPU by tests/sz_pattern_host.cpp
  sz_symbolic(bnz, nb, ds->row_off, ds->row_len, ds->N, TILE, T_c, T_r, pat, grad_border);
  if (grad_border) ctx->sz_dense = dense = pat.dense;
  ctx->sz_executed = pat.zeros_left ? pat.executed : dense;
  if (!pat.zeros_left) return 0;
  if (grad_border) {   // the needed result tiles: K's own (unfilled) tile pattern
    const int W = pat.words;
    std::vector<sz_pattern_word> need((size_t)T_c * W, 0);
    for (long i = 0; i < T_c; ++i) {
      const long p0 = i * TILE, p1 = std::min<long>(p0 + TILE, ds->N);
      for (long k = 0; k <= i; ++k) {
        const long q0 = k * TILE, q1 = std::min<long>(q0 + TILE, ds->N);
        bool on = i == k;
        for (int I = 0; I < nb && !on; ++I) {
          if (ds->row_len[I] <= 0 || !(ds->row_off[I] < p1 && ds->row_off[I] + ds->row_len[I] > p0)) continue;
          for (int J = 0; J < nb && !on; ++J)
            on = ds->row_len[J] > 0 && ds->row_off[J] < q1 && ds->row_off[J] + ds->row_len[J] > q0 && bnz[(size_t)I * nb + J] != 0;
        }
        if (on) need[(size_t)i * W + (k >> 6)] |= (sz_pattern_word)1 << (k & 63);
      }
    }
    pat.nz.insert(pat.nz.end(), need.begin(), need.end());
  }
  ctx->h_sz.swap(pat.nz);
  *words = pat.words;
  return 0;
}

// Per tile column of the factor of K + Sigma_y: the executed tile products of its contractions (SzPattern::col_work), from
// the HOST spec -- what multi.hip prices the column panels with before it deals them out to the ranks (own_table.h).  Empty:
// the model is structurally dense (one block, dense noise, no zero block pair, skipping switched off).
int sgp::drv_sz_col_work(const sgp_ctx* ctx, const sgp_cov_spec* sp, int noise_kind, long n_pad, long m_tot,
                         std::vector<double>& col_work) {
  col_work.clear();
  if (!ctx->struct_zeros || !sp || !sp->symmetric || noise_kind == SGP_NOISE_DENSE || sp->n_row_blocks < 2 ||
      sp->n_row_blocks != sp->n_col_blocks)
    return 0;
  const int nb = sp->n_row_blocks;
  std::vector<long> len((size_t)nb), off((size_t)nb);
  long N = 0;
  for (int I = 0; I < nb; ++I) {
    len[(size_t)I] = (long)sp->row_len[I];
    off[(size_t)I] = N;
    N += len[(size_t)I];
  }
  std::vector<char> bnz((size_t)nb * nb, 0);
  bool any_zero = false;
  for (int I = 0; I < nb; ++I)
    for (int J = 0; J < nb; ++J) {
      const int p = I * nb + J, q = J * nb + I;
      const bool has = sp->term_ptr[p + 1] > sp->term_ptr[p] || sp->term_ptr[q + 1] > sp->term_ptr[q];
      bnz[(size_t)I * nb + J] = has ? 1 : 0;
      if (!has && len[(size_t)I] > 0 && len[(size_t)J] > 0) any_zero = true;
    }
  if (!any_zero) return 0;
  SzPattern pat;
  sz_symbolic(bnz, nb, off, len, N, TILE, n_pad / TILE, m_tot / TILE, pat);
  if (pat.zeros_left) col_work.swap(pat.col_work);
  return 0;
}

// the device copy of a pattern (`h`: rows x words, possibly another context's -- the ranks of a multi-GPU context share one)
static int sz_upload(sgp_ctx* ctx, const std::vector<sz_word>& h, int words, hipStream_t s, SzMask* out) {
  *out = SzMask();
  if (words <= 0 || h.empty()) return 0;
  if (h.size() > ctx->n_sz) {
    SGP_HIP(hipStreamSynchronize(s));
    if (ctx->d_sz) hipFree(ctx->d_sz);
    ctx->d_sz = nullptr;
    ctx->n_sz = 0;
    SGP_HIP(hipMalloc(&ctx->d_sz, sizeof(sz_word) * h.size()));
    ctx->n_sz = h.size();
  }
  // `h` is pageable and the next call swaps / frees it (in a multi-GPU context every rank copies out of rank 0's): it is staged
  // through a pinned buffer of the context, so the copy is asynchronous AND safe (advisor, round 4: the pageable source; round
  // 5: the stream synchronisation that fixed it stalled the host enqueue of every structured logpdf / gradient call).  The
  // staging buffer is reused only after the previous copy out of it has completed (ev_sz: long done by then).
  const size_t bytes = sizeof(sz_word) * h.size();
  if (!ctx->ev_sz) SGP_HIP(hipEventCreateWithFlags(&ctx->ev_sz, hipEventDisableTiming));
  else SGP_HIP(hipEventSynchronize(ctx->ev_sz));
  if (bytes > ctx->n_sz_pin) {
    if (ctx->h_sz_pin) hipHostFree(ctx->h_sz_pin);
    ctx->h_sz_pin = nullptr;
    ctx->n_sz_pin = 0;
    SGP_HIP(hipHostMalloc((void**)&ctx->h_sz_pin, bytes + bytes / 2, hipHostMallocDefault));
    ctx->n_sz_pin = bytes + bytes / 2;
  }
  std::memcpy(ctx->h_sz_pin, h.data(), bytes);
  SGP_HIP(hipMemcpyAsync(ctx->d_sz, ctx->h_sz_pin, bytes, hipMemcpyHostToDevice, s));
  SGP_HIP(hipEventRecord(ctx->ev_sz, s));
  {   // room for the id map of the largest lower update this matrix can see
    const long rows = (long)(h.size() / (size_t)words);
    const long half = 16 + 8 * tri_ids_per_xcd(tri_shape(rows, std::min<long>(rows, (long)words * 64), -1));
    const long need = 2 * half;   // two maps: the look-ahead schedules launch structured updates on two streams
    if (need > ctx->n_szmap) {
      SGP_HIP(hipStreamSynchronize(s));
      if (ctx->d_szmap) hipFree(ctx->d_szmap);
      ctx->d_szmap = nullptr;
      ctx->n_szmap = 0;
      SGP_HIP(hipMalloc(&ctx->d_szmap, sizeof(int) * need));
      ctx->n_szmap = need;
    }
  }
  out->d_nz = ctx->d_sz;
  out->words = words;
  return 0;
}
static int sz_build(sgp_ctx* ctx, const sgp_dspec* ds, int noise_kind, long n_pad, long m_tot, hipStream_t s, SzMask* out) {
  *out = SzMask();
  int words = 0;
  CHECK_RC(sz_pattern(ctx, ds, noise_kind, n_pad, m_tot, &words));
  return sz_upload(ctx, ctx->h_sz, words, s, out);
}
static int build_bordered(sgp_ctx* ctx, const sgp_dspec* ds, double* dA, long n_pad, long m_tot,
                          const double* d_mean, int noise_kind, double sigma2,
                          const double* d_noise, const double* d_dense, long ld_dense,
                          const double* d_Y, long ldy, long ncols, hipStream_t s, SzMask* sz = nullptr) {
  long N = ds->N;
  if (sz) CHECK_RC(sz_build(ctx, ds, noise_kind, n_pad, m_tot, s, sz));
  int nk = noise_kind == SGP_NOISE_DENSE ? -1 : noise_kind;
  CHECK_RC(assemble(ds, dA, m_tot, 0, n_pad / TILE, 0, n_pad / TILE, 1, nk, sigma2, d_noise, s));
  if (noise_kind == SGP_NOISE_DENSE) CHECK_RC(launch_add_dense(dA, m_tot, d_dense, ld_dense, N, 1, s));
  CHECK_RC(launch_fill_pad(dA, m_tot, N, n_pad, 0, n_pad, m_tot, 0, s));
  CHECK_RC(launch_border_rows(dA, m_tot, n_pad, N, 0, n_pad, d_Y, ldy, ncols, d_mean, s));
  return 0;
}

static int fetch_info(sgp_ctx* ctx, hipStream_t s) {
  int info = 0;
  SGP_HIP(hipMemcpyAsync(&info, ctx->d_info, sizeof(int), hipMemcpyDeviceToHost, s));
  SGP_HIP(hipStreamSynchronize(s));
  if (info == SGP_DF_TIMEOUT) {
    ctx->df_timed_out = true;   // the entry point reruns the operator on the launch-based schedule (with_df_fallback)
    set_error("dataflow factorisation: a dependency wait inside the kernel ran into its bound (SGP_DF_TIMEOUT_S)");
  }
  return info;
}

static int dev_logpdf_impl(sgp_ctx* ctx, const sgp_dspec* ds, double* dA, const double* d_mean,
                           int noise_kind, double sigma2, const double* d_noise,
                           const double* d_dense, long ld_dense, const double* d_Y, long ldy,
                           long ncols, double* out_host, double* timings) {
  CHECK_ARG(ds->symmetric, "logpdf: spec must be symmetric");
  CHECK_ARG(ncols >= 1 && ncols + 16 <= ctx->n_scal, "logpdf: bad number of columns");
  hipStream_t s = ctx->stream;
  long N = ds->N;
  int64_t n_pad, m_tot;
  sgp_geometry(N, ncols, &n_pad, &m_tot);
  struct StageEvents {   // destroyed on every exit path
    hipEvent_t e[4] = {nullptr, nullptr, nullptr, nullptr};
    ~StageEvents() {
      for (auto x : e)
        if (x) hipEventDestroy(x);
    }
  } stage_events;
  hipEvent_t* ev = stage_events.e;
  if (timings)
    for (int q = 0; q < 4; ++q) SGP_HIP(hipEventCreate(&ev[q]));
  SGP_HIP(hipMemsetAsync(ctx->d_info, 0, sizeof(int), s));
  if (timings) SGP_HIP(hipEventRecord(ev[0], s));
  SzMask sz;
  CHECK_RC(build_bordered(ctx, ds, dA, n_pad, m_tot, d_mean, noise_kind, sigma2, d_noise, d_dense,
                          ld_dense, d_Y, ldy, ncols, s, &sz));
  if (timings) SGP_HIP(hipEventRecord(ev[1], s));
  ctx->time_updates = timings != nullptr;
  for (auto e : ctx->ev) hipEventDestroy(e);
  ctx->ev.clear();
  ctx->ev_flops.clear();
  int rc = chol_bordered(ctx, dA, m_tot, n_pad, m_tot, nullptr, s, 0, &sz);
  ctx->time_updates = false;
  if (rc) return rc;
  if (timings) SGP_HIP(hipEventRecord(ev[2], s));
  double* d_logdet = ctx->d_scal;
  double* d_sq = ctx->d_scal + 16;
  double* d_out = ctx->d_scal + 16 + ncols;
  CHECK_ARG(16 + 2 * ncols <= ctx->n_scal, "logpdf: too many columns");
  CHECK_RC(launch_rowsumsq(dA + n_pad, m_tot, N, ncols, d_sq, 0, s));
  CHECK_RC(launch_sum_array(ctx->d_slots, n_pad / TILE, d_logdet, s));
  CHECK_RC(launch_logpdf_final(d_logdet, d_sq, N, ncols, d_out, s));
  if (timings) SGP_HIP(hipEventRecord(ev[3], s));
  SGP_HIP(hipMemcpyAsync(out_host, d_out, sizeof(double) * ncols, hipMemcpyDeviceToHost, s));
  int info = fetch_info(ctx, s);
  if (timings) {
    float a = 0, b = 0, c = 0;
    SGP_HIP(hipEventElapsedTime(&a, ev[0], ev[1]));
    SGP_HIP(hipEventElapsedTime(&b, ev[1], ev[2]));
    SGP_HIP(hipEventElapsedTime(&c, ev[2], ev[3]));
    timings[0] = a;
    timings[1] = b;
    timings[2] = c;
    double ms_sum = 0, fl = 0;
    for (size_t i = 0; i + 1 < ctx->ev.size(); i += 2) {
      float ms = 0;
      SGP_HIP(hipEventElapsedTime(&ms, ctx->ev[i], ctx->ev[i + 1]));
      ms_sum += ms;
      fl += ctx->ev_flops[i / 2];
    }
    timings[3] = ms_sum;
    timings[4] = (double)(ctx->ev.size() / 2);
    timings[5] = fl;
    // the look-ahead runs update launches of two streams concurrently, so their durations overlap:
    // [6] = length of the union of the launch intervals (ms), the time the kernel was on the chip
    {
      std::vector<std::pair<float, float>> iv;
      for (size_t i = 0; i + 1 < ctx->ev.size(); i += 2) {
        float t0 = 0, t1 = 0;
        SGP_HIP(hipEventElapsedTime(&t0, ev[0], ctx->ev[i]));
        SGP_HIP(hipEventElapsedTime(&t1, ev[0], ctx->ev[i + 1]));
        iv.emplace_back(t0, t1);
      }
      std::sort(iv.begin(), iv.end());
      double uni = 0;
      if (!iv.empty()) {
        float cs = iv[0].first, ce = iv[0].second;
        for (size_t i = 1; i < iv.size(); ++i) {
          if (iv[i].first > ce) {
            uni += ce - cs;
            cs = iv[i].first;
            ce = iv[i].second;
          } else {
            ce = std::max(ce, iv[i].second);
          }
        }
        uni += ce - cs;
      }
      timings[6] = uni;
      timings[7] = 0.0;
    }
  }
  if (info < 0) return -3;
  if (info > 0) {
    set_error("matrix is not positive definite; Cholesky factorization failed at leading minor " +
              std::to_string(info));
    return info;
  }
  return 0;
}

// The dataflow kernel bounds every inter-workgroup wait by WALL-CLOCK time (SGP_DF_TIMEOUT_S): if the queue is preempted or
// time-sliced (another process on the GPU, a profiler serialising kernels) producers can be descheduled while the clock
// runs, and the kernel aborts although nothing is wrong (advisor, round 3).  An operator that comes back with that
// timeout is therefore run again, once, on the launch-based schedule -- every operator rebuilds its matrix from the spec,
// and every schedule gives the same bits, so the caller sees the result it would have seen, only later.
template <class F>
static int with_df_fallback(sgp_ctx* ctx, F&& run) {
  // the flag reset, the operator and its rerun are ONE critical section of the context (advisor, round 4: two host threads
  // on one context could erase each other's timeout flag, and the temporary dataflow = 0 leaked into the other's schedule)
  std::unique_lock<std::recursive_mutex> hold;
  if (ctx) hold = std::unique_lock<std::recursive_mutex>(ctx->mu);
  if (ctx) ctx->df_timed_out = false;
  int rc = run();
  if (ctx && rc == -3 && ctx->df_timed_out && ctx->df_fallback) {
    const int keep = ctx->dataflow, keep_h = ctx->hybrid;
    ctx->dataflow = 0;
    ctx->hybrid = 0;
    ctx->df_timed_out = false;
    ctx->df_fallbacks += 1;
    rc = run();
    ctx->dataflow = keep;
    ctx->hybrid = keep_h;
  }
  return rc;
}

static int sgp_dev_logpdf_impl(sgp_ctx* ctx, const sgp_dspec* ds, double* d_A, const double* d_mean,
                              int noise_kind, const double* noise_host, const double* d_noise,
                              const double* d_Y, int64_t ldy, int64_t ncols, double* out_host,
                              double* timings) {
  CHECK_ARG(ctx && ds && d_A && d_Y && out_host, "sgp_dev_logpdf: NULL argument");
  CHECK_ARG(noise_kind == SGP_NOISE_SCALAR || noise_kind == SGP_NOISE_DIAG,
            "sgp_dev_logpdf: noise kind must be SCALAR or DIAG");
  CtxScope scope(ctx);
  double s2 = noise_host ? noise_host[0] : 0.0;
  return dev_logpdf_impl(ctx, ds, d_A, d_mean, noise_kind, s2, d_noise, nullptr, 0, d_Y, ldy, ncols,
                         out_host, timings);
}
extern "C" int sgp_dev_logpdf(sgp_ctx* ctx, const sgp_dspec* ds, double* d_A, const double* d_mean,
                              int noise_kind, const double* noise_host, const double* d_noise,
                              const double* d_Y, int64_t ldy, int64_t ncols, double* out_host,
                              double* timings) {
  return with_df_fallback(ctx, [&]() { return sgp_dev_logpdf_impl(ctx, ds, d_A, d_mean, noise_kind, noise_host, d_noise, d_Y, ldy, ncols, out_host, timings); });
}

// ---------------------------------------------------------------------------------------
// host-buffer helpers
// ---------------------------------------------------------------------------------------
struct SpecGuard {
  sgp_dspec* ds = nullptr;
  ~SpecGuard() { dspec_free(ds); }
};

static int upload_matrix(DevBuf& b, const double* h, long ldh, long nr, long nc) {
  CHECK_RC(b.alloc((size_t)nr * nc));
  if (nr && nc &&
      hipMemcpy2D(b.p, sizeof(double) * nr, h, sizeof(double) * ldh, sizeof(double) * nr, (size_t)nc,
                  hipMemcpyHostToDevice) != hipSuccess) {
    set_error("hipMemcpy2D H2D failed");
    return -2;
  }
  return 0;
}

struct NoiseDev {
  int kind = SGP_NOISE_SCALAR;
  double sigma2 = 0.0;
  DevBuf diag, dense;
  long ld_dense = 0;
};
static int upload_noise(NoiseDev& nd, int kind, const double* noise, long N) {
  CHECK_ARG(noise != nullptr, "noise is NULL");
  CHECK_ARG(kind >= SGP_NOISE_SCALAR && kind <= SGP_NOISE_DENSE, "bad noise kind");
  nd.kind = kind;
  if (kind == SGP_NOISE_SCALAR) nd.sigma2 = noise[0];
  if (kind == SGP_NOISE_DIAG) CHECK_RC(nd.diag.upload(noise, N));
  if (kind == SGP_NOISE_DENSE) {
    CHECK_RC(nd.dense.upload(noise, (size_t)N * N));
    nd.ld_dense = N;
  }
  return 0;
}

extern "C" int sgp_kernelmatrix(sgp_ctx* ctx, const sgp_cov_spec* spec, double* K, int64_t ldk) {
  CHECK_ARG(ctx && spec && K, "sgp_kernelmatrix: NULL argument");
  CtxScope scope(ctx);
  if (ctx->multi) return sgp_multi_kernelmatrix(ctx, spec, K, ldk);
  SpecGuard g;
  CHECK_RC(dspec_create(ctx, spec, &g.ds));
  long N = g.ds->N, M = g.ds->M;
  CHECK_ARG(ldk >= N, "sgp_kernelmatrix: ldk < N");
  if (N == 0 || M == 0) return 0;
  DevBuf dK;
  CHECK_RC(dK.alloc((size_t)N * M));
  hipStream_t s = ctx->stream;
  CHECK_RC(assemble(g.ds, dK.p, N, 0, rup(N, TILE) / TILE, 0, rup(M, TILE) / TILE, g.ds->symmetric, -1, 0.0,
                    nullptr, s));
  if (g.ds->symmetric) CHECK_RC(launch_mirror_lower(dK.p, N, N, s));
  SGP_HIP(hipStreamSynchronize(s));
  SGP_HIP(hipMemcpy2D(K, sizeof(double) * ldk, dK.p, sizeof(double) * N, sizeof(double) * N, (size_t)M,
                      hipMemcpyDeviceToHost));
  return 0;
}

static int diag_of_spec(sgp_ctx* ctx, const sgp_dspec* ds, double* d_out, hipStream_t s) {
  CHECK_ARG(ds->nrb == ds->ncb, "kernelmatrix_diag: row / col block counts differ");
  for (int I = 0; I < ds->nrb; ++I) {
    CHECK_ARG(ds->row_len[I] == ds->col_len[I], "kernelmatrix_diag: block lengths differ");
    int p = I * ds->ncb + I;
    int t0 = ds->term_ptr[p], t1 = ds->term_ptr[p + 1];
    CHECK_RC(launch_diag_terms(d_out + ds->row_off[I], ds->row_len[I], ds->d_terms + t0, t1 - t0, s));
  }
  return 0;
}

extern "C" int sgp_kernelmatrix_diag(sgp_ctx* ctx, const sgp_cov_spec* spec, double* out) {
  CHECK_ARG(ctx && spec && out, "sgp_kernelmatrix_diag: NULL argument");
  CtxScope scope(ctx);
  if (ctx->multi) return sgp_multi_kernelmatrix_diag(ctx, spec, out);
  SpecGuard g;
  CHECK_RC(dspec_create(ctx, spec, &g.ds));
  long N = g.ds->N;
  if (N == 0) return 0;
  DevBuf d;
  CHECK_RC(d.alloc(N));
  CHECK_RC(diag_of_spec(ctx, g.ds, d.p, ctx->stream));
  SGP_HIP(hipStreamSynchronize(ctx->stream));
  SGP_HIP(hipMemcpy(out, d.p, sizeof(double) * N, hipMemcpyDeviceToHost));
  return 0;
}

static int sgp_logpdf_impl(sgp_ctx* ctx, const sgp_cov_spec* spec, const double* mean, int noise_kind,
                          const double* noise, const double* Y, int64_t ldy, int64_t ncols,
                          double* out) {
  CHECK_ARG(ctx && spec && Y && out, "sgp_logpdf: NULL argument");
  CHECK_ARG(spec->symmetric, "sgp_logpdf: spec must be symmetric");
  CtxScope scope(ctx);
  // a multi-GPU context (sgp_ctx_create_multi) shards the covariance over its devices
  // (dense Sigma_y is an N x N host matrix: that case stays on devices[0])
  if (ctx->multi && ncols >= 1 && noise)
    return sgp_multi_logpdf(ctx, spec, mean, noise_kind, noise, Y, ldy, ncols, out);
  SpecGuard g;
  CHECK_RC(dspec_create(ctx, spec, &g.ds));
  long N = g.ds->N;
  CHECK_ARG(N >= 1 && ncols >= 1 && ldy >= N, "sgp_logpdf: bad sizes");
  int64_t n_pad, m_tot;
  sgp_geometry(N, ncols, &n_pad, &m_tot);
  DevBuf dA, dmean, dY;
  NoiseDev nd;
  CHECK_RC(dA.alloc((size_t)m_tot * n_pad));
  if (mean) CHECK_RC(dmean.upload(mean, N));
  CHECK_RC(upload_noise(nd, noise_kind, noise, N));
  CHECK_RC(upload_matrix(dY, Y, ldy, N, ncols));
  return dev_logpdf_impl(ctx, g.ds, dA.p, mean ? dmean.p : nullptr, nd.kind, nd.sigma2, nd.diag.p,
                         nd.dense.p, nd.ld_dense, dY.p, N, ncols, out, nullptr);
}
extern "C" int sgp_logpdf(sgp_ctx* ctx, const sgp_cov_spec* spec, const double* mean, int noise_kind,
                          const double* noise, const double* Y, int64_t ldy, int64_t ncols,
                          double* out) {
  return with_df_fallback(ctx, [&]() { return sgp_logpdf_impl(ctx, spec, mean, noise_kind, noise, Y, ldy, ncols, out); });
}

// logpdf of nspec INDEPENDENT models in one call (round 6; sthenomi.h: sgp_logpdf_batch).  At the sizes where one
// factorisation is bound by its diagonal chain (N = 4096: 0.13 of the fp64 MFMA peak, ~220 CUs idle) the members of an equally
// sized batch are factored by ONE launch of the dataflow kernel as a single task pool (chol_df.hip: ids dealt round robin,
// progress counters per matrix): the B chains sit on different workgroups and hide each other.  Every member sees exactly
// the arithmetic of its own sgp_logpdf call (assembly, k-ascending contractions, the same reductions): the values are
// bit-equal.  "Equally sized" = the same PADDED size (the same number of 128-column tiles: the folds of a cross-validation,
// which differ by a point or two, pool).  Members of different padded sizes, dense noise, sizes outside the batched range, a
// multi-GPU context: one after the other through sgp_logpdf's own path.
static int logpdf_batch_impl(sgp_ctx* ctx, int nspec, const sgp_cov_spec* const* specs, const double* const* means,
                             int noise_kind, const double* const* noises, const double* const* ys, double* out, int* infos) {
  CHECK_ARG(ctx && specs && noises && ys && out && nspec >= 1, "sgp_logpdf_batch: NULL argument");
  for (int b = 0; b < nspec; ++b) {
    CHECK_ARG(specs[b] && noises[b] && ys[b], "sgp_logpdf_batch: NULL member");
    CHECK_ARG(specs[b]->symmetric, "sgp_logpdf_batch: specs must be symmetric");
    if (infos) infos[b] = 0;
  }
  auto rows_of = [](const sgp_cov_spec* sp) {
    long n = 0;
    for (int i = 0; i < sp->n_row_blocks; ++i) n += sp->row_len[i];
    return n;
  };
  // members pool when their PADDED geometry agrees (the folds of a cross-validation differ by a point or two: the same 128-column
  // tiles, each member's own N in its assembly, its row sums and its logpdf)
  const long N = rows_of(specs[0]);
  int64_t n_pad = 0, m_tot = 0;
  if (N >= 1) sgp_geometry(N, 1, &n_pad, &m_tot);
  bool same = N >= 1;
  for (int b = 1; b < nspec && same; ++b) {
    const long nb_ = rows_of(specs[b]);
    int64_t np_b = 0, mt_b = 0;
    if (nb_ >= 1) sgp_geometry(nb_, 1, &np_b, &mt_b);
    same = nb_ >= 1 && np_b == n_pad && mt_b == m_tot;
  }
  const bool pooled = same && N >= 1 && nspec >= 2 && !ctx->multi && ctx->dataflow != 0 &&
                      ctx->batch_max_n > 0 && n_pad <= ctx->batch_max_n &&
                      (noise_kind == SGP_NOISE_SCALAR || noise_kind == SGP_NOISE_DIAG);
  if (!pooled) {
    int first_bad = 0;
    for (int b = 0; b < nspec; ++b) {
      const int rc = sgp_logpdf(ctx, specs[b], means ? means[b] : nullptr, noise_kind, noises[b], ys[b], rows_of(specs[b]), 1, out + b);
      if (rc < 0) return rc;
      if (rc > 0) {
        out[b] = std::numeric_limits<double>::quiet_NaN();
        if (infos) infos[b] = rc;
        if (!first_bad) first_bad = rc;
      }
    }
    return infos ? 0 : first_bad;
  }
  CtxScope scope(ctx);
  hipStream_t s = ctx->stream;
  const long T_c = n_pad / TILE;
  const long per_small = T_c + 8;   // per member: logdet slots | [T_c] logdet | [T_c + 1] |L^-1 (y - m)|^2 | [T_c + 2] logpdf
  int first_bad = 0;
  for (int b0 = 0; b0 < nspec; b0 += DF_MAX_BATCH) {
    const int nb = std::min(DF_MAX_BATCH, nspec - b0);
    struct Member {
      SpecGuard g;
      DevBuf A, mean, y;
      NoiseDev nd;
    };
    std::vector<Member> mem((size_t)nb);
    DevBuf inv, small, infobuf;
    CHECK_RC(inv.alloc((size_t)nb * T_c * INVD_STRIDE));
    CHECK_RC(small.alloc((size_t)nb * per_small));
    CHECK_RC(infobuf.alloc((size_t)nb));   // (ints inside doubles' storage)
    int* d_infos = reinterpret_cast<int*>(infobuf.p);
    SGP_HIP(hipMemsetAsync(d_infos, 0, sizeof(int) * nb, s));
    CHECK_RC(df_scratch(ctx, m_tot, nb, 0, s));
    DfProb probs[DF_MAX_BATCH];
    for (int b = 0; b < nb; ++b) {
      Member& M = mem[(size_t)b];
      const int gb = b0 + b;
      const long Nb = rows_of(specs[gb]);   // this member's own size (n_pad, m_tot are the batch's)
      CHECK_RC(dspec_create(ctx, specs[gb], &M.g.ds));
      CHECK_RC(M.A.alloc((size_t)m_tot * n_pad));
      if (means && means[gb]) CHECK_RC(M.mean.upload(means[gb], Nb));
      CHECK_RC(upload_noise(M.nd, noise_kind, noises[gb], Nb));
      CHECK_RC(M.y.upload(ys[gb], Nb));
      // (no structural zeros inside a batch: its members need not share a pattern, and these sizes are chain-bound anyway)
      CHECK_RC(build_bordered(ctx, M.g.ds, M.A.p, n_pad, m_tot, M.mean.p, M.nd.kind, M.nd.sigma2, M.nd.diag.p, nullptr, 0, M.y.p,
                              Nb, 1, s, nullptr));
      probs[b] = DfProb{M.A.p, inv.p + (size_t)b * T_c * INVD_STRIDE, small.p + (size_t)b * per_small, d_infos + b};
    }
    CHECK_RC(launch_chol_dataflow_batch(probs, nb, m_tot, n_pad, m_tot, ctx->d_df_state, ctx->batch_fat ? ctx->hybrid_wgs : ctx->df_wgs,
                                        ctx->df_timeout_s, ctx->batch_fat, s));
    for (int b = 0; b < nb; ++b) {
      double* sm = small.p + (size_t)b * per_small;
      const long Nb = rows_of(specs[b0 + b]);
      CHECK_RC(launch_rowsumsq(mem[(size_t)b].A.p + n_pad, m_tot, Nb, 1, sm + T_c + 1, 0, s));
      CHECK_RC(launch_sum_array(sm, T_c, sm + T_c, s));
      CHECK_RC(launch_logpdf_final(sm + T_c, sm + T_c + 1, Nb, 1, sm + T_c + 2, s));
    }
    std::vector<double> h_small((size_t)nb * per_small);
    std::vector<int> h_info((size_t)nb);
    SGP_HIP(hipMemcpyAsync(h_small.data(), small.p, sizeof(double) * h_small.size(), hipMemcpyDeviceToHost, s));
    SGP_HIP(hipMemcpyAsync(h_info.data(), d_infos, sizeof(int) * nb, hipMemcpyDeviceToHost, s));
    SGP_HIP(hipStreamSynchronize(s));
    for (int b = 0; b < nb; ++b) {
      const int info = h_info[(size_t)b];
      if (info == SGP_DF_TIMEOUT) {
        ctx->df_timed_out = true;   // (with_df_fallback reruns the call member by member on the launch-based schedule)
        set_error("dataflow factorisation: a dependency wait inside the kernel ran into its bound (SGP_DF_TIMEOUT_S)");
        return -3;
      }
      out[b0 + b] = h_small[(size_t)b * per_small + T_c + 2];
      if (info > 0) {
        out[b0 + b] = std::numeric_limits<double>::quiet_NaN();
        if (infos) infos[b0 + b] = info;
        if (!first_bad) {
          first_bad = info;
          set_error("matrix is not positive definite; Cholesky factorization failed at leading minor " + std::to_string(info) +
                    " (batch member " + std::to_string(b0 + b) + ")");
        }
      }
    }
  }
  return infos ? 0 : first_bad;
}
extern "C" int sgp_logpdf_batch(sgp_ctx* ctx, int nspec, const sgp_cov_spec* const* specs, const double* const* means,
                                int noise_kind, const double* const* noises, const double* const* ys, double* out, int* infos) {
  return with_df_fallback(ctx, [&]() { return logpdf_batch_impl(ctx, nspec, specs, means, noise_kind, noises, ys, out, infos); });
}

// dst[i + c * ld] = mean[i] (i < N) else 0, for an nrows x ncols block
__global__ void fill_mean_cols_kernel(double* dst, long ld, long nrows, long ncols, long N, const double* mean) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= nrows * ncols) return;
  long i = idx % nrows, c = idx / nrows;
  dst[i + c * ld] = (mean && i < N) ? mean[i] : 0.0;
}

static int sgp_rand_impl(sgp_ctx* ctx, const sgp_cov_spec* spec, const double* mean, int noise_kind,
                        const double* noise, const double* Z, int64_t ldz, int64_t S, double* out,
                        int64_t ldo) {
  CHECK_ARG(ctx && spec && Z && out, "sgp_rand: NULL argument");
  CHECK_ARG(spec->symmetric, "sgp_rand: spec must be symmetric");
  CtxScope scope(ctx);
  if (ctx->multi && noise)
    return sgp_multi_rand(ctx, spec, mean, noise_kind, noise, Z, ldz, S, out, ldo);
  SpecGuard g;
  CHECK_RC(dspec_create(ctx, spec, &g.ds));
  long N = g.ds->N;
  CHECK_ARG(N >= 1 && S >= 1 && ldz >= N && ldo >= N, "sgp_rand: bad sizes");
  int64_t n_pad, m_tot;
  sgp_geometry(N, 0, &n_pad, &m_tot);
  long s_pad = rup(S, TILE);
  hipStream_t s = ctx->stream;
  DevBuf dA, dmean, dZ, dZt, dOut;
  NoiseDev nd;
  CHECK_RC(dA.alloc((size_t)m_tot * n_pad));
  if (mean) CHECK_RC(dmean.upload(mean, N));
  CHECK_RC(upload_noise(nd, noise_kind, noise, N));
  CHECK_RC(upload_matrix(dZ, Z, ldz, N, S));
  CHECK_RC(dZt.alloc((size_t)s_pad * n_pad));
  CHECK_RC(dOut.alloc((size_t)n_pad * s_pad));
  SGP_HIP(hipMemsetAsync(ctx->d_info, 0, sizeof(int), s));
  SzMask sz;
  CHECK_RC(build_bordered(ctx, g.ds, dA.p, n_pad, m_tot, nullptr, nd.kind, nd.sigma2, nd.diag.p,
                          nd.dense.p, nd.ld_dense, nullptr, 0, 0, s, &sz));
  CHECK_RC(chol_bordered(ctx, dA.p, m_tot, n_pad, m_tot, nullptr, s, 0, &sz));
  SGP_HIP(hipMemsetAsync(dZt.p, 0, sizeof(double) * s_pad * n_pad, s));
  // Zt[s, k] = Z[k, s]
  CHECK_RC(launch_transpose_add(dZ.p, N, N, S, dZt.p, s_pad, nullptr, s));
  // out[i, s] = mean[i] + sum_{k <= i} L[i, k] Zt[s, k]   (n_pad x s_pad, column-major like the result)
  hipLaunchKernelGGL(fill_mean_cols_kernel, dim3((unsigned)((n_pad * s_pad + 255) / 256)), dim3(256), 0, s, dOut.p,
                     (long)n_pad, (long)n_pad, s_pad, N, mean ? dmean.p : nullptr);
  SGP_HIP(hipGetLastError());
  CHECK_RC(launch_gemm_nt_lz(dA.p, m_tot, dZt.p, s_pad, dOut.p, n_pad, n_pad, s_pad, 1.0, s));
  int info = fetch_info(ctx, s);
  if (info < 0) return -3;
  if (info > 0) {
    set_error("matrix is not positive definite; Cholesky factorization failed at leading minor " +
              std::to_string(info));
    return info;
  }
  SGP_HIP(hipMemcpy2D(out, sizeof(double) * ldo, dOut.p, sizeof(double) * n_pad, sizeof(double) * N,
                      (size_t)S, hipMemcpyDeviceToHost));
  return 0;
}
extern "C" int sgp_rand(sgp_ctx* ctx, const sgp_cov_spec* spec, const double* mean, int noise_kind,
                        const double* noise, const double* Z, int64_t ldz, int64_t S, double* out,
                        int64_t ldo) {
  return with_df_fallback(ctx, [&]() { return sgp_rand_impl(ctx, spec, mean, noise_kind, noise, Z, ldz, S, out, ldo); });
}

// sum_ij G_ij dC_ij / d theta per flattened term of every block pair of `ds` (grad.hip):
// G = (alpha alpha' - Kinv) / 2 when alpha != nullptr, else G = the matrix at `Gm` itself.
static int contract_spec(const sgp_dspec* ds, const double* Gm, long ldg, const double* alpha, long n_tr,
                         long n_tc, DevBuf& dpart, double* dgc, double* dgs, hipStream_t s) {
  CHECK_RC(dpart.alloc((size_t)std::max<long>(1, n_tr * n_tc) * 16));
  for (int I = 0; I < ds->nrb; ++I) {
    if (ds->row_len[I] == 0) continue;
    for (int J = 0; J < ds->ncb; ++J) {
      if (ds->col_len[J] == 0) continue;
      long r0 = ds->row_off[I], nr = ds->row_len[I], c0 = ds->col_off[J], nc = ds->col_len[J];
      long trf = r0 / TILE, trl = (r0 + nr - 1) / TILE + 1, tcf = c0 / TILE, tcl = (c0 + nc - 1) / TILE + 1;
      int p = I * ds->ncb + J;
      int t0 = ds->term_ptr[p], t1 = ds->term_ptr[p + 1];
      int dmax = ds->pair_dmax[p];
      int per = std::min(8, std::max(1, 64 / dmax));
      for (int t = t0; t < t1; t += per) {
        int cnt = std::min(per, t1 - t);
        CHECK_RC(launch_grad_block(Gm, ldg, alpha, r0, nr, c0, nc, ds->d_terms + t, cnt, dmax, trf, tcf,
                                   trl - trf, tcl - tcf, dpart.p, dgc + t, dgs + t, s));
      }
    }
  }
  return 0;
}

// ---------------------------------------------------------------------------------------
// logpdf + reverse-mode gradient
// ---------------------------------------------------------------------------------------
// (every consumer of C^-1 -- the term contractions, the input-point and row-scale sums, the noise gradient -- walks the block
// pairs WITH terms or the diagonal: none needs a tile outside K's own tile pattern)
static int logpdf_grad_core(sgp_ctx* ctx, const sgp_cov_spec* spec, const double* mean, int noise_kind,
                            const double* noise, const double* y, double* logpdf_out, double* grad_y,
                            double* grad_mean, double* grad_noise, double* grad_coef, double* grad_inscale,
                            double* const* grad_inputs, double* const* grad_rowscale = nullptr) {
  CHECK_ARG(ctx && spec && noise && y && logpdf_out, "sgp_logpdf_grad: NULL argument");
  CHECK_ARG(spec->symmetric, "sgp_logpdf_grad: spec must be symmetric");
  CHECK_ARG(noise_kind >= SGP_NOISE_SCALAR && noise_kind <= SGP_NOISE_DENSE, "sgp_logpdf_grad: bad noise kind");
  CtxScope scope(ctx);
  // a multi-GPU context shards the gradient -- kernel terms, noise, y, the mean and (round 6) the input points and function
  // scales, a dense Sigma_y (multi.hip)
  if (ctx->multi)
    return sgp_multi_logpdf_grad(ctx, spec, mean, noise_kind, noise, y, logpdf_out, grad_y, grad_mean, grad_noise, grad_coef,
                                 grad_inscale, grad_inputs, grad_rowscale);
  SpecGuard g;
  CHECK_RC(dspec_create(ctx, spec, &g.ds));
  const sgp_dspec* ds = g.ds;
  long N = ds->N;
  CHECK_ARG(N >= 1, "sgp_logpdf_grad: empty data");
  const bool dense_noise = noise_kind == SGP_NOISE_DENSE;   // grad_noise is then N x N (ld = N): the cotangent G itself
  long n_pad = rup(N, TILE);
  long nrows = TILE + n_pad;            // the (y - m)' row (+ zero padding), then the identity rows
  long m_tot = n_pad + nrows;
  hipStream_t s = ctx->stream;
  DevBuf dA, dKinv, dmean, dy, dalpha, dpart, dgc, dgs, dgn;
  NoiseDev nd;
  CHECK_RC(dA.alloc((size_t)m_tot * n_pad));
  CHECK_RC(dKinv.alloc((size_t)n_pad * n_pad));
  if (mean) CHECK_RC(dmean.upload(mean, N));
  CHECK_RC(dy.upload(y, N));
  CHECK_RC(upload_noise(nd, noise_kind, noise, N));
  CHECK_RC(dalpha.alloc(n_pad));
  size_t nterms_total = ds->h_terms.size();
  CHECK_RC(dgc.alloc(std::max<size_t>(1, nterms_total)));
  CHECK_RC(dgs.alloc(std::max<size_t>(1, nterms_total)));
  CHECK_RC(dgn.alloc(dense_noise && grad_noise ? (size_t)N * N : (size_t)n_pad));
  SGP_HIP(hipMemsetAsync(ctx->d_info, 0, sizeof(int), s));
  SGP_HIP(hipMemsetAsync(dalpha.p, 0, sizeof(double) * n_pad, s));
  SGP_HIP(hipMemsetAsync(dgc.p, 0, sizeof(double) * std::max<size_t>(1, nterms_total), s));
  SGP_HIP(hipMemsetAsync(dgs.p, 0, sizeof(double) * std::max<size_t>(1, nterms_total), s));
  // K + Sigma_y, identity padding, bordered rows [(y - m)' ; I].  The identity rows come last so
  // that the rows a panel touches (K rows below it, the y row, identity rows above its last
  // column) are contiguous: the factorisation + inv(L) cost 2/3 N^3 instead of 4/3 N^3.
  CHECK_RC(assemble(ds, dA.p, m_tot, 0, n_pad / TILE, 0, n_pad / TILE, 1, dense_noise ? -1 : nd.kind, nd.sigma2,
                    nd.diag.p, s));
  if (dense_noise) CHECK_RC(launch_add_dense(dA.p, m_tot, nd.dense.p, nd.ld_dense, N, 1, s));
  CHECK_RC(launch_fill_pad(dA.p, m_tot, N, n_pad, 0, n_pad, m_tot, 0, s));
  CHECK_RC(launch_grad_border(dA.p, m_tot, n_pad, N, dy.p, mean ? dmean.p : nullptr, nrows, s));
  // Structural zeros (round 5): the factorisation of [K ; (y - m)' ; I] skips the tile products a programme with independent
  // components makes exact zeros -- in the factor AND in inv(L)', whose tile pattern is the closure of the factor's (the
  // identity rows run through the same symbolic elimination) -- C^-1 = inv(L)' inv(L) contracts only the k tiles both
  // operand tiles have, and only the tiles of C^-1 that a block pair WITH terms (or the diagonal: the noise gradient) reads
  // are computed at all.  Same bits as the dense schedule: every product left out is an exact zero or is never read.
  SzMask sz;
  {
    int words = 0;
    CHECK_RC(sz_pattern(ctx, ds, noise_kind, n_pad, m_tot, &words, true));
    CHECK_RC(sz_upload(ctx, ctx->h_sz, words, s, &sz));
  }
  CHECK_RC(chol_bordered(ctx, dA.p, m_tot, n_pad, m_tot, nullptr, s, n_pad + TILE, sz.d_nz ? &sz : nullptr));
  // row n_pad holds z' = (inv(L) (y - m))' ; rows n_pad + 128 .. now hold inv(L)' (upper triangular)
  const double* zrow = dA.p + n_pad;
  const double* Rinv = dA.p + n_pad + TILE;
  double* d_logdet = ctx->d_scal;
  double* d_sq = ctx->d_scal + 16;
  double* d_out = ctx->d_scal + 17;
  CHECK_RC(launch_rowsumsq(zrow, m_tot, N, 1, d_sq, 0, s));
  CHECK_RC(launch_sum_array(ctx->d_slots, n_pad / TILE, d_logdet, s));
  CHECK_RC(launch_logpdf_final(d_logdet, d_sq, N, 1, d_out, s));
  // alpha = inv(L)' z
  CHECK_RC(launch_gemv_rows(Rinv, m_tot, N, n_pad, zrow, m_tot, nullptr, dalpha.p, s, 1));
  // C^-1 = inv(L)' inv(L): lower tiles on the MFMA GEMM, then mirrored
  {
    TileSkip usk;
    if (sz.d_nz) {
      // the tiles outside K's own pattern are not computed: they must not be pool garbage for whoever reads all of C^-1
      // (today nobody does -- a dense Sigma_y has no pattern -- but the mirror below copies them; advisor, round 5)
      SGP_HIP(hipMemsetAsync(dKinv.p, 0, sizeof(double) * (size_t)n_pad * n_pad, s));
      usk.nz = sz.d_nz;
      usk.words = sz.words;
      usk.tr0 = usk.tc0 = (int)(n_pad / TILE + 1);   // the identity rows' pattern rows
      usk.need0 = (int)(m_tot / TILE);               // K's own tile pattern, behind the factor rows
    }
    CHECK_RC(launch_gemm_nt_uut(Rinv, m_tot, dKinv.p, n_pad, n_pad, s, usk.nz ? &usk : nullptr));
  }
  CHECK_RC(launch_mirror_lower(dKinv.p, n_pad, n_pad, s));
  if (grad_noise && dense_noise) CHECK_RC(launch_grad_noise_dense(dKinv.p, n_pad, dalpha.p, N, dgn.p, s));
  else if (grad_noise) CHECK_RC(launch_grad_noise(dKinv.p, n_pad, dalpha.p, N, nd.kind == SGP_NOISE_DIAG, dgn.p, s));
  if (grad_coef || grad_inscale)
    CHECK_RC(contract_spec(ds, dKinv.p, n_pad, dalpha.p, n_pad / TILE, n_pad / TILE, dpart, dgc.p, dgs.p, s));
  // gradient w.r.t. the input points: row-side contraction over every block pair; the spec is
  // symmetric (block (J, I) mirrors (I, J)) and so is G, hence the column side equals the row side of
  // the mirror block and the total is twice the row-side sum
  // The row-scale vectors of function-scaled processes (K_ij = coef rs_i k_ij cs_j) ride along: the same
  // row-side sums with k in place of its derivative; the column scale of a term is the row scale of its mirror
  // term, so "row side x 2" covers both roles as it does for the points.
  std::vector<DevBuf> dgx(grad_inputs ? spec->n_inputs : 0);
  std::vector<DevBuf> dgr(grad_rowscale ? nterms_total : 0);
  if (grad_inputs || grad_rowscale) {
    for (int k = 0; grad_inputs && k < spec->n_inputs; ++k) {
      size_t cnt = (size_t)std::max<long>(1, (long)ds->in_dim[k] * ds->in_n[k]);
      CHECK_RC(dgx[k].alloc(cnt));
      SGP_HIP(hipMemsetAsync(dgx[k].p, 0, sizeof(double) * cnt, s));
    }
    for (int I = 0; I < ds->nrb; ++I) {
      for (int J = 0; J < ds->ncb; ++J) {
        if (ds->row_len[I] == 0 || ds->col_len[J] == 0) continue;
        int p = I * ds->ncb + J;
        for (int t = ds->term_ptr[p]; t < ds->term_ptr[p + 1]; ++t) {
          int a = ds->term_row_input[t];
          double* gsv = nullptr;
          if (grad_rowscale && grad_rowscale[t] && ds->h_terms[t].rs) {
            CHECK_RC(dgr[t].alloc((size_t)ds->row_len[I]));
            SGP_HIP(hipMemsetAsync(dgr[t].p, 0, sizeof(double) * ds->row_len[I], s));
            gsv = dgr[t].p;
          }
          if (!grad_inputs && !gsv) continue;
          CHECK_RC(launch_grad_inputs(dKinv.p, 1, n_pad, dalpha.p, ds->row_off[I], ds->row_len[I], ds->col_off[J],
                                      ds->col_len[J], ds->h_terms[t], ds->pair_dmax[p], 2.0,
                                      grad_inputs ? dgx[a].p : nullptr, s, gsv));
        }
      }
    }
  }
  int info = fetch_info(ctx, s);
  if (info < 0) return -3;
  if (info > 0) {
    set_error("matrix is not positive definite; Cholesky factorization failed at leading minor " +
              std::to_string(info));
    return info;
  }
  SGP_HIP(hipMemcpy(logpdf_out, d_out, sizeof(double), hipMemcpyDeviceToHost));
  std::vector<double> ha(N);
  SGP_HIP(hipMemcpy(ha.data(), dalpha.p, sizeof(double) * N, hipMemcpyDeviceToHost));
  if (grad_y)
    for (long i = 0; i < N; ++i) grad_y[i] = -ha[i];
  if (grad_mean)
    for (long i = 0; i < N; ++i) grad_mean[i] = ha[i];
  if (grad_noise)
    SGP_HIP(hipMemcpy(grad_noise, dgn.p,
                      sizeof(double) * (dense_noise ? (size_t)N * N : (nd.kind == SGP_NOISE_DIAG ? (size_t)N : (size_t)1)),
                      hipMemcpyDeviceToHost));
  if (grad_coef && nterms_total)
    SGP_HIP(hipMemcpy(grad_coef, dgc.p, sizeof(double) * nterms_total, hipMemcpyDeviceToHost));
  if (grad_inscale && nterms_total)
    SGP_HIP(hipMemcpy(grad_inscale, dgs.p, sizeof(double) * nterms_total, hipMemcpyDeviceToHost));
  if (grad_inputs) {
    for (int k = 0; k < spec->n_inputs; ++k) {
      const sgp_input& in = spec->inputs[k];
      if (!grad_inputs[k] || in.n == 0) continue;
      SGP_HIP(hipMemcpy(grad_inputs[k], dgx[k].p, sizeof(double) * in.dim * in.n, hipMemcpyDeviceToHost));
    }
  }
  if (grad_rowscale) {
    for (int I = 0; I < ds->nrb; ++I)
      for (int J = 0; J < ds->ncb; ++J)
        for (int t = ds->term_ptr[I * ds->ncb + J]; t < ds->term_ptr[I * ds->ncb + J + 1]; ++t)
          if (dgr[t].p && ds->row_len[I] > 0)
            SGP_HIP(hipMemcpy(grad_rowscale[t], dgr[t].p, sizeof(double) * ds->row_len[I], hipMemcpyDeviceToHost));
  }
  return 0;
}

extern "C" int sgp_logpdf_grad_xs(sgp_ctx* ctx, const sgp_cov_spec* spec, const double* mean, int noise_kind,
                                  const double* noise, const double* y, double* logpdf_out, double* grad_y,
                                  double* grad_mean, double* grad_noise, double* grad_coef, double* grad_inscale,
                                  double* const* grad_inputs, double* const* grad_rowscale) {
  return with_df_fallback(ctx, [&]() {
    return logpdf_grad_core(ctx, spec, mean, noise_kind, noise, y, logpdf_out, grad_y, grad_mean, grad_noise,
                            grad_coef, grad_inscale, grad_inputs, grad_rowscale);
  });
}

extern "C" int sgp_logpdf_grad(sgp_ctx* ctx, const sgp_cov_spec* spec, const double* mean, int noise_kind,
                               const double* noise, const double* y, double* logpdf_out, double* grad_y,
                               double* grad_mean, double* grad_noise, double* grad_coef,
                               double* grad_inscale) {
  return with_df_fallback(ctx, [&]() {
    return logpdf_grad_core(ctx, spec, mean, noise_kind, noise, y, logpdf_out, grad_y, grad_mean, grad_noise,
                            grad_coef, grad_inscale, nullptr);
  });
}

extern "C" int sgp_logpdf_grad_x(sgp_ctx* ctx, const sgp_cov_spec* spec, const double* mean, int noise_kind,
                                 const double* noise, const double* y, double* logpdf_out, double* grad_y,
                                 double* grad_mean, double* grad_noise, double* grad_coef,
                                 double* grad_inscale, double* const* grad_inputs) {
  CHECK_ARG(grad_inputs != nullptr, "sgp_logpdf_grad_x: grad_inputs is NULL");
  return with_df_fallback(ctx, [&]() {
    return logpdf_grad_core(ctx, spec, mean, noise_kind, noise, y, logpdf_out, grad_y, grad_mean, grad_noise,
                            grad_coef, grad_inscale, grad_inputs);
  });
}

// ---------------------------------------------------------------------------------------
// posterior
// ---------------------------------------------------------------------------------------
struct sgp_post {
  sgp_ctx* ctx = nullptr;
  long ctx_serial = 0;
  long N = 0, n_pad = 0, m_tot = 0;
  double* dA = nullptr;     // L (lower tiles) + row n_pad = (L^-1 (y - m))'
  double* d_wall = nullptr; // inverse 16x16 diagonal blocks (INVD_STRIDE per 128-block)
  sgp_mpost* mp = nullptr;  // non-null: the factor is sharded over the ranks of a multi-GPU context (multi.hip)
};

// rows <- rows * L^-T for `nrows` (multiple of 128) rows stored at R (ld = ldr), against the factor L
// (ld = ldl) with its stored inverse 16x16 diagonal blocks.  Blocked left-looking: a 512-column
// block first receives everything left of it in ONE deep GEMM (K = its column offset: the efficient
// shape of the MFMA kernel), then is solved in place in 128-column steps with K = 128 updates inside
// the block only.  Same flops as the right-looking sweep, a fraction of its C-tile traffic.
// the columns [c, c + w) of a block, everything left of the block already applied: by halving -- the left half, ONE update
// of the right half with it (K = the left half's width), the right half.  The same ascending-k accumulation per entry as
// 128-column steps with K = 128 updates of everything to their right (each launch seeds its accumulators with the stored
// value), at 2 / 3 of their C-tile traffic in a 512-column block.
static int row_trsm_block(double* R, long ldr, long nrows, const double* L, long ldl, const double* d_invall, long c, long w,
                          long solve_div, hipStream_t s) {
  if (w <= TILE)
    return launch_panel_solve(R + c * ldr, ldr, nrows, L + c + c * ldl, ldl, d_invall + (c / TILE) * INVD_STRIDE, 256, 16, s,
                              nullptr, solve_div);
  const long wl = (w / TILE + 1) / 2 * TILE;
  CHECK_RC(row_trsm_block(R, ldr, nrows, L, ldl, d_invall, c, wl, solve_div, s));
  CHECK_RC(launch_gemm_nt(R + c * ldr, ldr, L + (c + wl) + c * ldl, ldl, R + (c + wl) * ldr, ldr, nrows, w - wl, wl, -1.0, 1.0,
                          NOMASK, 0, 0, s));
  return row_trsm_block(R, ldr, nrows, L, ldl, d_invall, c + wl, w - wl, solve_div, s);
}
static int row_trsm(sgp_ctx* ctx, double* R, long ldr, long nrows, const double* L, long ldl,
                    const double* d_invall, long n_pad, hipStream_t s) {
  // (block width measured round 3 on the N = 262144, M = 4096 ELBO: 128 / 256 / 512 / 1024 columns -> row solve
  // 80.3 / 78.7 / 78.6 / 80.6 ms: flat -- the in-block part is not what bounds it; round 6, with the halving inside the block
  // and two solve workgroups per CU: 256 / 512 / 1024 / 2048 columns -> step 152.2 / 150.7 / 150.7 / 150.4 ms, against 154.2
  // for the 128-column steps with K = 128 updates -- profiles/r06_experiments/elbo_c4.md)
  const long WB = 4 * TILE;
  for (long c0 = 0; c0 < n_pad; c0 += WB) {
    long wb = std::min(WB, n_pad - c0);
    // (with a leading dimension beyond ~64k rows every operand column sits in its own page and a deep
    // contraction thrashes the TLB: fall back to 512-column slices there)
    const long KC = (ldr > 65536 || ldl > 65536) ? WB : c0;
    for (long k0 = 0; k0 < c0; k0 += KC)
      CHECK_RC(launch_gemm_nt(R + k0 * ldr, ldr, L + c0 + k0 * ldl, ldl, R + c0 * ldr, ldr, nrows, wb,
                              std::min(KC, c0 - k0), -1.0, 1.0, NOMASK, 0, 0, s));
    CHECK_RC(row_trsm_block(R, ldr, nrows, L, ldl, d_invall, c0, wb, 512, s));
  }
  return 0;
}

// back substitution kernels for alpha = L^-T z (reduce.hip would do; kept here: tiny)
__global__ void backsolve_gemvt_kernel(const double* L, long ld, long k0, long n_pad,
                                       const double* alpha, double* z) {
  // z[k0 + j] -= sum_{r >= k0+128} L[r, k0+j] * alpha[r], one workgroup per j
  __shared__ double sh[4];
  const int j = blockIdx.x;
  const double* col = L + (long)(k0 + j) * ld;
  double acc = 0.0;
  for (long r = k0 + TILE + threadIdx.x; r < n_pad; r += blockDim.x) acc = fma(col[r], alpha[r], acc);
  for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) z[k0 + j] -= (sh[0] + sh[1]) + (sh[2] + sh[3]);
}
// alpha_k = L_kk^-T z_k for one 128-block: blocked back substitution over its eight 16x16
// sub-blocks, each diagonal solve an inverse product + one refinement step (as panel_solve_kernel).
// Thread (m, q): element m of the current sub-block, q = contraction lane (16 aligned lanes).
__global__ __launch_bounds__(256) void backsolve_diag_kernel(const double* invd, const double* Lkk, long ld,
                                                            const double* z, double* alpha, long k0) {
  __shared__ double zs[TILE], as[TILE], ts[16], a1s[16], rs[16];
  const int t = threadIdx.x, m = t >> 4, q = t & 15;
  if (t < TILE) zs[t] = z[k0 + t];
  __syncthreads();
  for (int c = 7; c >= 0; --c) {
    double acc = 0.0;  // sum_{p > c} L[16p + q][16c + m] a[16p + q]
    for (int p = c + 1; p < 8; ++p) acc = fma(Lkk[(16 * p + q) + (long)(16 * c + m) * ld], as[16 * p + q], acc);
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (q == 0) ts[m] = zs[16 * c + m] - acc;
    __syncthreads();
    const double iv = invd[c * 256 + m * 16 + q];  // inv(L_cc)[q][m]
    double v = iv * ts[q];
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    if (q == 0) a1s[m] = v;
    __syncthreads();
    double w = (q >= m) ? Lkk[(16 * c + q) + (long)(16 * c + m) * ld] * a1s[q] : 0.0;
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) w += __shfl_xor(w, off, 64);
    if (q == 0) rs[m] = ts[m] - w;  // t - L_cc' a1
    __syncthreads();
    double d = iv * rs[q];
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) d += __shfl_xor(d, off, 64);
    if (q == 0) as[16 * c + m] = a1s[m] + d;
    __syncthreads();
  }
  if (t < TILE) alpha[k0 + t] = as[t];
}

static int back_substitute_range(const double* Lv, long ld, const double* wall_v, long k_first, long k_last,
                                 long n_end, double* d_z, double* d_alpha, hipStream_t s) {
  for (long k0 = k_first; k0 >= k_last; k0 -= TILE) {
    if (k0 + TILE < n_end) {
      hipLaunchKernelGGL(backsolve_gemvt_kernel, dim3(TILE), dim3(256), 0, s, Lv, ld, k0, n_end, d_alpha, d_z);
      SGP_HIP(hipGetLastError());
    }
    hipLaunchKernelGGL(backsolve_diag_kernel, dim3(1), dim3(256), 0, s, wall_v + (k0 / TILE) * INVD_STRIDE,
                       Lv + k0 + k0 * ld, ld, d_z, d_alpha, k0);
    SGP_HIP(hipGetLastError());
  }
  return 0;
}
static int back_substitute(const sgp_post* post, double* d_z /*n_pad, overwritten*/,
                           double* d_alpha, hipStream_t s) {
  return back_substitute_range(post->dA, post->m_tot, post->d_wall, post->n_pad - TILE, 0, post->n_pad, d_z, d_alpha,
                               s);
}

__global__ void copy_strided_kernel(const double* src, long lds, long n, double* dst) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[i * lds];
}

extern "C" int sgp_posterior_destroy(sgp_post* p) {
  if (!p) return 0;
  if (p->mp) sgp_multi_posterior_destroy(p->mp);
  if (p->dA) hipFree(p->dA);
  if (p->d_wall) hipFree(p->d_wall);
  delete p;
  return 0;
}

static int sgp_posterior_create_impl(sgp_ctx* ctx, const sgp_cov_spec* spec, const double* mean,
                                    int noise_kind, const double* noise, const double* y,
                                    double* alpha_out, sgp_post** out) {
  CHECK_ARG(ctx && spec && y && out, "sgp_posterior_create: NULL argument");
  CHECK_ARG(spec->symmetric, "sgp_posterior_create: spec must be symmetric");
  CtxScope scope(ctx);
  if (ctx->multi && noise) {
    sgp_mpost* mp = nullptr;
    CHECK_RC(sgp_multi_posterior_create(ctx, spec, mean, noise_kind, noise, y, alpha_out, &mp));
    sgp_post* post = new sgp_post();
    post->ctx = ctx;
    post->ctx_serial = ctx->serial;
    post->mp = mp;
    *out = post;
    return 0;
  }
  SpecGuard g;
  CHECK_RC(dspec_create(ctx, spec, &g.ds));
  long N = g.ds->N;
  CHECK_ARG(N >= 1, "sgp_posterior_create: empty data");
  int64_t n_pad, m_tot;
  sgp_geometry(N, 1, &n_pad, &m_tot);
  hipStream_t s = ctx->stream;
  DevBuf dmean, dY;
  NoiseDev nd;
  if (mean) CHECK_RC(dmean.upload(mean, N));
  CHECK_RC(upload_noise(nd, noise_kind, noise, N));
  CHECK_RC(dY.upload(y, N));
  sgp_post* post = new sgp_post();
  post->ctx = ctx;
  post->ctx_serial = ctx->serial;
  post->N = N;
  post->n_pad = n_pad;
  post->m_tot = m_tot;
  struct PostGuard {  // every error exit (incl. the SGP_HIP early returns) frees the 8 N^2-byte factor
    sgp_post* p;
    ~PostGuard() {
      if (p) sgp_posterior_destroy(p);
    }
  } guard{post};
  auto fail = [&](int rc) { return rc; };
  if (hipMalloc(&post->dA, sizeof(double) * m_tot * n_pad) != hipSuccess ||
      hipMalloc(&post->d_wall, sizeof(double) * (n_pad / TILE) * INVD_STRIDE) != hipSuccess) {
    set_error("sgp_posterior_create: hipMalloc failed");
    return fail(-2);
  }
  SGP_HIP(hipMemsetAsync(ctx->d_info, 0, sizeof(int), s));
  SzMask sz;
  int rc = build_bordered(ctx, g.ds, post->dA, n_pad, m_tot, mean ? dmean.p : nullptr, nd.kind,
                          nd.sigma2, nd.diag.p, nd.dense.p, nd.ld_dense, dY.p, N, 1, s, &sz);
  if (rc) return fail(rc);
  rc = chol_bordered(ctx, post->dA, m_tot, n_pad, m_tot, post->d_wall, s, 0, &sz);
  if (rc) return fail(rc);
  if (alpha_out) {
    DevBuf dz, dal;
    if (dz.alloc(n_pad) || dal.alloc(n_pad)) return fail(-2);
    hipLaunchKernelGGL(copy_strided_kernel, dim3((unsigned)((n_pad + 255) / 256)), dim3(256), 0, s,
                       post->dA + n_pad, m_tot, n_pad, dz.p);
    rc = back_substitute(post, dz.p, dal.p, s);
    if (rc) return fail(rc);
    SGP_HIP(hipStreamSynchronize(s));
    SGP_HIP(hipMemcpy(alpha_out, dal.p, sizeof(double) * N, hipMemcpyDeviceToHost));
  }
  int info = fetch_info(ctx, s);
  if (info < 0) return -3;
  if (info > 0) {
    set_error("matrix is not positive definite; Cholesky factorization failed at leading minor " +
              std::to_string(info));
    return fail(info);
  }
  guard.p = nullptr;
  *out = post;
  return 0;
}
extern "C" int sgp_posterior_create(sgp_ctx* ctx, const sgp_cov_spec* spec, const double* mean,
                                    int noise_kind, const double* noise, const double* y,
                                    double* alpha_out, sgp_post** out) {
  return with_df_fallback(ctx, [&]() { return sgp_posterior_create_impl(ctx, spec, mean, noise_kind, noise, y, alpha_out, out); });
}

// shared by dense and sparse prediction: V rows (x* bordered rows, ns_pad x n_pad)
static int predict_common(sgp_ctx* ctx, const sgp_dspec* cross, const sgp_dspec* prior,
                          const double* d_means, long Ns, long ns_pad, double* dV, long n_pad,
                          const double* d_z, long ldz, long N_cols, double* mean_out,
                          double* var_out, double* cov_out, int64_t ldcov, double var_sign_second,
                          const double* dV2, hipStream_t s) {
  DevBuf dmu, dprior, dvar, dcov;
  if (mean_out) {
    CHECK_RC(dmu.alloc(Ns));
    CHECK_RC(launch_gemv_rows(dV, ns_pad, Ns, N_cols, d_z, ldz, d_means, dmu.p, s));
  }
  if (var_out) {
    CHECK_ARG(prior != nullptr, "predict: prior_ss spec required for var");
    CHECK_RC(dprior.alloc(Ns));
    CHECK_RC(dvar.alloc(Ns));
    CHECK_RC(diag_of_spec(ctx, prior, dprior.p, s));
    CHECK_RC(launch_colsumsq_sub(dV, ns_pad, Ns, N_cols, dprior.p, dvar.p, -1.0, s));
    if (dV2) CHECK_RC(launch_colsumsq_sub(dV2, ns_pad, Ns, N_cols, dvar.p, dvar.p, var_sign_second, s));
  }
  if (cov_out) {
    CHECK_ARG(prior != nullptr, "predict: prior_ss spec required for cov");
    CHECK_RC(dcov.alloc((size_t)ns_pad * ns_pad));
    SGP_HIP(hipMemsetAsync(dcov.p, 0, sizeof(double) * ns_pad * ns_pad, s));
    CHECK_RC(assemble(prior, dcov.p, ns_pad, 0, ns_pad / TILE, 0, ns_pad / TILE, 0, -1, 0.0, nullptr, s));
    CHECK_RC(launch_gemm_nt(dV, ns_pad, dV, ns_pad, dcov.p, ns_pad, ns_pad, ns_pad, n_pad, -1.0, 1.0,
                            NOMASK, 0, 0, s));
    if (dV2)
      CHECK_RC(launch_gemm_nt(dV2, ns_pad, dV2, ns_pad, dcov.p, ns_pad, ns_pad, ns_pad, n_pad,
                              var_sign_second, 1.0, NOMASK, 0, 0, s));
  }
  SGP_HIP(hipStreamSynchronize(s));
  if (mean_out) SGP_HIP(hipMemcpy(mean_out, dmu.p, sizeof(double) * Ns, hipMemcpyDeviceToHost));
  if (var_out) SGP_HIP(hipMemcpy(var_out, dvar.p, sizeof(double) * Ns, hipMemcpyDeviceToHost));
  if (cov_out)
    SGP_HIP(hipMemcpy2D(cov_out, sizeof(double) * ldcov, dcov.p, sizeof(double) * ns_pad,
                        sizeof(double) * Ns, (size_t)Ns, hipMemcpyDeviceToHost));
  return 0;
}

extern "C" int sgp_posterior_predict(sgp_post* post, const sgp_cov_spec* cross,
                                     const sgp_cov_spec* prior_ss, const double* mean_s,
                                     double* mean_out, double* var_out, double* cov_out,
                                     int64_t ldcov) {
  CHECK_ARG(post && cross, "sgp_posterior_predict: NULL argument");
  sgp_ctx* ctx = post->ctx;
  CHECK_ARG(ctx_is_live(ctx, post->ctx_serial),
            "sgp_posterior_predict: the context this posterior was created on has been destroyed");
  CtxScope scope(ctx);
  if (post->mp) return sgp_multi_posterior_predict(post->mp, cross, prior_ss, mean_s, mean_out, var_out, cov_out, ldcov);
  SpecGuard gc, gp;
  CHECK_RC(dspec_create(ctx, cross, &gc.ds));
  if (prior_ss) CHECK_RC(dspec_create(ctx, prior_ss, &gp.ds));
  long Ns = gc.ds->N;
  CHECK_ARG(gc.ds->M == post->N, "sgp_posterior_predict: cross spec columns != training size");
  CHECK_ARG(!gp.ds || gp.ds->N == Ns, "sgp_posterior_predict: prior_ss size != number of x*");
  CHECK_ARG(!cov_out || ldcov >= Ns, "sgp_posterior_predict: ldcov < Ns");
  if (Ns == 0) return 0;
  long ns_pad = rup(Ns, TILE), n_pad = post->n_pad;
  hipStream_t s = ctx->stream;
  DevBuf dV, dms;
  CHECK_RC(dV.alloc((size_t)ns_pad * n_pad));
  if (mean_s) CHECK_RC(dms.upload(mean_s, Ns));
  SGP_HIP(hipMemsetAsync(dV.p, 0, sizeof(double) * ns_pad * n_pad, s));
  CHECK_RC(assemble(gc.ds, dV.p, ns_pad, 0, ns_pad / TILE, 0, n_pad / TILE, 0, -1, 0.0, nullptr, s));
  CHECK_RC(row_trsm(ctx, dV.p, ns_pad, ns_pad, post->dA, post->m_tot, post->d_wall, n_pad, s));
  return predict_common(ctx, gc.ds, gp.ds, mean_s ? dms.p : nullptr, Ns, ns_pad, dV.p, n_pad,
                        post->dA + n_pad, post->m_tot, post->N, mean_out, var_out, cov_out, ldcov,
                        0.0, nullptr, s);
}

// Predictions against a kept factor with the cross-covariance GIVEN as a matrix (round 5): what conditioning on top of a process
// that is not a prior Stheno process needs -- the approximate (VFE) posterior is an ordinary AbstractGP in the reference
// (src/gp/sparse_finite_gp.jl:60-62: posterior(VFE(f(z)), fx, y) can be observed and conditioned again), but its covariance
// is not a sum of kernel terms, so the flattened spec cannot say it.  The host evaluates cov(f, x*, x), var / cov(f, x*) and
// mean(f, x*) with the operators it has (on the device) and hands them over; the factor was built by sgp_posterior_create on
// the zero-term spec with dense noise = cov(f, x) + Sigma_y.
extern "C" int sgp_posterior_predict_explicit(sgp_post* post, const double* cross, int64_t ldc, int64_t ns,
                                              const double* prior_var, const double* prior_cov, int64_t ldp,
                                              const double* mean_s, double* mean_out, double* var_out, double* cov_out,
                                              int64_t ldcov) {
  CHECK_ARG(post && cross, "sgp_posterior_predict_explicit: NULL argument");
  sgp_ctx* ctx = post->ctx;
  CHECK_ARG(ctx_is_live(ctx, post->ctx_serial),
            "sgp_posterior_predict_explicit: the context this posterior was created on has been destroyed");
  CHECK_ARG(!post->mp, "sgp_posterior_predict_explicit: not available on a multi-GPU context's sharded factor");
  const long Ns = ns, N = post->N;
  CHECK_ARG(Ns >= 0 && ldc >= std::max<long>(Ns, 1), "sgp_posterior_predict_explicit: bad sizes");
  CHECK_ARG(!var_out || prior_var, "sgp_posterior_predict_explicit: prior_var required for var");
  CHECK_ARG(!cov_out || (prior_cov && ldp >= Ns && ldcov >= Ns), "sgp_posterior_predict_explicit: prior_cov required for cov");
  if (Ns == 0) return 0;
  CtxScope scope(ctx);
  const long ns_pad = rup(Ns, TILE), n_pad = post->n_pad;
  hipStream_t s = ctx->stream;
  DevBuf dV, dms, dmu, dprior, dvar, dcov;
  CHECK_RC(dV.alloc((size_t)ns_pad * n_pad));
  if (mean_s) CHECK_RC(dms.upload(mean_s, Ns));
  SGP_HIP(hipMemsetAsync(dV.p, 0, sizeof(double) * ns_pad * n_pad, s));
  SGP_HIP(hipMemcpy2DAsync(dV.p, sizeof(double) * ns_pad, cross, sizeof(double) * ldc, sizeof(double) * Ns, (size_t)N,
                           hipMemcpyHostToDevice, s));
  CHECK_RC(row_trsm(ctx, dV.p, ns_pad, ns_pad, post->dA, post->m_tot, post->d_wall, n_pad, s));
  if (mean_out) {
    CHECK_RC(dmu.alloc(Ns));
    CHECK_RC(launch_gemv_rows(dV.p, ns_pad, Ns, N, post->dA + n_pad, post->m_tot, mean_s ? dms.p : nullptr, dmu.p, s));
  }
  if (var_out) {
    CHECK_RC(dprior.upload(prior_var, Ns));
    CHECK_RC(dvar.alloc(Ns));
    CHECK_RC(launch_colsumsq_sub(dV.p, ns_pad, Ns, N, dprior.p, dvar.p, -1.0, s));
  }
  if (cov_out) {
    CHECK_RC(dcov.alloc((size_t)ns_pad * ns_pad));
    SGP_HIP(hipMemsetAsync(dcov.p, 0, sizeof(double) * ns_pad * ns_pad, s));
    SGP_HIP(hipMemcpy2DAsync(dcov.p, sizeof(double) * ns_pad, prior_cov, sizeof(double) * ldp, sizeof(double) * Ns, (size_t)Ns,
                             hipMemcpyHostToDevice, s));
    CHECK_RC(launch_gemm_nt(dV.p, ns_pad, dV.p, ns_pad, dcov.p, ns_pad, ns_pad, ns_pad, n_pad, -1.0, 1.0, NOMASK, 0, 0, s));
  }
  SGP_HIP(hipStreamSynchronize(s));
  if (mean_out) SGP_HIP(hipMemcpy(mean_out, dmu.p, sizeof(double) * Ns, hipMemcpyDeviceToHost));
  if (var_out) SGP_HIP(hipMemcpy(var_out, dvar.p, sizeof(double) * Ns, hipMemcpyDeviceToHost));
  if (cov_out)
    SGP_HIP(hipMemcpy2D(cov_out, sizeof(double) * ldcov, dcov.p, sizeof(double) * ns_pad, sizeof(double) * Ns, (size_t)Ns,
                        hipMemcpyDeviceToHost));
  return 0;
}

// ---------------------------------------------------------------------------------------
// VFE: elbo and sparse posterior (App. A.6)
// ---------------------------------------------------------------------------------------
// o[0] = sum log s2_n, o[1] = sum delta_n^2, o[2] = sum var_n / s2_n, delta_n = (y_n - m_n)/sqrt(s2_n)
// Two stages (round 5: one 256-thread block over N = 262144 elements took 1.2 ms of the c4 step, latency-bound on its four
// waves): up to 128 blocks stride over the elements and leave one partial sum each, one block adds the partials in a fixed
// order -- the result depends on N only, not on the launch.
constexpr int ELBO_SC_BLOCKS = 128;
__global__ void elbo_scalars_kernel(const double* y, const double* mean, const double* var_x,
                                    int noise_kind, double sigma2, const double* noise, long N,
                                    double* delta, double* rsig, double* part) {
  __shared__ double sh[3][4];
  double a0 = 0, a1 = 0, a2 = 0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (long)gridDim.x * blockDim.x) {
    double s2 = noise_kind == 0 ? sigma2 : noise[i];
    double rs = 1.0 / sqrt(s2);
    double d = (y[i] - (mean ? mean[i] : 0.0)) * rs;
    delta[i] = d;
    rsig[i] = rs;
    a0 += log(s2);
    a1 = fma(d, d, a1);
    if (var_x) a2 += var_x[i] / s2;
  }
  for (int off = 32; off >= 1; off >>= 1) {
    a0 += __shfl_xor(a0, off, 64);
    a1 += __shfl_xor(a1, off, 64);
    a2 += __shfl_xor(a2, off, 64);
  }
  int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    sh[0][w] = a0;
    sh[1][w] = a1;
    sh[2][w] = a2;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int q = 0; q < 3; ++q) part[q * ELBO_SC_BLOCKS + blockIdx.x] = (sh[q][0] + sh[q][1]) + (sh[q][2] + sh[q][3]);
  }
}
__global__ void elbo_scalars_sum_kernel(const double* part, int blocks, double* o) {   // <<<1, ELBO_SC_BLOCKS>>>
  __shared__ double sh[3][2];
  const int t = threadIdx.x;
  double a[3];
  for (int q = 0; q < 3; ++q) a[q] = t < blocks ? part[q * ELBO_SC_BLOCKS + t] : 0.0;
  for (int off = 32; off >= 1; off >>= 1)
    for (int q = 0; q < 3; ++q) a[q] += __shfl_xor(a[q], off, 64);
  if ((t & 63) == 0)
    for (int q = 0; q < 3; ++q) sh[q][t >> 6] = a[q];
  __syncthreads();
  if (t == 0)
    for (int q = 0; q < 3; ++q) o[q] = sh[q][0] + sh[q][1];
}
static int launch_elbo_scalars(sgp_ctx* ctx, const double* y, const double* mean, const double* var_x, int noise_kind,
                               double sigma2, const double* noise, long N, double* delta, double* rsig, double* o,
                               hipStream_t s) {
  // the partial sums live at the tail of the context's scalar buffer (the per-column sums of a many-column logpdf use its
  // head; a context runs one operator at a time)
  double* part = ctx->d_scal + ctx->n_scal - 3 * ELBO_SC_BLOCKS;
  const int blocks = (int)std::max<long>(1, std::min<long>(ELBO_SC_BLOCKS, (N + 255) / 256));
  hipLaunchKernelGGL(elbo_scalars_kernel, dim3((unsigned)blocks), dim3(256), 0, s, y, mean, var_x, noise_kind, sigma2, noise, N,
                     delta, rsig, part);
  SGP_HIP(hipGetLastError());
  hipLaunchKernelGGL(elbo_scalars_sum_kernel, dim3(1), dim3(ELBO_SC_BLOCKS), 0, s, (const double*)part, blocks, o);
  SGP_HIP(hipGetLastError());
  return 0;
}
// per column j of the bordered rows R (nrows x ncols): dots[j] = sum_n R[n, j] delta[n],
// sq[j] = sum_n R[n, j]^2
__global__ void coldot_kernel(const double* R, long ld, long nrows, const double* delta, double* dots, double* sq) {
  __shared__ double sh[2][4];
  const long j = blockIdx.x;
  const double* col = R + j * ld;
  double a = 0, b = 0;
  for (long n = threadIdx.x; n < nrows; n += blockDim.x) {
    double v = col[n];
    a = fma(v, delta[n], a);
    b = fma(v, v, b);
  }
  for (int off = 32; off >= 1; off >>= 1) {
    a += __shfl_xor(a, off, 64);
    b += __shfl_xor(b, off, 64);
  }
  int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    sh[0][w] = a;
    sh[1][w] = b;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    dots[j] = (sh[0][0] + sh[0][1]) + (sh[0][2] + sh[0][3]);
    sq[j] = (sh[1][0] + sh[1][1]) + (sh[1][2] + sh[1][3]);
  }
}
__global__ void add_identity_kernel(double* G, long ld, long n) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) G[i + i * ld] += 1.0;
}
__global__ void accum_kernel(double* dst, const double* src) { dst[0] += src[0]; }
__global__ void set_row_kernel(double* G, long ld, long row, const double* v, long n) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) G[row + i * ld] = v[i];
}

struct sgp_sparse_post {
  sgp_ctx* ctx = nullptr;
  long ctx_serial = 0;
  long M = 0, m_pad = 0;
  double* dLz = nullptr;   // m_pad x m_pad factor of Kzz + Sigma_z (ld = m_pad)
  double* d_wz = nullptr;  // inverse diagonal blocks of Lz
  double* dG = nullptr;    // (m_pad + 128) x m_pad: factor of A A' + I, row m_pad = (Le^-1 A delta)'
  double* d_wg = nullptr;
  long ldg = 0;
};

extern "C" int sgp_sparse_posterior_destroy(sgp_sparse_post* p) {
  if (!p) return 0;
  if (p->dLz) hipFree(p->dLz);
  if (p->d_wz) hipFree(p->d_wz);
  if (p->dG) hipFree(p->dG);
  if (p->d_wg) hipFree(p->d_wg);
  delete p;
  return 0;
}

// The same pipeline for very many data points: K(z,z) is factored on its own and
// the rows K(x,z) Lambda go through in chunks of rows, each chunk a buffer of its own
// (leading dimension = chunk rows, so operand columns share pages and the TLB survives deep
// contractions): assemble -> scale -> blocked left-looking row solve (deep-K GEMMs against Lz) ->
// A delta / |A|^2 partials -> transpose -> split-K Gram partial, accumulated into G in chunk order
// (deterministic).  Nothing of size N x M is ever resident.
// Chunks of 65 536 rows (512 KB between operand columns, like the N = 65 536 factorisation) are as fast
// as the monolithic buffers on a good day (N = 262 144, M = 4096: 174 vs 177 ms) and three times
// faster on boxes where the monolithic 270 000-row leading dimension (2.1 MB between columns, every
// column in its own page) runs into the TLB: 628 ms there, same code, same inputs.  So anything
// beyond one chunk is chunked.  SGP_VFE_CHUNK=<rows> overrides both numbers (tests).
constexpr long VFE_CHUNK_ROWS = 65536;
constexpr long VFE_CHUNK_ABOVE = 65536;
static void vfe_chunking(long* chunk_rows, long* above) {
  const char* e = getenv("SGP_VFE_CHUNK");
  long v = e ? atol(e) / TILE * TILE : 0;
  *chunk_rows = v >= TILE ? v : VFE_CHUNK_ROWS;
  *above = v >= TILE ? v : VFE_CHUNK_ABOVE;
}

// The row-chunked pipeline is split in two so that the data points can be sharded (one
// exchange of M^2 + M + 2 doubles): `vfe_rows_partial` turns a slice of the data into its partial
// sums, `vfe_finish` consumes the (reduced) sums.  Layout of a "part" (device, contiguous):
//   G    (m_pad + 128) x m_pad, ld = m_pad + 128: lower tiles of sum_n a_n a_n' (no identity yet);
//        row m_pad is reserved for (A delta)'
//   dots m_pad: A delta
//   scal 8: [0] sum log s2_n, [1] delta' delta, [2] sum var_n / s2_n, [3] |A|_F^2
static long vfe_part_len(long m_pad) { return (m_pad + TILE) * m_pad + m_pad + 8; }

static int vfe_rows_partial(sgp_ctx* ctx, const sgp_dspec* dz, const sgp_dspec* dx, const double* var_x,
                            const double* mean_x, int noise_kind, const double* noise_x, int z_noise_kind,
                            const double* z_noise, const double* y, double* dLz, double* d_wz, double* d_part,
                            StageTimer& tm) {
  const long M = dz->N, N = dx->N;
  const long m_pad = rup(M, TILE), n_rows = rup(N, TILE);
  const long ldg = m_pad + TILE;
  double* dG = d_part;
  double* d_dots = d_part + ldg * m_pad;
  double* d_sc = d_dots + m_pad;
  long CH, ch_above;
  vfe_chunking(&CH, &ch_above);
  CH = std::min(CH, n_rows);
  hipStream_t s = ctx->stream;
  DevBuf dR, dAt, dPart, dy, dmean, dvar, ddelta, drsig, sq;
  NoiseDev ndx, ndz;
  CHECK_RC(dy.upload(y, N));
  if (mean_x) CHECK_RC(dmean.upload(mean_x, N));
  if (var_x) CHECK_RC(dvar.upload(var_x, N));
  CHECK_RC(upload_noise(ndx, noise_kind, noise_x, N));
  CHECK_RC(upload_noise(ndz, z_noise_kind, z_noise, M));
  CHECK_RC(ddelta.alloc(n_rows));
  CHECK_RC(drsig.alloc(n_rows));
  CHECK_RC(sq.alloc(m_pad));
  tm.mark(0);
  SGP_HIP(hipMemsetAsync(ctx->d_info, 0, sizeof(int), s));
  SGP_HIP(hipMemsetAsync(d_part, 0, sizeof(double) * vfe_part_len(m_pad), s));
  SGP_HIP(hipMemsetAsync(sq.p, 0, sizeof(double) * m_pad, s));
  SGP_HIP(hipMemsetAsync(ddelta.p, 0, sizeof(double) * n_rows, s));
  SGP_HIP(hipMemsetAsync(drsig.p, 0, sizeof(double) * n_rows, s));
  CHECK_RC(launch_elbo_scalars(ctx, dy.p, mean_x ? dmean.p : nullptr,
                     var_x ? dvar.p : nullptr, ndx.kind, ndx.sigma2, ndx.diag.p, N, ddelta.p, drsig.p, d_sc, s));
  // ---- Lz (replicated on every rank of a sharded run: M^3 / 3 flops, no communication)
  int nkz = ndz.kind == SGP_NOISE_DENSE ? -1 : ndz.kind;
  CHECK_RC(assemble(dz, dLz, m_pad, 0, m_pad / TILE, 0, m_pad / TILE, 1, nkz, ndz.sigma2, ndz.diag.p, s));
  if (ndz.kind == SGP_NOISE_DENSE) CHECK_RC(launch_add_dense(dLz, m_pad, ndz.dense.p, M, M, 1, s));
  CHECK_RC(launch_fill_pad(dLz, m_pad, M, m_pad, 0, m_pad, m_pad, 0, s));
  {
    // structural zeros (round 5): inducing points spread over independent processes give K(z,z) exact zero blocks, as they
    // give K(x,x) -- the same pattern machinery (no bordered rows here: T_r = T_c)
    SzMask szz;
    CHECK_RC(sz_build(ctx, dz, ndz.kind, m_pad, m_pad, s, &szz));
    CHECK_RC(chol_bordered(ctx, dLz, m_pad, m_pad, m_pad, d_wz, s, 0, szz.d_nz ? &szz : nullptr));
  }
  int info = fetch_info(ctx, s);
  if (info < 0) return -3;
  if (info > 0) {
    set_error("vfe: Kzz + Sigma_z is not positive definite (leading minor " + std::to_string(info) + ")");
    return info;
  }
  if (N == 0) return 0;
  const int nsplit = 8;  // one K slice per XCD (gemm_nt.hip, klo == 3)
  const long stride = ldg * m_pad;
  CHECK_RC(dR.alloc((size_t)CH * m_pad));
  // (Round 4 also built "two chunks in flight" -- the Gram product of chunk c on the second stream while chunk c + 1 is assembled
  // and solved on the first; measured on the N = 262144, M = 4096 bound, profiles/r04_experiments/elbo_c4.txt: 159.7 -> 161.5 ms,
  // two MFMA-bound streams lose more to each other than the solve's substitutions leave idle.  Removed in round 6.)
  CHECK_RC(dAt.alloc((size_t)m_pad * CH));
  DevBuf dCs;
  CHECK_RC(dCs.alloc((size_t)transpose_colsum_scratch(CH, m_pad)));
  {   // slabs of the chunk lengths actually used: the `sub` split depends on K % (8 sub 16), so a shorter LAST chunk can
      // need more slabs than a full one (advisor, round 4: SGP_VFE_CHUNK = 768, M = 4096, 1280 rows -> 8 vs 32)
    const long last = n_rows % CH == 0 ? std::min(CH, n_rows) : n_rows % CH;
    const long slabs = std::max(splitk_slabs(m_pad, std::min(CH, n_rows), nsplit), splitk_slabs(m_pad, last, nsplit));
    CHECK_RC(dPart.alloc((size_t)slabs * stride));
  }
  // ---- row chunks
  long chunk = 0;
  for (long r0 = 0; r0 < n_rows; r0 += CH, ++chunk) {
    const long ch = std::min(CH, n_rows - r0);            // rows of this chunk (multiple of 128)
    const long nv = std::max<long>(0, std::min(N - r0, ch));  // of which real data points
    double* At = dAt.p;
    tm.mark(1);
    // (a full chunk of an unpadded M is written entry for entry by the assembly: nothing to clear)
    if (nv < ch || M < m_pad) SGP_HIP(hipMemsetAsync(dR.p, 0, sizeof(double) * ch * m_pad, s));
    // global row r of K(x,z) lands at dR[(r - r0) + c * ch]
    CHECK_RC(assemble(dx, dR.p - r0, ch, r0 / TILE, (r0 + ch) / TILE, 0, m_pad / TILE, 0, -1, 0.0, nullptr, s));
    tm.mark(2);
    // The rows' Lambda_y^-1/2 factors commute with the solve (it acts on the inducing dimension): instead of a pass of
    // their own over the chunk before it (2 x 2.1 GB at N = 262 144, M = 4096: 0.9 ms per chunk) they are applied where
    // the solved rows are read anyway -- the column sums and the transposition (padded rows carry the factor 0)
    CHECK_RC(row_trsm(ctx, dR.p, ch, ch, dLz, m_pad, d_wz, m_pad, s));
    tm.mark(3);
    // (round 6: ONE pass over the solved rows -- the column sums ride with the transposition, 6.0 -> 3.6 ms per step at
    // N = 262 144, M = 4096)
    CHECK_RC(launch_transpose_colsum(dR.p, ch, ch, m_pad, At, m_pad, drsig.p + r0, ddelta.p + r0, d_dots, sq.p, r0 > 0 ? 1 : 0,
                                     dCs.p, s));
    tm.mark(4);
    // (the split-K slices need ch to be a multiple of 16 * nsplit = 128: it is)
    CHECK_RC(launch_gemm_nt_splitk(At, m_pad, At, m_pad, dPart.p, ldg, m_pad, m_pad, ch, nsplit, stride, 1, s));
    CHECK_RC(launch_splitk_reduce(dPart.p, stride, nsplit, dG, ldg, m_pad, m_pad, 1.0, r0 > 0 ? 1.0 : 0.0, 1, s, ch));
  }
  tm.mark(5);
  CHECK_RC(launch_sum_array(sq.p, m_pad, d_sc + 3, s));
  SGP_HIP(hipStreamSynchronize(s));  // chunk buffers go back to the cache at scope exit
  return 0;
}

// h[0..5] as documented at vfe_pipeline; d_wg (optional) keeps the inverse diagonal blocks of Le
static int vfe_finish(sgp_ctx* ctx, double* d_part, long m_pad, double* d_wg, double* h, StageTimer& tm) {
  hipStream_t s = ctx->stream;
  const long ldg = m_pad + TILE;
  double* dG = d_part;
  double* d_dots = d_part + ldg * m_pad;
  double* d_sc = d_dots + m_pad;
  tm.mark(5);
  hipLaunchKernelGGL(add_identity_kernel, dim3((unsigned)((m_pad + 255) / 256)), dim3(256), 0, s, dG, ldg, m_pad);
  hipLaunchKernelGGL(set_row_kernel, dim3((unsigned)((m_pad + 255) / 256)), dim3(256), 0, s, dG, ldg, m_pad, d_dots,
                     m_pad);
  SGP_HIP(hipGetLastError());
  SGP_HIP(hipMemsetAsync(ctx->d_info, 0, sizeof(int), s));
  CHECK_RC(chol_bordered(ctx, dG, ldg, m_pad, ldg, d_wg, s));
  CHECK_RC(launch_sum_array(ctx->d_slots, m_pad / TILE, d_sc + 4, s));
  CHECK_RC(launch_rowsumsq(dG + m_pad, ldg, m_pad, 1, d_sc + 5, 0, s));
  SGP_HIP(hipMemcpyAsync(h, d_sc, sizeof(double) * 6, hipMemcpyDeviceToHost, s));
  tm.finish();
  int info = fetch_info(ctx, s);
  if (info < 0) return -3;
  if (info > 0) {
    set_error("vfe: A A' + I is not positive definite (leading minor " + std::to_string(info) + ")");
    return info;
  }
  return 0;
}

static int vfe_pipeline_chunked(sgp_ctx* ctx, const sgp_dspec* dz, const sgp_dspec* dx, const double* var_x,
                                const double* mean_x, int noise_kind, const double* noise_x, int z_noise_kind,
                                const double* z_noise, const double* y, double* h, sgp_sparse_post* keep) {
  const long M = dz->N;
  const long m_pad = rup(M, TILE);
  DevBuf dLz_local, dinv_local, dG_local;
  double *dLz = nullptr, *d_wz = nullptr, *d_part = nullptr;
  if (keep) {
    SGP_HIP(hipMalloc(&keep->dLz, sizeof(double) * m_pad * m_pad));
    SGP_HIP(hipMalloc(&keep->d_wz, sizeof(double) * (m_pad / TILE) * INVD_STRIDE));
    SGP_HIP(hipMalloc(&keep->dG, sizeof(double) * vfe_part_len(m_pad)));
    SGP_HIP(hipMalloc(&keep->d_wg, sizeof(double) * (m_pad / TILE) * INVD_STRIDE));
    dLz = keep->dLz;
    d_wz = keep->d_wz;
    d_part = keep->dG;
  } else {
    CHECK_RC(dLz_local.alloc((size_t)m_pad * m_pad));
    CHECK_RC(dinv_local.alloc((size_t)(m_pad / TILE) * INVD_STRIDE));
    CHECK_RC(dG_local.alloc((size_t)vfe_part_len(m_pad)));
    dLz = dLz_local.p;
    d_wz = dinv_local.p;
    d_part = dG_local.p;
  }
  StageTimer tm(ctx, ctx->stream);
  CHECK_RC(vfe_rows_partial(ctx, dz, dx, var_x, mean_x, noise_kind, noise_x, z_noise_kind, z_noise, y, dLz, d_wz,
                            d_part, tm));
  CHECK_RC(vfe_finish(ctx, d_part, m_pad, keep ? keep->d_wg : nullptr, h, tm));
  if (keep) {
    keep->ctx = ctx;
    keep->ctx_serial = ctx->serial;
    keep->M = M;
    keep->m_pad = m_pad;
    keep->ldg = m_pad + TILE;
  }
  return 0;
}

// ---- N-sharded ELBO building blocks (device pointers, see sthenomi.h) ---------------
extern "C" int sgp_elbo_part_len(int64_t M, int64_t* len) {
  CHECK_ARG(M >= 1 && len, "sgp_elbo_part_len: bad argument");
  *len = vfe_part_len(rup(M, TILE));
  return 0;
}

// one rank's share: its slice of the data -> partial sums in d_part; the factor of K(z,z) + Sigma_z and its inverse
// diagonal blocks go to dLz (m_pad x m_pad) / d_wz when given (a sparse posterior keeps them), else to scratch
static int elbo_partial_core(sgp_ctx* ctx, const sgp_cov_spec* zz, const sgp_cov_spec* xz, const double* var_x,
                             const double* mean_x, int noise_kind, const double* noise_x, int z_noise_kind,
                             const double* z_noise, const double* y, double* dLz_keep, double* d_wz_keep,
                             double* d_part, int64_t part_len) {
  CHECK_ARG(ctx && zz && xz && noise_x && z_noise && d_part, "sgp_dev_elbo_partial: NULL argument");
  CHECK_ARG(zz->symmetric, "vfe: zz spec must be symmetric");
  CHECK_ARG(noise_kind == SGP_NOISE_SCALAR || noise_kind == SGP_NOISE_DIAG,
            "vfe: Sigma_y must be isotropic or diagonal (as in AbstractGPs.elbo)");
  CtxScope scope(ctx);
  SpecGuard gz, gx;
  CHECK_RC(dspec_create(ctx, zz, &gz.ds));
  CHECK_RC(dspec_create(ctx, xz, &gx.ds));
  const long M = gz.ds->N, N = gx.ds->N;
  CHECK_ARG(gx.ds->M == M && M >= 1, "vfe: xz spec columns != number of inducing points");
  CHECK_ARG(N == 0 || y, "sgp_dev_elbo_partial: NULL data");
  const long m_pad = rup(M, TILE);
  CHECK_ARG(part_len >= vfe_part_len(m_pad), "sgp_dev_elbo_partial: part buffer too small (sgp_elbo_part_len)");
  DevBuf dLz, dwz;
  if (!dLz_keep) CHECK_RC(dLz.alloc((size_t)m_pad * m_pad));
  if (!d_wz_keep) CHECK_RC(dwz.alloc((size_t)(m_pad / TILE) * INVD_STRIDE));
  StageTimer tm(ctx, ctx->stream);
  CHECK_RC(vfe_rows_partial(ctx, gz.ds, gx.ds, var_x, mean_x, noise_kind, noise_x, z_noise_kind, z_noise, y,
                            dLz_keep ? dLz_keep : dLz.p, d_wz_keep ? d_wz_keep : dwz.p, d_part, tm));
  tm.finish();
  return 0;
}

extern "C" int sgp_dev_elbo_partial(sgp_ctx* ctx, const sgp_cov_spec* zz, const sgp_cov_spec* xz,
                                    const double* var_x, const double* mean_x, int noise_kind,
                                    const double* noise_x, int z_noise_kind, const double* z_noise,
                                    const double* y, double* d_part, int64_t part_len) {
  CHECK_ARG(xz == nullptr || spec_rows_host(xz) == 0 || var_x, "sgp_dev_elbo_partial: NULL data");
  return elbo_partial_core(ctx, zz, xz, var_x, mean_x, noise_kind, noise_x, z_noise_kind, z_noise, y, nullptr, nullptr,
                           d_part, part_len);
}

static int elbo_finish_core(sgp_ctx* ctx, int64_t M, double* d_part, double* d_wg, double* h) {
  CtxScope scope(ctx);
  StageTimer tm(ctx, ctx->stream);
  return vfe_finish(ctx, d_part, rup(M, TILE), d_wg, h, tm);
}

extern "C" int sgp_dev_elbo_finish(sgp_ctx* ctx, int64_t M, int64_t N_total, double* d_part, double* out) {
  CHECK_ARG(ctx && d_part && out && M >= 1 && N_total >= 1, "sgp_dev_elbo_finish: bad argument");
  double h[6];
  CHECK_RC(elbo_finish_core(ctx, M, d_part, nullptr, h));
  double tmp = h[0] + h[4] + h[1] - h[5];
  double dtc = -0.5 * ((double)N_total * 1.8378770664093453 + tmp);
  out[0] = dtc - 0.5 * (h[2] - h[3]);
  return 0;
}

// shared VFE pipeline.  Returns elbo terms in h[0..5]:
//  h[0]=sum log s2, h[1]=delta'delta, h[2]=sum var/s2, h[3]=|A|_F^2, h[4]=logdet Le, h[5]=|Le^-1 A delta|^2
static int vfe_pipeline(sgp_ctx* ctx, const sgp_cov_spec* zz, const sgp_cov_spec* xz,
                        const double* var_x, const double* mean_x, int noise_kind,
                        const double* noise_x, int z_noise_kind, const double* z_noise,
                        const double* y, double* h, sgp_sparse_post* keep) {
  CHECK_ARG(zz->symmetric, "vfe: zz spec must be symmetric");
  CHECK_ARG(noise_kind == SGP_NOISE_SCALAR || noise_kind == SGP_NOISE_DIAG,
            "vfe: Sigma_y must be isotropic or diagonal (as in AbstractGPs.elbo)");
  SpecGuard gz, gx;
  CHECK_RC(dspec_create(ctx, zz, &gz.ds));
  CHECK_RC(dspec_create(ctx, xz, &gx.ds));
  long M = gz.ds->N, N = gx.ds->N;
  CHECK_ARG(gx.ds->M == M, "vfe: xz spec columns != number of inducing points");
  CHECK_ARG(M >= 1 && N >= 1, "vfe: empty inputs");
  long m_pad = rup(M, TILE), n_rows = rup(N, TILE);
  long ch_rows, ch_above;
  vfe_chunking(&ch_rows, &ch_above);
  if (n_rows > ch_above)
    return vfe_pipeline_chunked(ctx, gz.ds, gx.ds, var_x, mean_x, noise_kind, noise_x, z_noise_kind, z_noise, y, h,
                                keep);
  long ld = m_pad + n_rows;
  hipStream_t s = ctx->stream;
  DevBuf dA, dy, dmean, dvar, ddelta, drsig, dots, sq, dG_local;
  NoiseDev ndx, ndz;
  CHECK_RC(dA.alloc((size_t)ld * m_pad));
  CHECK_RC(dy.upload(y, N));
  if (mean_x) CHECK_RC(dmean.upload(mean_x, N));
  if (var_x) CHECK_RC(dvar.upload(var_x, N));
  CHECK_RC(upload_noise(ndx, noise_kind, noise_x, N));
  CHECK_RC(upload_noise(ndz, z_noise_kind, z_noise, M));
  CHECK_RC(ddelta.alloc(n_rows));
  CHECK_RC(drsig.alloc(n_rows));
  CHECK_RC(dots.alloc(m_pad));
  CHECK_RC(sq.alloc(m_pad));
  double* d_o = ctx->d_scal + 1;  // 3 scalars
  SGP_HIP(hipMemsetAsync(ctx->d_info, 0, sizeof(int), s));
  SGP_HIP(hipMemsetAsync(ddelta.p, 0, sizeof(double) * n_rows, s));
  SGP_HIP(hipMemsetAsync(drsig.p, 0, sizeof(double) * n_rows, s));
  CHECK_RC(launch_elbo_scalars(ctx, dy.p, mean_x ? dmean.p : nullptr,
                     var_x ? dvar.p : nullptr, ndx.kind, ndx.sigma2, ndx.diag.p, N, ddelta.p, drsig.p,
                     d_o, s));
  // top: Kzz + Sigma_z (lower), identity padding; bottom rows: K(x, z) Lambda_y^-1
  int nkz = ndz.kind == SGP_NOISE_DENSE ? -1 : ndz.kind;
  CHECK_RC(assemble(gz.ds, dA.p, ld, 0, m_pad / TILE, 0, m_pad / TILE, 1, nkz, ndz.sigma2, ndz.diag.p, s));
  if (ndz.kind == SGP_NOISE_DENSE) CHECK_RC(launch_add_dense(dA.p, ld, ndz.dense.p, M, M, 1, s));
  CHECK_RC(launch_fill_pad(dA.p, ld, M, m_pad, 0, m_pad, ld, 0, s));
  // zero the row padding of the bordered block, then assemble the cross block into it
  if (n_rows > N) {
    // rows [m_pad + N, ld) of every column
    SGP_HIP(hipMemset2DAsync(dA.p + m_pad + N, sizeof(double) * ld, 0, sizeof(double) * (n_rows - N),
                             (size_t)m_pad, s));
  }
  CHECK_RC(assemble(gx.ds, dA.p + m_pad, ld, 0, n_rows / TILE, 0, m_pad / TILE, 0, -1, 0.0, nullptr, s));
  CHECK_RC(launch_scale_rows(dA.p + m_pad, ld, N, m_pad, drsig.p, s));
  double* d_wz = nullptr;
  if (keep) {
    SGP_HIP(hipMalloc(&keep->d_wz, sizeof(double) * (m_pad / TILE) * INVD_STRIDE));
    d_wz = keep->d_wz;
  }
  // The N rows K(x,z) Lambda ride along as bordered rows of the factorisation (K = 512 updates).
  // Factoring Kzz alone and solving the rows with the deep-K row_trsm was tried: with a leading
  // dimension of 270 000 doubles every operand column lies in its own 2 MiB page and a K = 3584 GEMM
  // thrashes the TLB (N = 262 144, M = 4096: 225 -> 700 ms).
  {
    SzMask szz;   // structural zeros of K(z,z) (round 5); the N rows K(x,z) Lambda are dense bordered rows of the pattern
    CHECK_RC(sz_build(ctx, gz.ds, ndz.kind, m_pad, ld, s, &szz));
    CHECK_RC(chol_bordered(ctx, dA.p, ld, m_pad, ld, d_wz, s, 0, szz.d_nz ? &szz : nullptr));
  }
  int info = fetch_info(ctx, s);
  if (info < 0) return -3;
  if (info > 0) {
    set_error("vfe: Kzz + Sigma_z is not positive definite (leading minor " + std::to_string(info) + ")");
    return info;
  }
  // rows now hold A' (N x M).  A delta, |A|_F^2
  const double* R = dA.p + m_pad;
  hipLaunchKernelGGL(coldot_kernel, dim3((unsigned)m_pad), dim3(256), 0, s, R, ld, n_rows, ddelta.p,
                     dots.p, sq.p);
  SGP_HIP(hipGetLastError());
  CHECK_RC(launch_sum_array(sq.p, m_pad, ctx->d_scal + 4, s));
  // G = A A' + I, bordered with (A delta)'
  long ldg = m_pad + TILE;
  double* dG = nullptr;
  if (keep) {
    SGP_HIP(hipMalloc(&keep->dG, sizeof(double) * ldg * m_pad));
    SGP_HIP(hipMalloc(&keep->d_wg, sizeof(double) * (m_pad / TILE) * INVD_STRIDE));
    dG = keep->dG;
  } else {
    CHECK_RC(dG_local.alloc((size_t)ldg * m_pad));
    dG = dG_local.p;
  }
  SGP_HIP(hipMemsetAsync(dG, 0, sizeof(double) * ldg * m_pad, s));
  // G = A A' with the contraction over the N data points.  The rows hold A' (N x M, row index
  // contiguous = data point), so A (M x N) is formed by one coalesced transpose and the Gram
  // matrix runs on the NT MFMA kernel, split over K to fill the chip (few output tiles, huge K),
  // then reduced in fixed order.
  {
    DevBuf dAt, dPart;
    CHECK_RC(dAt.alloc((size_t)m_pad * n_rows));
    CHECK_RC(launch_transpose_add(R, ld, n_rows, m_pad, dAt.p, m_pad, nullptr, s));
    long tiles = (m_pad / TILE) * (m_pad / TILE + 1) / 2;
    int nsplit = 1;
    while (tiles * nsplit < 4096 && nsplit < 64 && (n_rows % (16L * nsplit * 2)) == 0 &&
           n_rows / (nsplit * 2) >= 2048)
      nsplit *= 2;
    long stride = ldg * m_pad;
    CHECK_RC(dPart.alloc((size_t)splitk_slabs(m_pad, n_rows, nsplit) * stride));
    CHECK_RC(launch_gemm_nt_splitk(dAt.p, m_pad, dAt.p, m_pad, dPart.p, ldg, m_pad, m_pad, n_rows, nsplit,
                                   stride, 1, s));
    CHECK_RC(launch_splitk_reduce(dPart.p, stride, nsplit, dG, ldg, m_pad, m_pad, 1.0, 0.0, 1, s, n_rows));
    SGP_HIP(hipStreamSynchronize(s));  // dAt / dPart are freed at scope exit
  }
  hipLaunchKernelGGL(add_identity_kernel, dim3((unsigned)((m_pad + 255) / 256)), dim3(256), 0, s, dG,
                     ldg, m_pad);
  hipLaunchKernelGGL(set_row_kernel, dim3((unsigned)((m_pad + 255) / 256)), dim3(256), 0, s, dG, ldg,
                     m_pad, dots.p, m_pad);
  SGP_HIP(hipGetLastError());
  SGP_HIP(hipMemsetAsync(ctx->d_info, 0, sizeof(int), s));
  CHECK_RC(chol_bordered(ctx, dG, ldg, m_pad, ldg, keep ? keep->d_wg : nullptr, s));
  CHECK_RC(launch_sum_array(ctx->d_slots, m_pad / TILE, ctx->d_scal + 5, s));
  CHECK_RC(launch_rowsumsq(dG + m_pad, ldg, m_pad, 1, ctx->d_scal + 6, 0, s));
  SGP_HIP(hipMemcpyAsync(h, ctx->d_scal + 1, sizeof(double) * 6, hipMemcpyDeviceToHost, s));
  info = fetch_info(ctx, s);
  if (info < 0) return -3;
  if (info > 0) {
    set_error("vfe: A A' + I is not positive definite (leading minor " + std::to_string(info) + ")");
    return info;
  }
  if (keep) {
    keep->ctx = ctx;
    keep->ctx_serial = ctx->serial;
    keep->M = M;
    keep->m_pad = m_pad;
    keep->ldg = ldg;
    // keep Lz: copy the top m_pad x m_pad block
    SGP_HIP(hipMalloc(&keep->dLz, sizeof(double) * m_pad * m_pad));
    SGP_HIP(hipMemcpy2DAsync(keep->dLz, sizeof(double) * m_pad, dA.p, sizeof(double) * ld,
                             sizeof(double) * m_pad, (size_t)m_pad, hipMemcpyDeviceToDevice, s));
    SGP_HIP(hipStreamSynchronize(s));
  }
  return 0;
}

static int sgp_elbo_impl(sgp_ctx* ctx, const sgp_cov_spec* zz, const sgp_cov_spec* xz,
                        const double* var_x, const double* mean_x, int noise_kind,
                        const double* noise_x, int z_noise_kind, const double* z_noise,
                        const double* y, double* out) {
  CHECK_ARG(ctx && zz && xz && var_x && noise_x && z_noise && y && out, "sgp_elbo: NULL argument");
  CtxScope scope(ctx);
  if (ctx->multi && ctx->multi_nranks > 1)   // data points sharded over the ranks, one reduction of M^2 + M + 2 doubles
  {
    double hm[6];
    CHECK_RC(sgp_multi_vfe(ctx, zz, xz, var_x, mean_x, noise_kind, noise_x, z_noise_kind, z_noise, y, hm, nullptr, nullptr,
                           nullptr, nullptr));
    const long Nm = spec_rows_host(xz);
    out[0] = -0.5 * ((double)Nm * 1.8378770664093453 + hm[0] + hm[4] + hm[1] - hm[5]) - 0.5 * (hm[2] - hm[3]);
    return 0;
  }
  double h[6];
  CHECK_RC(vfe_pipeline(ctx, zz, xz, var_x, mean_x, noise_kind, noise_x, z_noise_kind, z_noise, y, h,
                        nullptr));
  long N = 0;
  for (int i = 0; i < xz->n_row_blocks; ++i) N += xz->row_len[i];
  // elbo = -(N log 2pi + logdet Sy + logdet Le + d'd - |Le^-1 A d|^2)/2 - (sum var/s2 - |A|_F^2)/2
  double tmp = h[0] + h[4] + h[1] - h[5];
  double dtc = -0.5 * ((double)N * 1.8378770664093453 + tmp);
  out[0] = dtc - 0.5 * (h[2] - h[3]);
  return 0;
}
extern "C" int sgp_elbo(sgp_ctx* ctx, const sgp_cov_spec* zz, const sgp_cov_spec* xz,
                        const double* var_x, const double* mean_x, int noise_kind,
                        const double* noise_x, int z_noise_kind, const double* z_noise,
                        const double* y, double* out) {
  return with_df_fallback(ctx, [&]() { return sgp_elbo_impl(ctx, zz, xz, var_x, mean_x, noise_kind, noise_x, z_noise_kind, z_noise, y, out); });
}

// ---------------------------------------------------------------------------------------
// elbo + reverse-mode gradient (second half).  Formulas: oracle/abstractgps.py
// elbo_gradient_wrt_cov (checked against finite differences there).  Both factorisations carry
// identity rows (row-limited, `grow`) so that J = Lz^-T and Le^-T come out of the panel solves:
//   dA' = R Z + delta u'          (R = A', Z = I - B^-1 - u u')
//   G_xz = Lambda (R Z J' + delta (J u)')                      -> contraction over the xz spec
//   G_zz = -1/2 J (B + B^-1 - 2 I + u u') J'                   -> contraction over the zz spec
// ---------------------------------------------------------------------------------------
static int elbo_grad_core(sgp_ctx* ctx, const sgp_cov_spec* zz, const sgp_cov_spec* xz, const double* var_x,
                          const double* mean_x, int noise_kind, const double* noise_x, int z_noise_kind,
                          const double* z_noise, const double* y, double* elbo_out, double* grad_y,
                          double* grad_mean, double* grad_noise, double* grad_var_x, double* grad_z_noise,
                          double* grad_coef_zz, double* grad_inscale_zz, double* grad_coef_xz,
                          double* grad_inscale_xz, double* const* grad_inputs_zz, double* const* grad_inputs_xz,
                          double* const* grad_rowscale_zz = nullptr, double* const* grad_rowscale_xz = nullptr,
                          double* const* grad_colscale_xz = nullptr, const sgp::ElboGradShard* shard = nullptr) {
  // shard (round 6, multi.hip: sgp_multi_elbo_grad): this call sees ONE rank's slice of the data points (xz, var_x, mean_x, a
  // diagonal noise_x, y and the per-point results are the slice's).  The sums over data points -- A A', A delta, the four
  // scalar sums -- are added up over the ranks by shard->reduce (every rank gets the total), the M x M stage runs replicated
  // on identical numbers, the data-point stage on the slice; the zz-side results are the primary rank's business only.
  CHECK_ARG(ctx && zz && xz && var_x && noise_x && z_noise && y && elbo_out, "sgp_elbo_grad: NULL argument");
  CHECK_ARG(zz->symmetric, "sgp_elbo_grad: zz spec must be symmetric");
  CHECK_ARG(noise_kind == SGP_NOISE_SCALAR || noise_kind == SGP_NOISE_DIAG,
            "sgp_elbo_grad: Sigma_y must be isotropic or diagonal (as in AbstractGPs.elbo)");
  CHECK_ARG(z_noise_kind >= SGP_NOISE_SCALAR && z_noise_kind <= SGP_NOISE_DENSE, "sgp_elbo_grad: bad Sigma_z kind");
  CtxScope scope(ctx);
  SpecGuard gz, gx;
  CHECK_RC(dspec_create(ctx, zz, &gz.ds));
  CHECK_RC(dspec_create(ctx, xz, &gx.ds));
  const long M = gz.ds->N, N = gx.ds->N;
  CHECK_ARG(gx.ds->M == M, "sgp_elbo_grad: xz spec columns != number of inducing points");
  CHECK_ARG(M >= 1 && N >= 1, "sgp_elbo_grad: empty inputs");
  const long m_pad = rup(M, TILE), n_rows = rup(N, TILE);
  const long ld = m_pad + n_rows + m_pad;   // [Kzz + Sigma_z ; K(x,z) Lambda ; I]
  const long ldg = m_pad + TILE + m_pad;    // [A A' + I ; (A delta)' ; I]
  hipStream_t s = ctx->stream;
  DevBuf dA, dG, dB, dBinv, dZ, dS, dRZ, dT1, dGzz, dy, dmean, dvar, ddelta, drsig, dots, sq, du, dut, dgy, dgsy,
      dpart, dgcz, dgsz, dgcx, dgsx;
  NoiseDev ndx, ndz;
  CHECK_RC(dA.alloc((size_t)ld * m_pad));
  CHECK_RC(dG.alloc((size_t)ldg * m_pad));
  CHECK_RC(dB.alloc((size_t)m_pad * m_pad));
  CHECK_RC(dBinv.alloc((size_t)m_pad * m_pad));
  CHECK_RC(dZ.alloc((size_t)m_pad * m_pad));
  CHECK_RC(dS.alloc((size_t)m_pad * m_pad));
  CHECK_RC(dT1.alloc((size_t)m_pad * m_pad));
  CHECK_RC(dGzz.alloc((size_t)m_pad * m_pad));
  CHECK_RC(dRZ.alloc((size_t)n_rows * m_pad));
  CHECK_RC(dy.upload(y, N));
  if (mean_x) CHECK_RC(dmean.upload(mean_x, N));
  CHECK_RC(dvar.upload(var_x, N));
  CHECK_RC(upload_noise(ndx, noise_kind, noise_x, N));
  CHECK_RC(upload_noise(ndz, z_noise_kind, z_noise, M));
  CHECK_RC(ddelta.alloc(n_rows));
  CHECK_RC(drsig.alloc(n_rows));
  CHECK_RC(dots.alloc(m_pad));
  CHECK_RC(sq.alloc(m_pad));
  CHECK_RC(du.alloc(m_pad));
  CHECK_RC(dut.alloc(m_pad));
  CHECK_RC(dgy.alloc(N));
  CHECK_RC(dgsy.alloc(N));
  const size_t ntz = std::max<size_t>(1, gz.ds->h_terms.size()), ntx = std::max<size_t>(1, gx.ds->h_terms.size());
  CHECK_RC(dgcz.alloc(ntz));
  CHECK_RC(dgsz.alloc(ntz));
  CHECK_RC(dgcx.alloc(ntx));
  CHECK_RC(dgsx.alloc(ntx));
  SGP_HIP(hipMemsetAsync(dgcz.p, 0, sizeof(double) * ntz, s));
  SGP_HIP(hipMemsetAsync(dgsz.p, 0, sizeof(double) * ntz, s));
  SGP_HIP(hipMemsetAsync(dgcx.p, 0, sizeof(double) * ntx, s));
  SGP_HIP(hipMemsetAsync(dgsx.p, 0, sizeof(double) * ntx, s));
  double* d_o = ctx->d_scal + 1;  // h[0..2]; h[3] at +4, h[4] at +5, h[5] at +6
  SGP_HIP(hipMemsetAsync(ctx->d_info, 0, sizeof(int), s));
  SGP_HIP(hipMemsetAsync(ddelta.p, 0, sizeof(double) * n_rows, s));
  SGP_HIP(hipMemsetAsync(drsig.p, 0, sizeof(double) * n_rows, s));
  CHECK_RC(launch_elbo_scalars(ctx, dy.p, mean_x ? dmean.p : nullptr, dvar.p,
                     ndx.kind, ndx.sigma2, ndx.diag.p, N, ddelta.p, drsig.p, d_o, s));
  // ---- factor 1: [Kzz + Sigma_z ; K(x,z) Lambda ; I]  ->  Lz, R = A', J = Lz^-T
  double* Rw = dA.p + m_pad;
  double* Jm = dA.p + m_pad + n_rows;
  CHECK_RC(assemble(gz.ds, dA.p, ld, 0, m_pad / TILE, 0, m_pad / TILE, 1, ndz.kind == SGP_NOISE_DENSE ? -1 : ndz.kind,
                    ndz.sigma2, ndz.diag.p, s));
  if (ndz.kind == SGP_NOISE_DENSE) CHECK_RC(launch_add_dense(dA.p, ld, ndz.dense.p, M, M, 1, s));   // dense Sigma_z (M x M, ld M)
  CHECK_RC(launch_fill_pad(dA.p, ld, M, m_pad, 0, m_pad, ld, 0, s));
  if (n_rows > N)
    SGP_HIP(hipMemset2DAsync(Rw + N, sizeof(double) * ld, 0, sizeof(double) * (n_rows - N), (size_t)m_pad, s));
  CHECK_RC(assemble(gx.ds, Rw, ld, 0, n_rows / TILE, 0, m_pad / TILE, 0, -1, 0.0, nullptr, s));
  CHECK_RC(launch_scale_rows(Rw, ld, N, m_pad, drsig.p, s));
  SGP_HIP(hipMemset2DAsync(Jm, sizeof(double) * ld, 0, sizeof(double) * m_pad, (size_t)m_pad, s));
  hipLaunchKernelGGL(add_identity_kernel, dim3((unsigned)((m_pad + 255) / 256)), dim3(256), 0, s, Jm, ld, m_pad);
  SGP_HIP(hipGetLastError());
  CHECK_RC(chol_bordered(ctx, dA.p, ld, m_pad, ld, nullptr, s, m_pad + n_rows));
  int info = fetch_info(ctx, s);
  if (info < 0) return -3;
  if (info > 0) {
    set_error("vfe: Kzz + Sigma_z is not positive definite (leading minor " + std::to_string(info) + ")");
    return info;
  }
  const double* R = Rw;
  hipLaunchKernelGGL(coldot_kernel, dim3((unsigned)m_pad), dim3(256), 0, s, R, ld, n_rows, ddelta.p, dots.p, sq.p);
  SGP_HIP(hipGetLastError());
  CHECK_RC(launch_sum_array(sq.p, m_pad, ctx->d_scal + 4, s));
  // ---- B = A A' + I (transpose + split-K Gram, as in the forward pass), kept full in dB
  SGP_HIP(hipMemsetAsync(dG.p, 0, sizeof(double) * ldg * m_pad, s));
  {
    DevBuf dAt, dPart;
    CHECK_RC(dAt.alloc((size_t)m_pad * n_rows));
    CHECK_RC(launch_transpose_add(R, ld, n_rows, m_pad, dAt.p, m_pad, nullptr, s));
    long tiles = (m_pad / TILE) * (m_pad / TILE + 1) / 2;
    int nsplit = 1;
    while (tiles * nsplit < 4096 && nsplit < 64 && (n_rows % (16L * nsplit * 2)) == 0 &&
           n_rows / (nsplit * 2) >= 2048)
      nsplit *= 2;
    long stride = ldg * m_pad;
    CHECK_RC(dPart.alloc((size_t)splitk_slabs(m_pad, n_rows, nsplit) * stride));
    CHECK_RC(launch_gemm_nt_splitk(dAt.p, m_pad, dAt.p, m_pad, dPart.p, ldg, m_pad, m_pad, n_rows, nsplit, stride, 1,
                                   s));
    CHECK_RC(launch_splitk_reduce(dPart.p, stride, nsplit, dG.p, ldg, m_pad, m_pad, 1.0, 0.0, 1, s, n_rows));
    SGP_HIP(hipStreamSynchronize(s));
  }
  if (shard) {
    // the slice's sums, packed: A A' (m_pad x m_pad, lower), A delta (m_pad), h[0..3] -> summed over the ranks -> back
    const long mm = m_pad * m_pad, len = mm + m_pad + 8;
    DevBuf dP;
    CHECK_RC(dP.alloc((size_t)len));
    SGP_HIP(hipMemsetAsync(dP.p + mm + m_pad, 0, sizeof(double) * 8, s));
    SGP_HIP(hipMemcpy2DAsync(dP.p, sizeof(double) * m_pad, dG.p, sizeof(double) * ldg, sizeof(double) * m_pad, (size_t)m_pad,
                             hipMemcpyDeviceToDevice, s));
    SGP_HIP(hipMemcpyAsync(dP.p + mm, dots.p, sizeof(double) * m_pad, hipMemcpyDeviceToDevice, s));
    SGP_HIP(hipMemcpyAsync(dP.p + mm + m_pad, ctx->d_scal + 1, sizeof(double) * 4, hipMemcpyDeviceToDevice, s));
    SGP_HIP(hipStreamSynchronize(s));
    CHECK_RC(shard->reduce(dP.p, len));   // collective: returns on every rank with the total in place, visible to `s`
    SGP_HIP(hipMemcpy2DAsync(dG.p, sizeof(double) * ldg, dP.p, sizeof(double) * m_pad, sizeof(double) * m_pad, (size_t)m_pad,
                             hipMemcpyDeviceToDevice, s));
    SGP_HIP(hipMemcpyAsync(dots.p, dP.p + mm, sizeof(double) * m_pad, hipMemcpyDeviceToDevice, s));
    SGP_HIP(hipMemcpyAsync(ctx->d_scal + 1, dP.p + mm + m_pad, sizeof(double) * 4, hipMemcpyDeviceToDevice, s));
    SGP_HIP(hipStreamSynchronize(s));   // dP goes out of scope
  }
  hipLaunchKernelGGL(add_identity_kernel, dim3((unsigned)((m_pad + 255) / 256)), dim3(256), 0, s, dG.p, ldg, m_pad);
  SGP_HIP(hipMemcpy2DAsync(dB.p, sizeof(double) * m_pad, dG.p, sizeof(double) * ldg, sizeof(double) * m_pad,
                           (size_t)m_pad, hipMemcpyDeviceToDevice, s));
  CHECK_RC(launch_mirror_lower(dB.p, m_pad, m_pad, s));
  hipLaunchKernelGGL(set_row_kernel, dim3((unsigned)((m_pad + 255) / 256)), dim3(256), 0, s, dG.p, ldg, m_pad, dots.p,
                     m_pad);
  double* Je = dG.p + m_pad + TILE;
  hipLaunchKernelGGL(add_identity_kernel, dim3((unsigned)((m_pad + 255) / 256)), dim3(256), 0, s, Je, ldg, m_pad);
  SGP_HIP(hipGetLastError());
  // ---- factor 2: [B ; (A delta)' ; I]  ->  Le, b' = (Le^-1 A delta)', Je = Le^-T
  SGP_HIP(hipMemsetAsync(ctx->d_info, 0, sizeof(int), s));
  CHECK_RC(chol_bordered(ctx, dG.p, ldg, m_pad, ldg, nullptr, s, m_pad + TILE));
  CHECK_RC(launch_sum_array(ctx->d_slots, m_pad / TILE, ctx->d_scal + 5, s));
  CHECK_RC(launch_rowsumsq(dG.p + m_pad, ldg, m_pad, 1, ctx->d_scal + 6, 0, s));
  double h[6];
  SGP_HIP(hipMemcpyAsync(h, ctx->d_scal + 1, sizeof(double) * 6, hipMemcpyDeviceToHost, s));
  info = fetch_info(ctx, s);
  if (info < 0) return -3;
  if (info > 0) {
    set_error("vfe: A A' + I is not positive definite (leading minor " + std::to_string(info) + ")");
    return info;
  }
  {
    double tmp = h[0] + h[4] + h[1] - h[5];
    double dtc = -0.5 * ((double)(shard ? shard->n_total : N) * 1.8378770664093453 + tmp);
    elbo_out[0] = dtc - 0.5 * (h[2] - h[3]);
  }
  // ---- M x M stage
  CHECK_RC(launch_gemv_rows(Je, ldg, m_pad, m_pad, dG.p + m_pad, ldg, nullptr, du.p, s, 1));   // u = Le^-T b
  CHECK_RC(launch_gemm_nt_uut(Je, ldg, dBinv.p, m_pad, m_pad, s));                               // B^-1
  CHECK_RC(launch_mirror_lower(dBinv.p, m_pad, m_pad, s));
  CHECK_RC(launch_vfe_zs(dB.p, dBinv.p, du.p, dZ.p, dS.p, m_pad, s));
  CHECK_RC(launch_gemv_rows(Jm, ld, m_pad, m_pad, du.p, 1, nullptr, dut.p, s, 1));               // J u
  // ---- data-point stage: R Z, row statistics, G_xz
  CHECK_RC(launch_gemm_nt(R, ld, dZ.p, m_pad, dRZ.p, n_rows, n_rows, m_pad, m_pad, 1.0, 0.0, NOMASK, 0, 0, s));
  CHECK_RC(launch_vfe_rowstats(R, ld, dRZ.p, n_rows, du.p, ddelta.p, drsig.p, dvar.p, N, m_pad, dgy.p, dgsy.p, s));
  // E = (R Z) J' overwrites R's storage?  no: keep R intact, reuse the Gram scratch-sized dRZ -> new buffer
  DevBuf dE;
  CHECK_RC(dE.alloc((size_t)n_rows * m_pad));
  CHECK_RC(launch_gemm_nt(dRZ.p, n_rows, Jm, ld, dE.p, n_rows, n_rows, m_pad, m_pad, 1.0, 0.0, NOMASK, 0, 0, s));
  CHECK_RC(launch_vfe_gxz(dE.p, n_rows, ddelta.p, dut.p, drsig.p, n_rows, m_pad, s));
  // ---- G_zz = -1/2 J S J' (a sharded call: on the primary rank only)
  if (!shard || shard->primary) {
    CHECK_RC(launch_gemm_nt(Jm, ld, dS.p, m_pad, dT1.p, m_pad, m_pad, m_pad, m_pad, 1.0, 0.0, NOMASK, 0, 0, s));
    CHECK_RC(launch_gemm_nt(dT1.p, m_pad, Jm, ld, dGzz.p, m_pad, m_pad, m_pad, m_pad, -0.5, 0.0, NOMASK, 0, 0, s));
  }
  // ---- contractions against the flattened terms
  if (grad_coef_xz || grad_inscale_xz)
    CHECK_RC(contract_spec(gx.ds, dE.p, n_rows, nullptr, n_rows / TILE, m_pad / TILE, dpart, dgcx.p, dgsx.p, s));
  if (grad_coef_zz || grad_inscale_zz)
    CHECK_RC(contract_spec(gz.ds, dGzz.p, m_pad, nullptr, m_pad / TILE, m_pad / TILE, dpart, dgcz.p, dgsz.p, s));
  // ---- input points: zz is symmetric (twice the row side, as in logpdf_grad_core); xz is
  // rectangular: row side for the x inputs, and the transposed contraction for the z inputs
  std::vector<DevBuf> dgz(grad_inputs_zz ? zz->n_inputs : 0), dgxz(grad_inputs_xz ? xz->n_inputs : 0);
  auto zero_inputs = [&](std::vector<DevBuf>& v, const sgp_dspec* ds) -> int {
    for (size_t k = 0; k < v.size(); ++k) {
      size_t cnt = (size_t)std::max<long>(1, (long)ds->in_dim[k] * ds->in_n[k]);
      CHECK_RC(v[k].alloc(cnt));
      SGP_HIP(hipMemsetAsync(v[k].p, 0, sizeof(double) * cnt, s));
    }
    return 0;
  };
  // Function-valued scales sigma(x) * f (product.jl:25-48) ride along as in logpdf_grad_core: K_ij = coef rs_i k_ij cs_j,
  // the row-scale sums come out of the same passes (zz symmetric: "row side x 2" covers the column role through the
  // mirror term; xz: the row side gives d / d rs (scales at x), the transposed pass d / d cs (scales at z)).
  const size_t ntz_s = gz.ds->h_terms.size(), ntx_s = gx.ds->h_terms.size();
  std::vector<DevBuf> drz(grad_rowscale_zz ? ntz_s : 0), drx(grad_rowscale_xz ? ntx_s : 0), dcx(grad_colscale_xz ? ntx_s : 0);
  auto scale_buf = [&](std::vector<DevBuf>& v, double* const* want, size_t t, const double* vec, long len, double** out) -> int {
    *out = nullptr;
    if (!want || !want[t] || !vec || len <= 0) return 0;
    CHECK_RC(v[t].alloc((size_t)len));
    SGP_HIP(hipMemsetAsync(v[t].p, 0, sizeof(double) * len, s));
    *out = v[t].p;
    return 0;
  };
  if (grad_inputs_zz || grad_rowscale_zz) {
    if (grad_inputs_zz) CHECK_RC(zero_inputs(dgz, gz.ds));
    const sgp_dspec* ds = gz.ds;
    for (int I = 0; I < ds->nrb; ++I)
      for (int J = 0; J < ds->ncb; ++J) {
        if (ds->row_len[I] == 0 || ds->col_len[J] == 0) continue;
        int p = I * ds->ncb + J;
        for (int t = ds->term_ptr[p]; t < ds->term_ptr[p + 1]; ++t) {
          double* gsv = nullptr;
          CHECK_RC(scale_buf(drz, grad_rowscale_zz, t, ds->h_terms[t].rs, ds->row_len[I], &gsv));
          if (!grad_inputs_zz && !gsv) continue;
          CHECK_RC(launch_grad_inputs(dGzz.p, 1, m_pad, nullptr, ds->row_off[I], ds->row_len[I], ds->col_off[J],
                                      ds->col_len[J], ds->h_terms[t], ds->pair_dmax[p], 2.0,
                                      grad_inputs_zz ? dgz[ds->term_row_input[t]].p : nullptr, s, gsv));
        }
      }
  }
  if (grad_inputs_xz || grad_rowscale_xz || grad_colscale_xz) {
    if (grad_inputs_xz) CHECK_RC(zero_inputs(dgxz, gx.ds));
    const sgp_dspec* ds = gx.ds;
    for (int I = 0; I < ds->nrb; ++I)
      for (int J = 0; J < ds->ncb; ++J) {
        if (ds->row_len[I] == 0 || ds->col_len[J] == 0) continue;
        int p = I * ds->ncb + J;
        for (int t = ds->term_ptr[p]; t < ds->term_ptr[p + 1]; ++t) {
          const DevTerm& T = ds->h_terms[t];
          double *gsr = nullptr, *gsc = nullptr;
          CHECK_RC(scale_buf(drx, grad_rowscale_xz, t, T.rs, ds->row_len[I], &gsr));
          CHECK_RC(scale_buf(dcx, grad_colscale_xz, t, T.cs, ds->col_len[J], &gsc));
          if (grad_inputs_xz || gsr || gsc)
            if (grad_inputs_xz || gsr)
            CHECK_RC(launch_grad_inputs(dE.p, 1, n_rows, nullptr, ds->row_off[I], ds->row_len[I], ds->col_off[J],
                                        ds->col_len[J], T, ds->pair_dmax[p], 1.0,
                                        grad_inputs_xz ? dgxz[ds->term_row_input[t]].p : nullptr, s, gsr));
          if (grad_inputs_xz || gsc) {
            DevTerm Tt = T;  // the same term seen from its column points
            Tt.xr = T.xc;
            Tt.ldr = T.ldc;
            Tt.xc = T.xr;
            Tt.ldc = T.ldr;
            Tt.rs = T.cs;
            Tt.cs = T.rs;
            CHECK_RC(launch_grad_inputs(dE.p, n_rows, 1, nullptr, ds->col_off[J], ds->col_len[J], ds->row_off[I],
                                        ds->row_len[I], Tt, ds->pair_dmax[p], 1.0,
                                        grad_inputs_xz ? dgxz[ds->term_col_input[t]].p : nullptr, s, gsc));
          }
        }
      }
  }
  SGP_HIP(hipStreamSynchronize(s));
  // ---- results
  std::vector<double> hy(N), hs(N);
  SGP_HIP(hipMemcpy(hy.data(), dgy.p, sizeof(double) * N, hipMemcpyDeviceToHost));
  SGP_HIP(hipMemcpy(hs.data(), dgsy.p, sizeof(double) * N, hipMemcpyDeviceToHost));
  if (grad_y)
    for (long i = 0; i < N; ++i) grad_y[i] = hy[i];
  if (grad_mean)
    for (long i = 0; i < N; ++i) grad_mean[i] = -hy[i];
  if (grad_noise) {
    if (noise_kind == SGP_NOISE_DIAG) {
      for (long i = 0; i < N; ++i) grad_noise[i] = hs[i];
    } else {
      double acc = 0.0;
      for (long i = 0; i < N; ++i) acc += hs[i];  // fixed order
      grad_noise[0] = acc;
    }
  }
  if (grad_var_x) {
    for (long i = 0; i < N; ++i) {
      double s2 = noise_kind == SGP_NOISE_SCALAR ? noise_x[0] : noise_x[i];
      grad_var_x[i] = -0.5 / s2;
    }
  }
  if (grad_z_noise && z_noise_kind == SGP_NOISE_DENSE) {
    // dense Sigma_z: the cotangent of Kzz + Sigma_z itself, M x M (ld M) -- as sgp_logpdf_grad returns G for a dense Sigma_y
    SGP_HIP(hipMemcpy2D(grad_z_noise, sizeof(double) * M, dGzz.p, sizeof(double) * m_pad, sizeof(double) * M, (size_t)M,
                        hipMemcpyDeviceToHost));
  } else if (grad_z_noise) {
    std::vector<double> dg(M);
    SGP_HIP(hipMemcpy2D(dg.data(), sizeof(double), dGzz.p, sizeof(double) * (m_pad + 1), sizeof(double), (size_t)M,
                        hipMemcpyDeviceToHost));
    if (z_noise_kind == SGP_NOISE_DIAG) {
      for (long i = 0; i < M; ++i) grad_z_noise[i] = dg[i];
    } else {
      double acc = 0.0;
      for (long i = 0; i < M; ++i) acc += dg[i];
      grad_z_noise[0] = acc;
    }
  }
  if (grad_coef_zz) SGP_HIP(hipMemcpy(grad_coef_zz, dgcz.p, sizeof(double) * gz.ds->h_terms.size(), hipMemcpyDeviceToHost));
  if (grad_inscale_zz)
    SGP_HIP(hipMemcpy(grad_inscale_zz, dgsz.p, sizeof(double) * gz.ds->h_terms.size(), hipMemcpyDeviceToHost));
  if (grad_coef_xz) SGP_HIP(hipMemcpy(grad_coef_xz, dgcx.p, sizeof(double) * gx.ds->h_terms.size(), hipMemcpyDeviceToHost));
  if (grad_inscale_xz)
    SGP_HIP(hipMemcpy(grad_inscale_xz, dgsx.p, sizeof(double) * gx.ds->h_terms.size(), hipMemcpyDeviceToHost));
  for (size_t k = 0; k < dgz.size(); ++k)
    if (grad_inputs_zz[k] && zz->inputs[k].n > 0)
      SGP_HIP(hipMemcpy(grad_inputs_zz[k], dgz[k].p, sizeof(double) * zz->inputs[k].dim * zz->inputs[k].n,
                        hipMemcpyDeviceToHost));
  for (size_t k = 0; k < dgxz.size(); ++k)
    if (grad_inputs_xz[k] && xz->inputs[k].n > 0)
      SGP_HIP(hipMemcpy(grad_inputs_xz[k], dgxz[k].p, sizeof(double) * xz->inputs[k].dim * xz->inputs[k].n,
                        hipMemcpyDeviceToHost));
  auto scale_out = [&](std::vector<DevBuf>& v, double* const* want, const sgp_dspec* ds, bool cols) -> int {
    if (!want) return 0;
    for (int I = 0; I < ds->nrb; ++I)
      for (int J = 0; J < ds->ncb; ++J)
        for (int t = ds->term_ptr[I * ds->ncb + J]; t < ds->term_ptr[I * ds->ncb + J + 1]; ++t)
          if (want[t] && v[t].p)
            SGP_HIP(hipMemcpy(want[t], v[t].p, sizeof(double) * (cols ? ds->col_len[J] : ds->row_len[I]),
                              hipMemcpyDeviceToHost));
    return 0;
  };
  CHECK_RC(scale_out(drz, grad_rowscale_zz, gz.ds, false));
  CHECK_RC(scale_out(drx, grad_rowscale_xz, gx.ds, false));
  CHECK_RC(scale_out(dcx, grad_colscale_xz, gx.ds, true));
  return 0;
}

// the three entry points: one record of arguments; a multi-GPU context shards the data points over its ranks (round 6)
static int elbo_grad_entry(sgp_ctx* ctx, const sgp::ElboGradArgs& a) {
  CHECK_ARG(ctx, "sgp_elbo_grad: NULL context");
  if (ctx->multi && ctx->multi_nranks > 1) return sgp_multi_elbo_grad(ctx, a);
  return sgp::drv_elbo_grad(ctx, a, nullptr);
}
#define SGP_ELBO_GRAD_ARGS(a)                                                                                      \
  sgp::ElboGradArgs a;                                                                                             \
  a.zz = zz, a.xz = xz, a.var_x = var_x, a.mean_x = mean_x, a.noise_kind = noise_kind, a.noise_x = noise_x;       \
  a.z_noise_kind = z_noise_kind, a.z_noise = z_noise, a.y = y, a.elbo_out = elbo_out, a.grad_y = grad_y;          \
  a.grad_mean = grad_mean, a.grad_noise = grad_noise, a.grad_var_x = grad_var_x, a.grad_z_noise = grad_z_noise;   \
  a.grad_coef_zz = grad_coef_zz, a.grad_inscale_zz = grad_inscale_zz, a.grad_coef_xz = grad_coef_xz;              \
  a.grad_inscale_xz = grad_inscale_xz

extern "C" int sgp_elbo_grad(sgp_ctx* ctx, const sgp_cov_spec* zz, const sgp_cov_spec* xz, const double* var_x,
                             const double* mean_x, int noise_kind, const double* noise_x, int z_noise_kind,
                             const double* z_noise, const double* y, double* elbo_out, double* grad_y,
                             double* grad_mean, double* grad_noise, double* grad_var_x, double* grad_z_noise,
                             double* grad_coef_zz, double* grad_inscale_zz, double* grad_coef_xz,
                             double* grad_inscale_xz) {
  SGP_ELBO_GRAD_ARGS(a);
  return elbo_grad_entry(ctx, a);
}

extern "C" int sgp_elbo_grad_x(sgp_ctx* ctx, const sgp_cov_spec* zz, const sgp_cov_spec* xz, const double* var_x,
                               const double* mean_x, int noise_kind, const double* noise_x, int z_noise_kind,
                               const double* z_noise, const double* y, double* elbo_out, double* grad_y,
                               double* grad_mean, double* grad_noise, double* grad_var_x, double* grad_z_noise,
                               double* grad_coef_zz, double* grad_inscale_zz, double* grad_coef_xz,
                               double* grad_inscale_xz, double* const* grad_inputs_zz,
                               double* const* grad_inputs_xz) {
  CHECK_ARG(grad_inputs_zz && grad_inputs_xz, "sgp_elbo_grad_x: grad_inputs_* is NULL");
  SGP_ELBO_GRAD_ARGS(a);
  a.grad_inputs_zz = grad_inputs_zz, a.grad_inputs_xz = grad_inputs_xz;
  return elbo_grad_entry(ctx, a);
}

extern "C" int sgp_elbo_grad_xs(sgp_ctx* ctx, const sgp_cov_spec* zz, const sgp_cov_spec* xz, const double* var_x,
                                const double* mean_x, int noise_kind, const double* noise_x, int z_noise_kind,
                                const double* z_noise, const double* y, double* elbo_out, double* grad_y,
                                double* grad_mean, double* grad_noise, double* grad_var_x, double* grad_z_noise,
                                double* grad_coef_zz, double* grad_inscale_zz, double* grad_coef_xz,
                                double* grad_inscale_xz, double* const* grad_inputs_zz,
                                double* const* grad_inputs_xz, double* const* grad_rowscale_zz,
                                double* const* grad_rowscale_xz, double* const* grad_colscale_xz) {
  SGP_ELBO_GRAD_ARGS(a);
  a.grad_inputs_zz = grad_inputs_zz, a.grad_inputs_xz = grad_inputs_xz, a.grad_rowscale_zz = grad_rowscale_zz;
  a.grad_rowscale_xz = grad_rowscale_xz, a.grad_colscale_xz = grad_colscale_xz;
  return elbo_grad_entry(ctx, a);
}
#undef SGP_ELBO_GRAD_ARGS

// sum_i w[i] d var_i / d theta over the diagonal of `spec` (the blocks (I, I) kernelmatrix_diag reads)
static int diag_grad_core(sgp_ctx* ctx, const sgp_cov_spec* spec, const double* w, double* grad_coef,
                          double* grad_inscale, double* const* grad_inputs, double* const* grad_rowscale = nullptr,
                          double* const* grad_colscale = nullptr) {
  CHECK_ARG(ctx && spec && w && grad_coef && grad_inscale, "sgp_kernelmatrix_diag_grad: NULL argument");
  CtxScope scope(ctx);
  SpecGuard g;
  CHECK_RC(dspec_create(ctx, spec, &g.ds));
  const sgp_dspec* ds = g.ds;
  CHECK_ARG(ds->nrb == ds->ncb, "kernelmatrix_diag_grad: row / col block counts differ");
  size_t nt = ds->h_terms.size();
  for (size_t t = 0; t < nt; ++t) grad_coef[t] = grad_inscale[t] = 0.0;
  if (ds->N == 0 || nt == 0) return 0;
  DevBuf dw, dgc, dgs;
  CHECK_RC(dw.upload(w, ds->N));
  CHECK_RC(dgc.alloc(nt));
  CHECK_RC(dgs.alloc(nt));
  hipStream_t s = ctx->stream;
  SGP_HIP(hipMemsetAsync(dgc.p, 0, sizeof(double) * nt, s));
  SGP_HIP(hipMemsetAsync(dgs.p, 0, sizeof(double) * nt, s));
  for (int I = 0; I < ds->nrb; ++I) {
    CHECK_ARG(ds->row_len[I] == ds->col_len[I], "kernelmatrix_diag_grad: block lengths differ");
    int p = I * ds->ncb + I;
    int t0 = ds->term_ptr[p], t1 = ds->term_ptr[p + 1];
    if (ds->row_len[I] == 0 || t1 == t0) continue;
    CHECK_RC(launch_diag_grad(dw.p + ds->row_off[I], ds->row_len[I], ds->d_terms + t0, t1 - t0, dgc.p + t0,
                              dgs.p + t0, s));
  }
  std::vector<DevBuf> dgx(grad_inputs ? spec->n_inputs : 0);
  if (grad_inputs) {
    for (int k = 0; k < spec->n_inputs; ++k) {
      size_t cnt = (size_t)std::max<long>(1, (long)ds->in_dim[k] * ds->in_n[k]);
      CHECK_RC(dgx[k].alloc(cnt));
      SGP_HIP(hipMemsetAsync(dgx[k].p, 0, sizeof(double) * cnt, s));
    }
    for (int I = 0; I < ds->nrb; ++I) {
      int p = I * ds->ncb + I;
      if (ds->row_len[I] == 0) continue;
      for (int t = ds->term_ptr[p]; t < ds->term_ptr[p + 1]; ++t)
        CHECK_RC(launch_diag_grad_inputs(dw.p + ds->row_off[I], ds->row_len[I], ds->h_terms[t],
                                         dgx[ds->term_row_input[t]].p, dgx[ds->term_col_input[t]].p, s));
    }
  }
  std::vector<DevBuf> drs(grad_rowscale ? nt : 0), dcs(grad_colscale ? nt : 0);
  if (grad_rowscale || grad_colscale)
    for (int I = 0; I < ds->nrb; ++I) {
      const long len = ds->row_len[I];
      if (len == 0) continue;
      for (int t = ds->term_ptr[I * ds->ncb + I]; t < ds->term_ptr[I * ds->ncb + I + 1]; ++t) {
        const DevTerm& T = ds->h_terms[t];
        double *o_r = nullptr, *o_c = nullptr;
        if (grad_rowscale && grad_rowscale[t] && T.rs) {
          CHECK_RC(drs[t].alloc((size_t)len));
          SGP_HIP(hipMemsetAsync(drs[t].p, 0, sizeof(double) * len, s));
          o_r = drs[t].p;
        }
        if (grad_colscale && grad_colscale[t] && T.cs) {
          CHECK_RC(dcs[t].alloc((size_t)len));
          SGP_HIP(hipMemsetAsync(dcs[t].p, 0, sizeof(double) * len, s));
          o_c = dcs[t].p;
        }
        if (!o_r && !o_c) continue;
        CHECK_RC(launch_diag_scale_grad(dw.p + ds->row_off[I], len, T, o_r, o_c, s));
      }
    }
  SGP_HIP(hipStreamSynchronize(s));
  SGP_HIP(hipMemcpy(grad_coef, dgc.p, sizeof(double) * nt, hipMemcpyDeviceToHost));
  SGP_HIP(hipMemcpy(grad_inscale, dgs.p, sizeof(double) * nt, hipMemcpyDeviceToHost));
  for (int I = 0; I < ds->nrb && (grad_rowscale || grad_colscale); ++I)
    for (int t = ds->term_ptr[I * ds->ncb + I]; t < ds->term_ptr[I * ds->ncb + I + 1]; ++t) {
      if (grad_rowscale && grad_rowscale[t] && drs[t].p)
        SGP_HIP(hipMemcpy(grad_rowscale[t], drs[t].p, sizeof(double) * ds->row_len[I], hipMemcpyDeviceToHost));
      if (grad_colscale && grad_colscale[t] && dcs[t].p)
        SGP_HIP(hipMemcpy(grad_colscale[t], dcs[t].p, sizeof(double) * ds->row_len[I], hipMemcpyDeviceToHost));
    }
  for (size_t k = 0; k < dgx.size(); ++k)
    if (grad_inputs[k] && spec->inputs[k].n > 0)
      SGP_HIP(hipMemcpy(grad_inputs[k], dgx[k].p, sizeof(double) * spec->inputs[k].dim * spec->inputs[k].n,
                        hipMemcpyDeviceToHost));
  return 0;
}

extern "C" int sgp_kernelmatrix_diag_grad(sgp_ctx* ctx, const sgp_cov_spec* spec, const double* w,
                                          double* grad_coef, double* grad_inscale) {
  return diag_grad_core(ctx, spec, w, grad_coef, grad_inscale, nullptr);
}

extern "C" int sgp_kernelmatrix_diag_grad_x(sgp_ctx* ctx, const sgp_cov_spec* spec, const double* w,
                                            double* grad_coef, double* grad_inscale,
                                            double* const* grad_inputs) {
  CHECK_ARG(grad_inputs != nullptr, "sgp_kernelmatrix_diag_grad_x: grad_inputs is NULL");
  return diag_grad_core(ctx, spec, w, grad_coef, grad_inscale, grad_inputs);
}

extern "C" int sgp_kernelmatrix_diag_grad_xs(sgp_ctx* ctx, const sgp_cov_spec* spec, const double* w,
                                             double* grad_coef, double* grad_inscale, double* const* grad_inputs,
                                             double* const* grad_rowscale, double* const* grad_colscale) {
  return diag_grad_core(ctx, spec, w, grad_coef, grad_inscale, grad_inputs, grad_rowscale, grad_colscale);
}

static int sgp_sparse_posterior_create_impl(sgp_ctx* ctx, const sgp_cov_spec* zz,
                                           const sgp_cov_spec* xz, const double* mean_x,
                                           int noise_kind, const double* noise_x, int z_noise_kind,
                                           const double* z_noise, const double* y,
                                           sgp_sparse_post** out) {
  CHECK_ARG(ctx && zz && xz && noise_x && z_noise && y && out, "sgp_sparse_posterior_create: NULL argument");
  CtxScope scope(ctx);
  sgp_sparse_post* p = new sgp_sparse_post();
  double h[6];
  if (ctx->multi && ctx->multi_nranks > 1) {
    // data points sharded over the ranks; the M x M factors of the posterior are kept on devices[0]
    const long M = spec_rows_host(zz), m_pad = rup(M, TILE);
    p->ctx = ctx;
    p->ctx_serial = ctx->serial;
    p->M = M;
    p->m_pad = m_pad;
    p->ldg = m_pad + TILE;
    int rc = 0;
    if (hipMalloc(&p->dLz, sizeof(double) * m_pad * m_pad) != hipSuccess ||
        hipMalloc(&p->d_wz, sizeof(double) * (m_pad / TILE) * INVD_STRIDE) != hipSuccess ||
        hipMalloc(&p->dG, sizeof(double) * vfe_part_len(m_pad)) != hipSuccess ||
        hipMalloc(&p->d_wg, sizeof(double) * (m_pad / TILE) * INVD_STRIDE) != hipSuccess) {
      (void)hipGetLastError();
      set_error("sgp_sparse_posterior_create: hipMalloc failed");
      rc = -2;
    }
    if (!rc)
      rc = sgp_multi_vfe(ctx, zz, xz, nullptr, mean_x, noise_kind, noise_x, z_noise_kind, z_noise, y, h, p->dLz, p->d_wz,
                         p->dG, p->d_wg);
    if (rc) {
      sgp_sparse_posterior_destroy(p);
      return rc;
    }
    *out = p;
    return 0;
  }
  int rc = vfe_pipeline(ctx, zz, xz, nullptr, mean_x, noise_kind, noise_x, z_noise_kind, z_noise, y,
                        h, p);
  if (rc) {
    sgp_sparse_posterior_destroy(p);
    return rc;
  }
  *out = p;
  return 0;
}
extern "C" int sgp_sparse_posterior_create(sgp_ctx* ctx, const sgp_cov_spec* zz,
                                           const sgp_cov_spec* xz, const double* mean_x,
                                           int noise_kind, const double* noise_x, int z_noise_kind,
                                           const double* z_noise, const double* y,
                                           sgp_sparse_post** out) {
  return with_df_fallback(ctx, [&]() { return sgp_sparse_posterior_create_impl(ctx, zz, xz, mean_x, noise_kind, noise_x, z_noise_kind, z_noise, y, out); });
}

extern "C" int sgp_sparse_posterior_predict(sgp_sparse_post* post, const sgp_cov_spec* cross,
                                            const sgp_cov_spec* prior_ss, const double* mean_s,
                                            double* mean_out, double* var_out, double* cov_out,
                                            int64_t ldcov) {
  CHECK_ARG(post && cross, "sgp_sparse_posterior_predict: NULL argument");
  sgp_ctx* ctx = post->ctx;
  CHECK_ARG(ctx_is_live(ctx, post->ctx_serial),
            "sgp_sparse_posterior_predict: the context this posterior was created on has been destroyed");
  CtxScope scope(ctx);
  SpecGuard gc, gp;
  CHECK_RC(dspec_create(ctx, cross, &gc.ds));
  if (prior_ss) CHECK_RC(dspec_create(ctx, prior_ss, &gp.ds));
  long Ns = gc.ds->N;
  CHECK_ARG(gc.ds->M == post->M, "sparse predict: cross spec columns != number of inducing points");
  CHECK_ARG(!gp.ds || gp.ds->N == Ns, "sparse predict: prior_ss size != number of x*");
  CHECK_ARG(!cov_out || ldcov >= Ns, "sparse predict: ldcov < Ns");
  if (Ns == 0) return 0;
  long ns_pad = rup(Ns, TILE), m_pad = post->m_pad;
  hipStream_t s = ctx->stream;
  DevBuf dB, dB2, dms;
  CHECK_RC(dB.alloc((size_t)ns_pad * m_pad));
  CHECK_RC(dB2.alloc((size_t)ns_pad * m_pad));
  if (mean_s) CHECK_RC(dms.upload(mean_s, Ns));
  SGP_HIP(hipMemsetAsync(dB.p, 0, sizeof(double) * ns_pad * m_pad, s));
  CHECK_RC(assemble(gc.ds, dB.p, ns_pad, 0, ns_pad / TILE, 0, m_pad / TILE, 0, -1, 0.0, nullptr, s));
  // B' = K(x*, z) Lz^-T ; then C' = B' Le^-T
  CHECK_RC(row_trsm(ctx, dB.p, ns_pad, ns_pad, post->dLz, m_pad, post->d_wz, m_pad, s));
  SGP_HIP(hipMemcpyAsync(dB2.p, dB.p, sizeof(double) * ns_pad * m_pad, hipMemcpyDeviceToDevice, s));
  CHECK_RC(row_trsm(ctx, dB2.p, ns_pad, ns_pad, post->dG, post->ldg, post->d_wg, m_pad, s));
  // mean* = m* + K*z alpha, alpha = Lz^-T Le^-T (Le^-1 A delta)  =>  mean* = m* + C' (Le^-1 A delta)
  // var*  = k** - |B|^2 + |C|^2
  if (mean_out)
    CHECK_RC(predict_common(ctx, gc.ds, gp.ds, mean_s ? dms.p : nullptr, Ns, ns_pad, dB2.p, m_pad,
                            post->dG + m_pad, post->ldg, post->M, mean_out, nullptr, nullptr, 0,
                            0.0, nullptr, s));
  if (var_out || cov_out)
    CHECK_RC(predict_common(ctx, gc.ds, gp.ds, nullptr, Ns, ns_pad, dB.p, m_pad, post->dG + m_pad,
                            post->ldg, post->M, nullptr, var_out, cov_out, ldcov, +1.0, dB2.p, s));
  return 0;
}

// ---------------------------------------------------------------------------------------
// multi-GPU building blocks (device pointers)
// ---------------------------------------------------------------------------------------
extern "C" int sgp_dev_assemble_cols(sgp_ctx* ctx, const sgp_dspec* ds, int64_t N, int64_t c0,
                                     int64_t nc, double* d_dst, int64_t ldd, int64_t m_tot,
                                     const double* d_mean, int noise_kind, const double* noise_host,
                                     const double* d_noise, const double* d_Y, int64_t ldy,
                                     int64_t ncols, void* stream) {
  CHECK_ARG(ctx && ds && d_dst, "sgp_dev_assemble_cols: NULL argument");
  CHECK_ARG(ds->symmetric && ds->N == N, "sgp_dev_assemble_cols: spec must be symmetric of size N");
  CHECK_ARG(c0 % TILE == 0 && nc % TILE == 0, "sgp_dev_assemble_cols: c0, nc must be multiples of 128");
  CHECK_ARG(noise_kind == SGP_NOISE_SCALAR || noise_kind == SGP_NOISE_DIAG, "bad noise kind");
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);  // ctx scratch (logdet slots, inverse blocks) is shared
  SGP_HIP(hipSetDevice(ctx->device));
  hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
  int64_t n_pad, mt;
  sgp_geometry(N, ncols, &n_pad, &mt);
  // ldd >= m_tot - c0: a panel may be stored packed (rows c0 .. m_tot only); the caller then passes the
  // virtual address of global row 0 (first stored row minus c0), which is never dereferenced above row c0
  // m_tot > mt: the caller appends more bordered rows below the observation rows (sgp_dev_assemble_cross_rows)
  CHECK_ARG(mt <= m_tot && m_tot % TILE == 0 && ldd >= m_tot - c0, "sgp_dev_assemble_cols: geometry mismatch");
  double* Kv = d_dst - c0 * ldd;  // virtual base: global column index
  double s2 = noise_host ? noise_host[0] : 0.0;
  CHECK_RC(assemble(ds, Kv, ldd, c0 / TILE, n_pad / TILE, c0 / TILE, (c0 + nc) / TILE, 1, noise_kind,
                    s2, d_noise, s));
  CHECK_RC(launch_fill_pad(d_dst, ldd, N, n_pad, c0, nc, m_tot, c0, s));
  CHECK_RC(launch_border_rows(d_dst, ldd, n_pad, N, c0, nc, d_Y, ldy, ncols, d_mean, s));
  return 0;
}

// Sharded posterior (AbstractGPs.posterior + mean / var / cov of the PosteriorGP [EXT], App. A.5):
// K(x*, x) rides through the sharded factorisation as extra bordered rows and comes out as V' = K(x*, x) L^-T,
// next to the observation row z' = (L^-1 (y - m))'; then  mean* = m* + V' z,  var* = k** - rowsumsq(V'),
// cov* = K** - V' V  are sums over columns = over ranks: one all-reduce of 2 n* (+ n*^2) doubles.
extern "C" int sgp_dev_assemble_cross_rows(sgp_ctx* ctx, const sgp_dspec* cross, int64_t c0, int64_t nc,
                                           double* d_dst, int64_t ldd, int64_t row0, void* stream) {
  CHECK_ARG(ctx && cross && d_dst, "sgp_dev_assemble_cross_rows: NULL argument");
  CHECK_ARG(!cross->symmetric, "sgp_dev_assemble_cross_rows: needs a cross-covariance spec (rows x*, columns x)");
  CHECK_ARG(c0 % TILE == 0 && nc % TILE == 0 && row0 % TILE == 0 && row0 >= c0 + nc,
            "sgp_dev_assemble_cross_rows: c0, nc, row0 must be multiples of 128, rows below the square part");
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);
  SGP_HIP(hipSetDevice(ctx->device));
  hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
  const long ns = cross->N, Nx = cross->M;
  const long ns_pad = (ns + TILE - 1) / TILE * TILE;
  CHECK_ARG(ldd >= row0 + ns_pad - c0, "sgp_dev_assemble_cross_rows: leading dimension too small");
  const long nc_valid = std::max<long>(0, std::min<long>(nc, Nx - c0));   // columns >= Nx are identity padding
  if (nc_valid > 0) {
    double* Kv = d_dst + row0 - c0 * ldd;   // element (r, c) of K(x*, x) at Kv[r + c * ldd]
    CHECK_RC(assemble(cross, Kv, ldd, 0, ns_pad / TILE, c0 / TILE, (c0 + nc) / TILE, 0, -1, 0.0, nullptr, s));
    CHECK_RC(launch_zero_rows(d_dst, ldd, row0 + ns, row0 + ns_pad, nc_valid, s));
  }
  if (nc_valid < nc) CHECK_RC(launch_zero_rows(d_dst + nc_valid * ldd, ldd, row0, row0 + ns_pad, nc - nc_valid, s));
  return 0;
}

extern "C" int sgp_dev_rows_dot(sgp_ctx* ctx, const double* d_rows, int64_t ld, int64_t nrows, int64_t nc,
                                const double* d_zrow, double* d_sumsq, double* d_dot, void* stream) {
  CHECK_ARG(ctx && d_rows && d_zrow && d_sumsq && d_dot, "sgp_dev_rows_dot: NULL argument");
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);
  SGP_HIP(hipSetDevice(ctx->device));
  hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
  return launch_rows_dot(d_rows, ld, nrows, nc, d_zrow, d_sumsq, d_dot, s);
}

// G (nrows_pad x nrows_pad, ld ldg, all tiles) += R R' over the nc columns of the bordered rows R
extern "C" int sgp_dev_rows_gram(sgp_ctx* ctx, const double* d_rows, int64_t ld, int64_t nrows_pad, int64_t nc,
                                 double* d_G, int64_t ldg, void* stream) {
  CHECK_ARG(ctx && d_rows && d_G, "sgp_dev_rows_gram: NULL argument");
  CHECK_ARG(nrows_pad % TILE == 0 && nc % 16 == 0 && ldg >= nrows_pad, "sgp_dev_rows_gram: bad sizes");
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);
  SGP_HIP(hipSetDevice(ctx->device));
  hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
  return launch_gemm_nt(d_rows, ld, d_rows, ld, d_G, ldg, nrows_pad, nrows_pad, nc, 1.0, 1.0, NOMASK, 0, 0, s);
}

extern "C" int sgp_dev_panel_factor(sgp_ctx* ctx, double* d_P, int64_t ld, int64_t m, int64_t w,
                                    int64_t g0, double* d_logdet, int* d_info, void* stream) {
  CHECK_ARG(ctx && d_P && d_logdet && d_info, "sgp_dev_panel_factor: NULL argument");
  CHECK_ARG(w % TILE == 0 && m % TILE == 0 && m >= w, "sgp_dev_panel_factor: bad sizes");
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);  // ctx scratch (logdet slots, inverse blocks) is shared
  SGP_HIP(hipSetDevice(ctx->device));
  hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
  // per-block logdet slots live in ctx scratch; accumulate their sum into d_logdet[0]
  FuseScope fuse_scope(ctx, fuse_mode(ctx, m));   // inner fused launches by the panel's height
  CHECK_RC(panel_factor(ctx, d_P, ld, m, w, g0, ctx->d_slots, d_info, nullptr, s));
  CHECK_RC(launch_sum_array(ctx->d_slots, w / TILE, ctx->d_scal + 8, s));
  hipLaunchKernelGGL(accum_kernel, dim3(1), dim3(1), 0, s, d_logdet, ctx->d_scal + 8);
  SGP_HIP(hipGetLastError());
  return 0;
}

extern "C" int sgp_dev_panel_update_batch(sgp_ctx* ctx, const sgp_panel_src* srcs, int nsrc, const sgp_panel_dst* dsts,
                                          int ndst, int64_t m_tot, void* stream) {
  CHECK_ARG(ctx && (srcs || nsrc == 0) && (dsts || ndst == 0), "sgp_dev_panel_update_batch: NULL argument");
  CHECK_ARG(nsrc >= 0 && nsrc <= SEG_MAX_SRC && ndst >= 0, "sgp_dev_panel_update_batch: at most 8 source panels");
  if (ndst == 0 || nsrc == 0) return 0;
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);
  SGP_HIP(hipSetDevice(ctx->device));
  hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
  SegBatch b;
  b.m_tot = m_tot;
  for (int q = 0; q < nsrc; ++q) {
    CHECK_ARG(srcs[q].base && srcs[q].w > 0 && srcs[q].w % 16 == 0 && srcs[q].w < (1 << 30),
              "sgp_dev_panel_update_batch: bad source panel");
    b.src[q] = SegSrc{srcs[q].base, (long)srcs[q].ld, (long)srcs[q].row0, (int)srcs[q].w};
  }
  for (int q = nsrc; q < SEG_MAX_SRC; ++q) b.src[q] = SegSrc{nullptr, 0, 0, 0};
  for (int d0 = 0; d0 < ndst; d0 += SEG_MAX_DST) {
    b.n_dst = 0;
    for (int d = d0; d < std::min(ndst, d0 + SEG_MAX_DST); ++d) {
      const sgp_panel_dst& D = dsts[d];
      CHECK_ARG(D.base && D.src_first >= 0 && D.src_count >= 0 && D.src_first + D.src_count <= nsrc && D.w < (1 << 30),
                "sgp_dev_panel_update_batch: bad destination panel");
      if (D.src_count == 0) continue;
      b.dst[b.n_dst++] = SegDst{D.base, (long)D.ld, (long)D.c0, (int)D.w, D.src_first, D.src_count, 0u};
    }
    CHECK_RC(launch_gemm_nt_seg(b, s));
  }
  return 0;
}

extern "C" int sgp_dev_panel_update(sgp_ctx* ctx, const double* d_P, int64_t ldp, int64_t p_row0,
                                    int64_t w, double* d_C, int64_t ldc, int64_t c0, int64_t nc,
                                    int64_t m_tot, void* stream) {
  CHECK_ARG(ctx && d_P && d_C, "sgp_dev_panel_update: NULL argument");
  CHECK_ARG(c0 % TILE == 0 && nc % TILE == 0 && w % 16 == 0 && c0 >= p_row0, "sgp_dev_panel_update: bad sizes");
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);  // ctx scratch (logdet slots, inverse blocks) is shared
  SGP_HIP(hipSetDevice(ctx->device));
  hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
  const double* A = d_P + (c0 - p_row0);  // panel rows c0.. (global)
  // d_C column 0 == global column c0; rows are global
  CHECK_RC(launch_gemm_nt_update(A, ldp, d_C + c0, ldc, m_tot - c0, nc, w, s));
  return 0;
}

extern "C" int sgp_dev_rowsumsq(sgp_ctx* ctx, const double* d_rows, int64_t ld, int64_t nc,
                                int64_t nrows, double* d_out, void* stream) {
  CHECK_ARG(ctx && d_rows && d_out, "sgp_dev_rowsumsq: NULL argument");
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);  // ctx scratch (logdet slots, inverse blocks) is shared
  SGP_HIP(hipSetDevice(ctx->device));
  hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
  return launch_rowsumsq(d_rows, ld, nc, nrows, d_out, 1, s);
}

// ---------------------------------------------------------------------------------------
// driver.h: the routines above for the multi-GPU driver (multi.hip)
// ---------------------------------------------------------------------------------------
__global__ void axpy_block_kernel(double* C, long ldc, const double* S, long lds, long nr, long nc, double a) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= nr * nc) return;
  const long r = idx % nr, c = idx / nr;
  C[r + c * ldc] += a * S[r + c * lds];
}

namespace sgp {
int drv_assemble(const sgp_dspec* ds, double* Kv, long ld, long tile_r_lo, long tile_r_hi, long tile_c_lo,
                 long tile_c_hi, int lower_only, int noise_kind, double sigma2, const double* d_noise_diag,
                 hipStream_t s) {
  return assemble(ds, Kv, ld, tile_r_lo, tile_r_hi, tile_c_lo, tile_c_hi, lower_only, noise_kind, sigma2,
                  d_noise_diag, s);
}
// Factor a PACKED panel of the sharded factorisation (w columns over m rows, element [0] = (row g0, column g0) of the matrix).
// df != 0 (round 6: the hybrid schedule carried into the sharded sweep): ONE launch of the dataflow kernel -- the diagonal
// chain and the row solves below it as tile tasks, the pattern d_nz read at the panel's offset; px (optional): factor only
// the first px->n_fact columns and update the others with them, external source panels applied first (chol_df.hip).
// df == 0: the launch-based chain of rounds 2 - 5 (px must be empty; the caller issues the updates as launches of its own).
int drv_panel_factor(sgp_ctx* ctx, double* P, long ld, long m, long w, long g0, double* d_logdet, int* d_info,
                     double* d_invstore, hipStream_t s, int df, const sz_word* d_nz, int nz_words, const DfPanel* px, int lean) {
  CHECK_ARG(w % TILE == 0 && m % TILE == 0 && m >= w, "drv_panel_factor: bad sizes");
  long n_fact = w;
  if (df) {
    if (px) n_fact = px->n_fact;
    CHECK_ARG(n_fact / TILE <= ctx->n_slots, "drv_panel_factor: panel too wide for the logdet slot buffer");
    CHECK_RC(df_scratch(ctx, m, 1, d_invstore ? 0 : n_fact, s));
    // lean: two workgroups per CU (the panel kernel of ranks that SHARE a GPU: it must fit beside other ranks' update launches)
    CHECK_RC(launch_chol_dataflow(P, ld, w, m, ctx->d_df_state, d_invstore ? d_invstore : ctx->d_df_inv, ctx->d_slots, d_info,
                                  lean ? ctx->df_wgs : ctx->hybrid_wgs, ctx->df_timeout_s, s, nullptr, nullptr,
                                  lean ? 0 : ctx->hybrid_fat, d_nz, nz_words, g0, px));
  } else {
    CHECK_ARG(!px || (px->n_ext == 0 && px->n_fact == w), "drv_panel_factor: the launch-based chain takes whole panels only");
    FuseScope fuse_scope(ctx, fuse_mode(ctx, m));
    if (d_nz) gemm_set_structure(P, ld, d_nz, nz_words, w, nullptr, 0, g0 / TILE);
    const int rc = panel_factor(ctx, P, ld, m, w, g0, ctx->d_slots, d_info, d_invstore, s);
    if (d_nz) gemm_set_structure(nullptr, 0, nullptr, 0);
    CHECK_RC(rc);
  }
  CHECK_RC(launch_sum_array(ctx->d_slots, n_fact / TILE, ctx->d_scal + 8, s));
  hipLaunchKernelGGL(accum_kernel, dim3(1), dim3(1), 0, s, d_logdet, ctx->d_scal + 8);
  SGP_HIP(hipGetLastError());
  return 0;
}
int drv_row_trsm(sgp_ctx* ctx, double* R, long ldr, long nrows, const double* L, long ldl, const double* d_invall,
                 long n, hipStream_t s) {
  return row_trsm(ctx, R, ldr, nrows, L, ldl, d_invall, n, s);
}
int drv_back_substitute(const double* Lv, long ld, const double* wall_v, long k_first, long k_last, long n_end,
                        double* d_z, double* d_alpha, hipStream_t s) {
  return back_substitute_range(Lv, ld, wall_v, k_first, k_last, n_end, d_z, d_alpha, s);
}
int drv_diag_of_spec(sgp_ctx* ctx, const sgp_dspec* ds, double* d_out, hipStream_t s) {
  return diag_of_spec(ctx, ds, d_out, s);
}
int drv_dspec_create(sgp_ctx* ctx, const sgp_cov_spec* sp, sgp_dspec** out) { return dspec_create(ctx, sp, out); }
void drv_dspec_free(sgp_dspec* ds) { dspec_free(ds); }
long drv_invd_stride() { return INVD_STRIDE; }
int drv_copy_strided(const double* src, long stride, long n, double* dst, hipStream_t s) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(copy_strided_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, stride, n, dst);
  SGP_HIP(hipGetLastError());
  return 0;
}
int drv_axpy_block(double* C, long ldc, const double* S, long lds, long nr, long nc, double a, hipStream_t s) {
  if (nr <= 0 || nc <= 0) return 0;
  hipLaunchKernelGGL(axpy_block_kernel, dim3((unsigned)((nr * nc + 255) / 256)), dim3(256), 0, s, C, ldc, S, lds, nr, nc,
                     a);
  SGP_HIP(hipGetLastError());
  return 0;
}
int drv_fill_mean_cols(double* dst, long ld, long nrows, long ncols, long N, const double* mean, hipStream_t s) {
  if (nrows <= 0 || ncols <= 0) return 0;
  hipLaunchKernelGGL(fill_mean_cols_kernel, dim3((unsigned)((nrows * ncols + 255) / 256)), dim3(256), 0, s, dst, ld,
                     nrows, ncols, N, mean);
  SGP_HIP(hipGetLastError());
  return 0;
}
long drv_vfe_part_len(long m_pad) { return vfe_part_len(m_pad); }
int drv_vfe_partial(sgp_ctx* ctx, const sgp_cov_spec* zz, const sgp_cov_spec* xz, const double* var_x,
                    const double* mean_x, int noise_kind, const double* noise_x, int z_noise_kind, const double* z_noise,
                    const double* y, double* dLz, double* d_wz, double* d_part, long part_len) {
  return elbo_partial_core(ctx, zz, xz, var_x, mean_x, noise_kind, noise_x, z_noise_kind, z_noise, y, dLz, d_wz, d_part,
                           part_len);
}
int drv_elbo_grad(sgp_ctx* ctx, const ElboGradArgs& a, const ElboGradShard* shard) {
  auto run = [&]() {
    return elbo_grad_core(ctx, a.zz, a.xz, a.var_x, a.mean_x, a.noise_kind, a.noise_x, a.z_noise_kind, a.z_noise, a.y,
                          a.elbo_out, a.grad_y, a.grad_mean, a.grad_noise, a.grad_var_x, a.grad_z_noise, a.grad_coef_zz,
                          a.grad_inscale_zz, a.grad_coef_xz, a.grad_inscale_xz, a.grad_inputs_zz, a.grad_inputs_xz,
                          a.grad_rowscale_zz, a.grad_rowscale_xz, a.grad_colscale_xz, shard);
  };
  // (a rank of a sharded call cannot rerun on its own -- the reduction is collective: multi.hip reruns all of them)
  return shard ? run() : with_df_fallback(ctx, run);
}
int drv_vfe_finish(sgp_ctx* ctx, long M, double* d_part, double* d_wg, double* h6) {
  return elbo_finish_core(ctx, M, d_part, d_wg, h6);
}
// the multi-GPU driver's view (multi.hip): rank 0's context computes the pattern, every rank uploads it
int drv_sz_pattern(sgp_ctx* ctx, const sgp_dspec* ds, int noise_kind, long n_pad, long m_tot, int* words) {
  return ::sz_pattern(ctx, ds, noise_kind, n_pad, m_tot, words);
}
int drv_sz_upload(sgp_ctx* ctx, const sgp_ctx* from, int words, hipStream_t s, const sz_word** d_nz) {
  SzMask m;
  CHECK_RC(::sz_upload(ctx, from->h_sz, words, s, &m));
  *d_nz = m.d_nz;
  return 0;
}
// fraction of the lower tiles of destination columns [c0, c0 + w) (rows c0 .. m_tot) that the k tiles [kt0, kt1) touch
double drv_sz_live_fraction(const sgp_ctx* ctx, int words, long c0, long w, long m_tot, long kt0, long kt1) {
  if (words <= 0) return 1.0;
  const long t0 = c0 / TILE, n_tc = w / TILE, n_tr = (m_tot - c0) / TILE;
  double live = 0, all = 0;
  for (long tc = 0; tc < n_tc; ++tc)
    for (long tr = tc; tr < n_tr; ++tr) {
      const sz_word* ra = &ctx->h_sz[(size_t)(t0 + tr) * words];
      const sz_word* rb = &ctx->h_sz[(size_t)(t0 + tc) * words];
      bool on = false;
      for (long k = kt0; k < kt1 && !on; ++k) on = ((ra[k >> 6] & rb[k >> 6]) >> (k & 63)) & 1;
      all += 1;
      live += on ? 1 : 0;
    }
  return all > 0 ? live / all : 1.0;
}

}  // namespace sgp
