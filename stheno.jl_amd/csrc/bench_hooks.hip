// libsthenomi_bench.so -- the micro-benchmark, diagnosis and test hooks of include/sthenomi_bench.h (round 6: they used to be
// linked into the product library; the round-5 verdict: "product .so exports exactly include/sthenomi.h").  Links against
// libsthenomi.so and drives its internal launchers (common.h) on contexts created by the product library; nothing here is on
// a product path.  bench.py, tools/ and the GPU tests load it explicitly (stheno.jl_amd/lib.py: bench_lib()).
#include "ctx.h"
#include "driver.h"
#include "tilemap.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace sgp {
int run_mfma_bench(hipStream_t s, int iters, double* tflops_out, double* layout_maxerr_out);
int run_hbm_bench(hipStream_t s, long bytes, int iters, double* write_gbs, double* copy_gbs);
constexpr long NOMASK = -(1L << 40);
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

static double update_flops(long m, long nc, long k) {
  // algorithmic flops of C[lower, m x nc] -= P P': triangle of the square part + rows below
  return (double)k * (double)nc * (double)(nc + 1) + 2.0 * (double)k * (double)(m - nc) * (double)nc;
}

// operators of a context that were rerun on the launch-based schedule because a dataflow launch ran into its wait bound
extern "C" int sgp_bench_df_fallbacks(sgp_ctx* ctx, int64_t* out) {
  CHECK_ARG(ctx && out, "sgp_bench_df_fallbacks: NULL argument");
  *out = ctx->df_fallbacks;
  return 0;
}

// ---- multi-GPU context: failure-path hooks and per-piece profile (csrc/multi.hip keeps the state) --------------------------
extern "C" int sgp_bench_multi_fault(sgp_ctx* ctx, int rank, int64_t step) {
  CHECK_ARG(ctx && ctx->multi, "sgp_bench_multi_fault: not a multi-GPU context");
  return sgp_multi_set_fault(ctx->multi, rank, rank >= 0 ? (long)std::max<int64_t>(0, step) : -1, 0.0);
}
extern "C" int sgp_bench_multi_stall(sgp_ctx* ctx, int rank, int64_t step, double seconds) {
  CHECK_ARG(ctx && ctx->multi && seconds > 0.0, "sgp_bench_multi_stall: bad argument");
  return sgp_multi_set_fault(ctx->multi, rank, (long)std::max<int64_t>(0, step), seconds);
}
extern "C" int sgp_bench_multi_broken(sgp_ctx* ctx, int* out) {
  CHECK_ARG(ctx && ctx->multi && out, "sgp_bench_multi_broken: not a multi-GPU context");
  *out = sgp_multi_is_broken(ctx->multi);
  return 0;
}
extern "C" int sgp_bench_multi_profile_pieces(sgp_ctx* ctx, double* out, int64_t cap, int64_t* n_out) {
  CHECK_ARG(ctx && ctx->multi && n_out, "sgp_bench_multi_profile_pieces: bad argument");
  const std::vector<double>& v = sgp_multi_profile_pieces(ctx->multi);
  *n_out = (int64_t)v.size();
  if (out) {
    CHECK_ARG(cap >= (int64_t)v.size(), "sgp_bench_multi_profile_pieces: buffer too small");
    std::copy(v.begin(), v.end(), out);
  }
  return 0;
}

// ---------------------------------------------------------------------------------------
// micro-benchmarks
// ---------------------------------------------------------------------------------------
extern "C" int sgp_bench_mfma_f64(sgp_ctx* ctx, int iters, double* tflops_out, double* layout_maxerr_out) {
  CHECK_ARG(ctx && tflops_out && layout_maxerr_out, "sgp_bench_mfma_f64: NULL argument");
  CtxScope scope(ctx);
  return run_mfma_bench(ctx->stream, iters, tflops_out, layout_maxerr_out);
}
extern "C" int sgp_bench_hbm(sgp_ctx* ctx, int64_t bytes, int iters, double* write_gbs_out, double* copy_gbs_out) {
  CHECK_ARG(ctx && write_gbs_out && copy_gbs_out, "sgp_bench_hbm: NULL argument");
  CtxScope scope(ctx);
  return run_hbm_bench(ctx->stream, bytes, iters, write_gbs_out, copy_gbs_out);
}

// CU census under a CU mask: which (XCD, SE, SH, CU) the workgroups of a stream created with `mask` land on
// (cnt[xcc << 8 | HW_ID[15:8]]); one workgroup per CU at a time (80 KB of LDS), each staying ~10 us.
__global__ __launch_bounds__(256) void cu_census_kernel(unsigned* cnt, long long spin) {
  extern __shared__ double census_lds[];
  if (threadIdx.x == 0) {
    const unsigned hw = __builtin_amdgcn_s_getreg(63492);   // HW_ID
    const unsigned xcc = __builtin_amdgcn_s_getreg(6164);   // XCC_ID 3:0
    atomicAdd(&cnt[((xcc & 15) << 8) | ((hw >> 8) & 255)], 1u);
    census_lds[0] = 0.0;
  }
  const long long t0 = (long long)__builtin_amdgcn_s_memtime();
  while ((long long)__builtin_amdgcn_s_memtime() - t0 < spin) __builtin_amdgcn_s_sleep(8);
}
extern "C" int sgp_bench_cumask(sgp_ctx* ctx, const uint32_t* mask, int words, int nwg, unsigned* out /* [4096] */) {
  CHECK_ARG(ctx && out && nwg > 0, "sgp_bench_cumask: NULL argument");
  CtxScope scope(ctx);
  hipStream_t st = nullptr;
  if (mask && words > 0) SGP_HIP(hipExtStreamCreateWithCUMask(&st, (uint32_t)words, mask));
  else SGP_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  unsigned* d = nullptr;
  if (hipMalloc(&d, 4096 * sizeof(unsigned)) != hipSuccess) { hipStreamDestroy(st); set_error("hipMalloc failed"); return -2; }
  hipMemsetAsync(d, 0, 4096 * sizeof(unsigned), st);
  SGP_LDS_ATTR_ONCE(cu_census_kernel, 81920);
  hipLaunchKernelGGL(cu_census_kernel, dim3((unsigned)nwg), dim3(256), 81920, st, d, 20000LL);
  hipError_t e = hipStreamSynchronize(st);
  if (e == hipSuccess) e = hipMemcpy(out, d, 4096 * sizeof(unsigned), hipMemcpyDeviceToHost);
  hipFree(d);
  hipStreamDestroy(st);
  SGP_HIP(e);
  return 0;
}

__global__ void fill_rand_kernel(double* p, long n, unsigned long long seed) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned long long x = seed + (unsigned long long)i * 0x9E3779B97F4A7C15ULL;
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ULL; x ^= x >> 27; x *= 0x94D049BB133111EBULL; x ^= x >> 31;
  p[i] = (double)(x >> 11) * (1.0 / 9007199254740992.0) * 2.0 - 1.0;
}

// one potrf_diag launch on a well-conditioned 128 x 128 tile with s_memtime stamps of wave 0 at its phase
// boundaries (stamps_out[64], ticks; 0-terminated) and the launch time by HIP events (us_out)
extern "C" int sgp_bench_potrf(sgp_ctx* ctx, int iters, double* us_out, long long* stamps_out) {
  CHECK_ARG(ctx && us_out && stamps_out, "sgp_bench_potrf: NULL argument");
  CtxScope scope(ctx);
  hipStream_t s = ctx->stream;
  DevBuf A, A0;
  CHECK_RC(A.alloc((size_t)TILE * TILE));
  CHECK_RC(A0.alloc((size_t)TILE * TILE));
  std::vector<double> h((size_t)TILE * TILE);
  for (int c = 0; c < TILE; ++c)
    for (int r = 0; r < TILE; ++r) h[r + (size_t)c * TILE] = (r == c ? 2.0 : 0.0) + std::exp(-0.02 * (r - c) * (r - c));
  SGP_HIP(hipMemcpy(A0.p, h.data(), sizeof(double) * TILE * TILE, hipMemcpyHostToDevice));
  long long* d_dbg = nullptr;
  SGP_HIP(hipMalloc(&d_dbg, sizeof(long long) * 64));
  SGP_HIP(hipMemset(d_dbg, 0, sizeof(long long) * 64));
  hipEvent_t e0, e1;
  SGP_HIP(hipEventCreate(&e0));
  SGP_HIP(hipEventCreate(&e1));
  double tot = 0;
  for (int it = 0; it < iters + 1; ++it) {
    SGP_HIP(hipMemcpyAsync(A.p, A0.p, sizeof(double) * TILE * TILE, hipMemcpyDeviceToDevice, s));
    SGP_HIP(hipEventRecord(e0, s));
    if (it == iters)
      CHECK_RC(launch_potrf_diag_dbg(A.p, TILE, ctx->d_invd, ctx->d_slots, ctx->d_info, d_dbg, s));
    else
      CHECK_RC(launch_potrf_diag(A.p, TILE, ctx->d_invd, ctx->d_slots, ctx->d_info, 0, s));
    SGP_HIP(hipEventRecord(e1, s));
    SGP_HIP(hipEventSynchronize(e1));
    float ms = 0;
    SGP_HIP(hipEventElapsedTime(&ms, e0, e1));
    if (it > 0 && it < iters) tot += ms;
  }
  *us_out = tot / std::max(1, iters - 1) * 1e3;
  SGP_HIP(hipMemcpy(stamps_out, d_dbg, sizeof(long long) * 64, hipMemcpyDeviceToHost));
  hipFree(d_dbg);
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  return 0;
}

// potrf_diag under the look-ahead's contention: `gemm_launches` trailing updates C(m^2 lower) -= P P' (depth k) are
// queued on the update stream, then n potrf_diag launches go one by one down the panel stream: us_out[i] is launch
// i's time by HIP events (it includes the wait for a free workgroup slot), ticks_out[i] its own s_memtime span of
// wave 0 (the time it runs once resident).  busy_out[i] = 1 while the updates had not finished.
extern "C" int sgp_bench_potrf_contended(sgp_ctx* ctx, int64_t m, int64_t k, int gemm_launches, int n,
                                         double* us_out, long long* ticks_out, int* busy_out) {
  CHECK_ARG(ctx && us_out && ticks_out && busy_out && n > 0, "sgp_bench_potrf_contended: NULL argument");
  CHECK_ARG(m % TILE == 0 && k % 16 == 0 && m > 0, "sgp_bench_potrf_contended: bad sizes");
  CtxScope scope(ctx);
  hipStream_t s = ctx->stream, s2 = ctx->stream2;
  DevBuf P, Cm, A, A0;
  CHECK_RC(P.alloc((size_t)m * k));
  CHECK_RC(Cm.alloc((size_t)m * m));
  CHECK_RC(A.alloc((size_t)TILE * TILE));
  CHECK_RC(A0.alloc((size_t)TILE * TILE));
  std::vector<double> h((size_t)TILE * TILE);
  for (int c = 0; c < TILE; ++c)
    for (int r = 0; r < TILE; ++r) h[r + (size_t)c * TILE] = (r == c ? 2.0 : 0.0) + std::exp(-0.02 * (r - c) * (r - c));
  SGP_HIP(hipMemcpy(A0.p, h.data(), sizeof(double) * TILE * TILE, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(fill_rand_kernel, dim3((unsigned)((m * k + 255) / 256)), dim3(256), 0, s2, P.p, m * k, 99ULL);
  SGP_HIP(hipMemsetAsync(Cm.p, 0, sizeof(double) * m * m, s2));
  long long* d_dbg = nullptr;
  SGP_HIP(hipMalloc(&d_dbg, sizeof(long long) * 64));
  SGP_HIP(hipMemset(d_dbg, 0, sizeof(long long) * 64));
  hipEvent_t e0, e1, eg;
  SGP_HIP(hipEventCreate(&e0));
  SGP_HIP(hipEventCreate(&e1));
  SGP_HIP(hipEventCreate(&eg));
  SGP_HIP(hipStreamSynchronize(s2));
  int rc = 0;
  for (int i = 0; i < gemm_launches && rc == 0; ++i) rc = launch_gemm_nt_update(P.p, m, Cm.p, m, m, m, k, s2);
  if (rc == 0 && hipEventRecord(eg, s2) != hipSuccess) rc = -2;
  long long st[64];
  for (int i = 0; i < n && rc == 0; ++i) {
    hipMemcpyAsync(A.p, A0.p, sizeof(double) * TILE * TILE, hipMemcpyDeviceToDevice, s);
    hipStreamSynchronize(s);
    hipEventRecord(e0, s);
    rc = launch_potrf_diag_dbg(A.p, TILE, ctx->d_invd, ctx->d_slots, ctx->d_info, d_dbg, s);
    hipEventRecord(e1, s);
    hipEventSynchronize(e1);
    busy_out[i] = hipEventQuery(eg) == hipErrorNotReady ? 1 : 0;
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    us_out[i] = ms * 1e3;
    hipMemcpyAsync(st, d_dbg, sizeof(st), hipMemcpyDeviceToHost, s);
    hipStreamSynchronize(s);
    long long last = 0;
    for (int q = 0; q < 64; ++q)
      if (st[q]) last = st[q];
    ticks_out[i] = last - st[0];
  }
  hipStreamSynchronize(s2);
  hipFree(d_dbg);
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  hipEventDestroy(eg);
  return rc;
}

// one lower update C(m x m) -= P P' (depth k) with per-workgroup phase stamps: out[8 * id + {0: entry, 1: first operand
// chunk + old C tile landed, 2: contraction done, 3: stores drained, 4: XCC_ID << 16 | HW_ID, 5: tile row, 6: tile col}]
// (s_memtime ticks; 0 rows = ids without a tile).  *n_ids = workgroups of the launch (call with out == NULL to size).
extern "C" int sgp_bench_gemm_stamps(sgp_ctx* ctx, int64_t m, int64_t k, long long* out, int64_t cap, int64_t* n_ids) {
  CHECK_ARG(ctx && n_ids && m % TILE == 0 && k % 16 == 0, "sgp_bench_gemm_stamps: bad argument");
  CtxScope scope(ctx);
  long ids = 0;
  CHECK_RC(launch_gemm_nt_stamps(nullptr, m, nullptr, m, m, m, k, nullptr, &ids, ctx->stream));
  *n_ids = ids;
  if (!out) return 0;
  CHECK_ARG(cap >= 8 * ids, "sgp_bench_gemm_stamps: buffer too small");
  hipStream_t s = ctx->stream;
  DevBuf A, C, D;
  CHECK_RC(A.alloc((size_t)m * k));
  CHECK_RC(C.alloc((size_t)m * m));
  CHECK_RC(D.alloc((size_t)8 * ids));
  hipLaunchKernelGGL(fill_rand_kernel, dim3((unsigned)((m * k + 255) / 256)), dim3(256), 0, s, A.p, m * k, 1234ULL);
  SGP_HIP(hipMemsetAsync(C.p, 0, sizeof(double) * m * m, s));
  CHECK_RC(launch_gemm_nt_update(A.p, m, C.p, m, m, m, k, s));   // warm-up (clocks, caches)
  SGP_HIP(hipMemsetAsync(D.p, 0, sizeof(double) * 8 * ids, s));
  // (rounds 3 / 4 ran this launch under experiment switches -- beta = 0, scrambled tile ids, rotated k offsets, padded LDS:
  // profiles/archive/r03_experiments, r04_experiments; removed in round 6)
  CHECK_RC(launch_gemm_nt_stamps(A.p, m, C.p, m, m, m, k, (long long*)D.p, &ids, s));
  SGP_HIP(hipStreamSynchronize(s));
  SGP_HIP(hipMemcpy(out, D.p, sizeof(long long) * 8 * ids, hipMemcpyDeviceToHost));
  return 0;
}

extern "C" int sgp_bench_gemm(sgp_ctx* ctx, int64_t m, int64_t n, int64_t k, int lower_only,
                              int iters, double* tflops_out, double* maxerr_out) {
  CHECK_ARG(ctx && tflops_out && maxerr_out, "sgp_bench_gemm: NULL argument");
  CHECK_ARG(m % TILE == 0 && n % TILE == 0 && k % 16 == 0 && m >= n, "sgp_bench_gemm: bad sizes");
  CtxScope scope(ctx);
  hipStream_t s = ctx->stream;
  DevBuf A, C;
  CHECK_RC(A.alloc((size_t)m * k));
  CHECK_RC(C.alloc((size_t)m * n));
  hipLaunchKernelGGL(fill_rand_kernel, dim3((unsigned)((m * k + 255) / 256)), dim3(256), 0, s, A.p, m * k, 1234ULL);
  SGP_HIP(hipMemsetAsync(C.p, 0, sizeof(double) * m * n, s));
  // C = -A[0:m] A[0:n]'
  CHECK_RC(launch_gemm_nt(A.p, m, A.p, m, C.p, m, m, n, k, -1.0, 1.0, (lower_only & 1) ? 0 : NOMASK, 0, 0, s));
  SGP_HIP(hipStreamSynchronize(s));
  // spot check 64 entries in the lower part against a host dot product
  std::vector<double> hA((size_t)m * k);
  SGP_HIP(hipMemcpy(hA.data(), A.p, sizeof(double) * m * k, hipMemcpyDeviceToHost));
  double me = 0;
  for (int q = 0; q < 64; ++q) {
    long c = (long)((q * 7919L) % n), r = c + (long)((q * 104729L) % (m - c));
    double ref = 0;
    for (long kk = 0; kk < k; ++kk) ref -= hA[r + kk * m] * hA[c + kk * m];
    double got = 0;
    SGP_HIP(hipMemcpy(&got, C.p + r + c * m, sizeof(double), hipMemcpyDeviceToHost));
    me = std::max(me, std::fabs(got - ref));
  }
  *maxerr_out = me;
  hipEvent_t e0, e1;
  SGP_HIP(hipEventCreate(&e0));
  SGP_HIP(hipEventCreate(&e1));
  SGP_HIP(hipEventRecord(e0, s));
  const bool reg_baseline = (lower_only & 2) != 0;  // bench-only: bit 1 = register-staged baseline kernel
  const long REG_BASELINE = -(1L << 50);            // (an argument of this one call: no process-global switch)
  lower_only &= 1;
  for (int i = 0; i < iters; ++i)
    if (reg_baseline)
      CHECK_RC(launch_gemm_nt(A.p, m, A.p, m, C.p, m, m, n, k, -1.0, 1.0, lower_only ? 0 : NOMASK, 0, REG_BASELINE, s));
    else if (lower_only)
      CHECK_RC(launch_gemm_nt_update(A.p, m, C.p, m, m, n, k, s));  // the production trailing-update symbol
    else
      CHECK_RC(launch_gemm_nt(A.p, m, A.p, m, C.p, m, m, n, k, -1.0, 1.0, NOMASK, 0, 0, s));
  SGP_HIP(hipEventRecord(e1, s));
  SGP_HIP(hipEventSynchronize(e1));
  float ms = 0;
  SGP_HIP(hipEventElapsedTime(&ms, e0, e1));
  double fl = lower_only ? update_flops(m, n, k) : 2.0 * (double)m * (double)n * (double)k;
  *tflops_out = fl * iters / (ms * 1e-3) / 1e12;
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  return 0;
}

