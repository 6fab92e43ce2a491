/* sthenomi_bench.h -- micro-benchmark and diagnosis hooks of libsthenomi.so.
 *
 * NOT part of the drop-in boundary (include/sthenomi.h): nothing a Julia / C host needs to run the GP operators is
 * declared here, and an installation ships sthenomi.h alone.  bench.py, tools/ and two GPU tests use these entry points to
 * pin the roofline peaks, the MFMA lane maps and the panel kernels' phase timings on the box.
 * Plain C like the product header. */
#ifndef STHENOMI_BENCH_H
#define STHENOMI_BENCH_H

#include "sthenomi.h"

#ifdef __cplusplus
extern "C" {
#endif

/* micro-benchmarks used to pin the roofline peaks on the box */
int sgp_bench_mfma_f64(sgp_ctx* ctx, int iters, double* tflops_out, double* layout_maxerr_out);
int sgp_bench_hbm(sgp_ctx* ctx, int64_t bytes, int iters, double* write_gbs_out, double* copy_gbs_out);
/* one potrf_diag launch (128 x 128 diagonal block) timed by HIP events, plus s_memtime stamps of its phases */
int sgp_bench_potrf(sgp_ctx* ctx, int iters, double* us_out, long long* stamps_out /* [64] */);
/* potrf_diag while `gemm_launches` trailing updates (m^2 lower, depth k) run on the update stream: per launch the
 * HIP-event time (incl. the wait for a workgroup slot), the kernel's own s_memtime span, and whether the updates
 * were still running. */
/* which CUs a stream created with the CU mask `mask` (`words` 32-bit words; NULL = no mask) runs on:
 * out[xcc << 8 | HW_ID[15:8]] = workgroups seen there (tools/gpu_cumask.py decodes it) */
int sgp_bench_cumask(sgp_ctx* ctx, const uint32_t* mask, int words, int nwg, unsigned* out /* [4096] */);
int sgp_bench_potrf_contended(sgp_ctx* ctx, int64_t m, int64_t k, int gemm_launches, int n, double* us_out /* [n] */,
                              long long* ticks_out /* [n] */, int* busy_out /* [n] */);
/* one lower trailing update C(m x m) -= P P' (depth k) with per-workgroup phase stamps of the tile program (s_memtime
 * ticks): out[8 id + {0 entry, 1 first operand chunk + old C tile landed, 2 contraction done, 3 stores drained,
 * 4 XCC_ID << 16 | HW_ID, 5 tile row, 6 tile column}]; out == NULL: only *n_ids (workgroups of the launch). */
int sgp_bench_gemm_stamps(sgp_ctx* ctx, int64_t m, int64_t k, long long* out, int64_t cap, int64_t* n_ids);
/* raw GEMM-NT kernel timing: C(m x n) -= A(m x k) B(n x k)' on random data */
int sgp_bench_gemm(sgp_ctx* ctx, int64_t m, int64_t n, int64_t k, int lower_only, int iters,
                   double* tflops_out, double* maxerr_out);

/* how many operators of this context were rerun on the launch-based schedule because the dataflow factorisation ran into
 * its wait bound (SGP_DF_TIMEOUT_S; capi.hip: with_df_fallback) */
int sgp_bench_df_fallbacks(sgp_ctx* ctx, int64_t* out);

/* Multi-GPU context, failure path (tests/test_gpu_multi_faults.py; csrc/multi.hip: sgp_multi::fault_rank).  TEST-ONLY fault
 * hook: the next sharded factorisation fails mid-schedule -- at panel `step`, on the enqueue thread of `rank` (rank < 0
 * disarms) -- exactly as a failing HIP / RCCL call on that thread would; the hook disarms itself when it fires.  Also
 * SGP_MULTI_FAULT=rank:step at context creation. */
int sgp_bench_multi_fault(sgp_ctx* ctx, int rank, int64_t step);
/* ... or makes that thread SLEEP `seconds` there without failing: the other ranks' threads then wait for its event records and
 * must give up at their wall-clock bound (SGP_MULTI_SPIN_TIMEOUT_S) instead of spinning for ever */
int sgp_bench_multi_stall(sgp_ctx* ctx, int rank, int64_t step, double seconds);
/* profile mode (sgp_ctx_multi_profile): per panel 8 doubles, the ms of each sub-panel's launch group alone on the hardware */
int sgp_bench_multi_profile_pieces(sgp_ctx* ctx, double* out, int64_t cap, int64_t* n_out);
/* *out = 1 once a failed call has left the context's RCCL communicators aborted (it then refuses sharded calls) */
int sgp_bench_multi_broken(sgp_ctx* ctx, int* out);

#ifdef __cplusplus
}
#endif
#endif /* STHENOMI_BENCH_H */
