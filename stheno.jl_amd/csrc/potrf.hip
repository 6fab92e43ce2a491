// Diagonal-block kernels of the blocked Cholesky (the latency-bound critical path):
//   potrf_diag_kernel : Cholesky of one 128x128 diagonal block, entirely in LDS, one workgroup.
//   (round 6: the explicit-inverse trtri_kernel of the SGP_REFINE = 0 / 2 A/B modes is gone: the refined 16 x 16 substitution,
//   panel_solve_kernel, has been the only panel solve since round 2)
// Together with the panel solve (panel_solve.h) it replaces LAPACK dpotf2/dtrsm inside dpotrf on the reference path
// (LinearAlgebra.cholesky under AbstractGPs.logpdf/posterior/rand/elbo [EXT]).
//
// potrf_diag (potrf_diag.h: potrf_diag_body, shared with the fused update + diagonal block launches of
// gemm_nt.hip): right-looking over 16-column sub-panels.  The rank-16 updates and the sub-panel
// solves run on v_mfma_f64_16x16x4_f64 straight out of LDS (packed 16x16 blocks, k-major:
// conflict-free operand reads); the 16x16 micro-Cholesky and its inverse run in the registers of one wave with
// DPP row broadcasts inside the FMAs (no LDS round trips, no barriers) -- "wavefront shuffles for the
// diagonal panel" in north-star terms.
#include "common.h"
#include "potrf_diag.h"   // potrf_diag_body: the diagonal-block factorisation itself
#include "panel_solve.h"  // panel_solve_fill / panel_solve_strip: the 16-row blocked substitution (shared with chol_df.hip)
#include <algorithm>
#include <cstdlib>

namespace sgp {

__global__ __launch_bounds__(PD_THREADS, 4) void potrf_diag_kernel(double* A, long ld, double* invd,
                                                                   double* logdet_slot, int* info,
                                                                   long gcol0, int prio) {
  potrf_diag_body<false, double>(A, ld, invd, logdet_slot, info, gcol0, prio, nullptr);
}
__global__ __launch_bounds__(PD_THREADS, 4) void potrf_diag_f32_kernel(float* A, long ld, double* invd,
                                                                       double* logdet_slot, int* info,
                                                                       long gcol0, int prio) {
  potrf_diag_body<false, float>(A, ld, invd, logdet_slot, info, gcol0, prio, nullptr);
}
// same code with s_memtime stamps of wave 0 at the phase boundaries (sgp_bench_potrf)
__global__ __launch_bounds__(PD_THREADS, 4) void potrf_diag_dbg_kernel(double* A, long ld, double* invd,
                                                                       double* logdet_slot, int* info,
                                                                       long gcol0, int prio, long long* dbg) {
  potrf_diag_body<true, double>(A, ld, invd, logdet_slot, info, gcol0, prio, dbg);
}


// bench / diagnosis: one launch with phase stamps (s_memtime ticks of wave 0) into dbg[64]
int launch_potrf_diag_dbg(double* A, long ld, double* d_invd, double* d_logdet_slot, int* d_info, long long* dbg,
                          hipStream_t s) {
  SGP_LDS_ATTR_ONCE(potrf_diag_dbg_kernel, PD_LDS);
  hipLaunchKernelGGL(potrf_diag_dbg_kernel, dim3(1), dim3(PD_THREADS), PD_LDS, s, A, ld, d_invd,
                     d_logdet_slot, d_info, 0L, panel_prio(), dbg);
  SGP_HIP(hipGetLastError());
  return 0;
}

int launch_potrf_diag_f32(float* A, long ld, double* d_invd, double* d_logdet_slot, int* d_info, long gcol0,
                          hipStream_t s) {
  SGP_LDS_ATTR_ONCE(potrf_diag_f32_kernel, PD_LDS);
  hipLaunchKernelGGL(potrf_diag_f32_kernel, dim3(1), dim3(PD_THREADS), PD_LDS, s, A, ld, d_invd, d_logdet_slot, d_info,
                     gcol0, panel_prio());
  SGP_HIP(hipGetLastError());
  return 0;
}

int launch_potrf_diag(double* A, long ld, double* d_invd, double* d_logdet_slot, int* d_info,
                      long gcol0, hipStream_t s) {
  SGP_LDS_ATTR_ONCE(potrf_diag_kernel, PD_LDS);
  hipLaunchKernelGGL(potrf_diag_kernel, dim3(1), dim3(PD_THREADS), PD_LDS, s, A, ld, d_invd,
                     d_logdet_slot, d_info, gcol0, panel_prio());
  SGP_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------------------
// panel_solve_kernel: X <- X inv(L11)' for a strip of rows (the panel TRSM; dtrsm on the reference
// path), blocked substitution over the eight 16-column blocks of L11:
//   T_c = B_c - sum_{p<c} X_p L_cp',   X_c = T_c inv(L_cc)' + one refinement step against L_cc
// (the refinement makes the inverse products backward stable, see potrf_diag step 3 and
// tools/emul_chol.py).  One wave owns 16 rows and all 128 columns, transposed into the f64 MFMA
// accumulator layout (m = column, n = row): the accumulator lane map equals the B-operand lane map,
// so every product consumes the previous one's registers -- no LDS round trips, no barriers after
// the operand fill.  The 36 lower 16x16 blocks of L11 sit in LDS k-major (each k-step one contiguous
// 512-byte read), the 8 inverse diagonal blocks in registers.  73.7 KB of LDS and 4 waves: the
// workgroup fits into the slot one trailing-update GEMM workgroup leaves on a CU (see potrf_diag).
// ---------------------------------------------------------------------------------------
constexpr int PS_ROWS = 64;  // rows per workgroup (4 waves)
constexpr size_t PS_LDS = (size_t)36 * 256 * sizeof(double);  // == one GEMM workgroup's LDS

template <typename TS>
__device__ __forceinline__ void panel_solve_body(TS* X, long ldx, const TS* L, long ldl, const double* inv,
                                                 long inv_cstride, long inv_kstride, int strips, long rows, int prio,
                                                 StripSkip sk = StripSkip());

__global__ __launch_bounds__(256, 2) void panel_solve_kernel(double* X, long ldx, const double* L, long ldl,
                                                          const double* inv, long inv_cstride,
                                                          long inv_kstride, int strips, long rows, int prio, StripSkip sk) {
  panel_solve_body<double>(X, ldx, L, ldl, inv, inv_cstride, inv_kstride, strips, rows, prio, sk);
}
// fp32 storage, fp64 arithmetic (f32.hip): the rows and L11 are converted as they are read, the solved rows as
// they are written
__global__ __launch_bounds__(256, 2) void panel_solve_f32_kernel(float* X, long ldx, const float* L, long ldl,
                                                              const double* inv, long inv_cstride,
                                                              long inv_kstride, int strips, long rows, int prio, StripSkip sk) {
  panel_solve_body<float>(X, ldx, L, ldl, inv, inv_cstride, inv_kstride, strips, rows, prio, sk);
}

template <typename TS>
__device__ __forceinline__ void panel_solve_body(TS* X, long ldx, const TS* L, long ldl, const double* inv,
                                                 long inv_cstride, long inv_kstride, int strips, long rows, int prio,
                                                 StripSkip sk) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  set_wave_prio(prio);
  double* sL = smem;  // block (c, p), c >= p at (c (c + 1) / 2 + p) * 256, [k][m]
  const int t = threadIdx.x;
  const int lane = t & 63, w = t >> 6;
  const int l15 = lane & 15, lq = lane >> 4;
  panel_solve_fill<TS, 4>(sL, L, ldl, __builtin_amdgcn_readfirstlane(w), lane);
  __syncthreads();
  // `strips` 64-row strips per workgroup: under the look-ahead overlap CU slots are the scarce
  // resource (they free up at the rate trailing-update workgroups retire), so a slot once taken
  // amortises its operand fill over several strips
#pragma unroll 1
  for (int st = 0; st < strips; ++st) {
    __builtin_amdgcn_sched_barrier(0);
    const long row0 = ((long)blockIdx.x * strips + st) * PS_ROWS;
    if (row0 >= rows) break;
    // structural zeros: the 128-row tile these 64 rows belong to holds exact zeros in this block column -- nothing to solve
    if (sk.nz) {
      const long tr = sk.tr0 + (row0 >> 7);
      if (!((sk.nz[tr * sk.words + (sk.kt >> 6)] >> (sk.kt & 63)) & 1)) continue;
    }
    // uniform column base (scalar registers) + one 32-bit lane offset: keeps the 32 column
    // addresses out of the vector registers
    const int loff = (int)(row0 + w * 16 + l15 + lq * ldx);
    panel_solve_strip<TS>(X, ldx, loff, sL, inv, inv_cstride, inv_kstride, lane);
  }
}

int launch_panel_solve_f32(float* X, long ldx, long rows, const float* L, long ldl, const double* inv,
                           long inv_cstride, long inv_kstride, hipStream_t s, const StripSkip* sk) {
  if (rows <= 0) return 0;
  if (rows % PS_ROWS) {
    set_error("panel_solve_f32: rows must be a multiple of 64");
    return -1;
  }
  SGP_LDS_ATTR_ONCE(panel_solve_f32_kernel, PS_LDS);
  long nstrips = rows / PS_ROWS;
  int strips = (int)std::min<long>(8, std::max<long>(1, (nstrips + 255) / 256));
  long nwg = (nstrips + strips - 1) / strips;
  hipLaunchKernelGGL(panel_solve_f32_kernel, dim3((unsigned)nwg), dim3(256), PS_LDS, s, X, ldx, L, ldl, inv,
                     inv_cstride, inv_kstride, strips, rows, panel_prio(), sk ? *sk : StripSkip());
  SGP_HIP(hipGetLastError());
  return 0;
}

int launch_panel_solve(double* X, long ldx, long rows, const double* L, long ldl, const double* inv,
                       long inv_cstride, long inv_kstride, hipStream_t s, const StripSkip* sk, long div) {
  if (rows <= 0) return 0;
  if (rows % PS_ROWS) {
    set_error("panel_solve: rows must be a multiple of 64");
    return -1;
  }
  SGP_LDS_ATTR_ONCE(panel_solve_kernel, PS_LDS);
  // one strip per workgroup until the chip is full (`div` workgroups: 256 = one per CU, what the factorisation's panel solve
  // wants under the look-ahead overlap; a stand-alone row solve takes two per CU), then fatter workgroups
  if (div <= 0) div = 256;
  long nstrips = rows / PS_ROWS;
  int strips = (int)std::min<long>(8, std::max<long>(1, (nstrips + div - 1) / div));
  long nwg = (nstrips + strips - 1) / strips;
  hipLaunchKernelGGL(panel_solve_kernel, dim3((unsigned)nwg), dim3(256), PS_LDS, s, X, ldx, L, ldl, inv,
                     inv_cstride, inv_kstride, strips, rows, panel_prio(), sk ? *sk : StripSkip());
  SGP_HIP(hipGetLastError());
  return 0;
}

}  // namespace sgp
