// Internal definitions shared by the HIP translation units of libsthenomi.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include <string>
#include <vector>

namespace sgp {

constexpr int TILE = 128;      // tile edge of every blocked kernel (rows and cols)
constexpr int MICRO = 16;      // MFMA f64 16x16x4 micro-tile edge
constexpr int LDS_LD = 144;    // LDS leading dimension for 128-row operand panels:
                               // 144 % 32 == 16 makes the (16 rows x 2 k) ds_read_b64 pattern
                               // of a 32-lane group hit 32 distinct 8-byte slots.

typedef double d4 __attribute__((ext_vector_type(4)));

// D = A(16x4) * B(4x16) + C on one wave.  Operand/result lane maps (guide section 3):
//   A-operand lane l holds A[m = l & 15][k = l >> 4]
//   B-operand lane l holds B[k = l >> 4][n = l & 15]
//   result    lane l, reg r holds D[m = (l >> 4) + 4 r][n = l & 15]
__device__ __forceinline__ d4 mfma_f64(double a, double b, d4 c) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// 4-block form: D_b(4x4) = A_b(4x4) B_b(4x4) + C_b for b = 0..3, one f64 per lane
// (lane maps in gemm_nt.hip).  Issues every 16 cycles on gfx950 -- the full fp64 matrix rate.
__device__ __forceinline__ double mfma44_f64(double a, double b, double c) {
  return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
}

// ---- fp64 elementary functions of the covariance kernels -----------------------------------------
// The assembly kernel is fp64-VALU bound: libm's exp / sqrt / a true division
// cost it ~100 issue slots per Matern-5/2 entry.  These straight-line versions (no special-case
// branches; arguments are known to be <= 0 resp. >= 0) are accurate to ~1 ulp, which is far inside
// the 1e-12 entry-wise parity the tests hold against the CPU oracle.
// exp(x) for x <= 0: n = rint(x log2 e), r = x - n ln2 (two-part), degree-13 Taylor on |r| <= 0.347
// (remainder 4e-18), scaled by 2^n with v_ldexp_f64 (correct down to denormals / 0).
__device__ __forceinline__ double exp_nonpos(double x) {
  x = fmax(x, -800.0);
  const double n = __builtin_rint(x * 1.4426950408889634);
  double r = fma(n, -6.93147180369123816490e-01, x);
  r = fma(n, -1.90821492927058770002e-10, r);
  double p = 1.6059043836821613e-10;          // 1/13!
  p = fma(p, r, 2.08767569878681e-09);        // 1/12!
  p = fma(p, r, 2.505210838544172e-08);       // 1/11!
  p = fma(p, r, 2.755731922398589e-07);       // 1/10!
  p = fma(p, r, 2.7557319223985893e-06);      // 1/9!
  p = fma(p, r, 2.48015873015873e-05);        // 1/8!
  p = fma(p, r, 0.0001984126984126984);       // 1/7!
  p = fma(p, r, 0.001388888888888889);        // 1/6!
  p = fma(p, r, 0.008333333333333333);        // 1/5!
  p = fma(p, r, 0.041666666666666664);        // 1/4!
  p = fma(p, r, 0.16666666666666666);         // 1/3!
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  return __builtin_ldexp(p, (int)n);
}
// sqrt(a) for a >= 0: v_rsq_f64 seed (~2^-26), one Newton step on 1/sqrt, then one Newton step on the
// root itself (quadratic: rounding-level result).  a is clamped to 1e-300 so that a == 0 needs no
// branch: the result is then 1e-150, which every kernel maps to exactly kappa(0) (1 + 1e-150 == 1).
__device__ __forceinline__ double sqrt_nonneg(double a) {
  a = fmax(a, 1e-300);
  double r = __builtin_amdgcn_rsq(a);
  r = r * fma(-0.5 * a, r * r, 1.5);
  double d = a * r;
  return fma(0.5 * r, fma(-d, d, a), d);
}

// device-side description of one covariance term of one block pair
struct DevTerm {
  int kind;
  int dim;
  double coef;
  double param;
  const double* xr;  // D x nr, first point of the row block
  long ldr;
  const double* xc;  // D x nc
  long ldc;
  const double* rs;  // row scale or nullptr
  const double* cs;  // col scale or nullptr
};

void set_error(const std::string& s);

#define SGP_HIP(expr)                                                                     \
  do {                                                                                    \
    hipError_t _e = (expr);                                                               \
    if (_e != hipSuccess) {                                                               \
      sgp::set_error(std::string(#expr) + ": " + hipGetErrorString(_e) + " at " + __FILE__ + \
                     ":" + std::to_string(__LINE__));                                     \
      return -2;                                                                          \
    }                                                                                     \
  } while (0)

// Raise a kernel's dynamic-LDS limit once per device (function attributes are per device; a process
// may hold contexts on several GPUs).
#define SGP_LDS_ATTR_ONCE(func, bytes)                                                              \
  do {                                                                                             \
    static std::atomic<bool> done_[64];                                                                  \
    int dev_ = 0;                                                                                  \
    SGP_HIP(hipGetDevice(&dev_));                                                                  \
    if (dev_ >= 0 && dev_ < 64 && !done_[dev_]) {                                                  \
      SGP_HIP(hipFuncSetAttribute((const void*)(func), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                  (int)(bytes)));                                                  \
      done_[dev_] = true;                                                                          \
    }                                                                                              \
  } while (0)

// ---- launchers implemented in the .hip files -------------------------------------------
// kernelmatrix.hip
int launch_assemble_block(double* K, long ld, long r0, long nr, long c0, long nc,
                          const DevTerm* d_terms, int nterms, int lds_dim_sum, int lower_only,
                          int accumulate, int noise_kind, double sigma2, const double* d_noise_diag,
                          long tile_r_first, long tile_c_first, long tile_r_cnt, long tile_c_cnt,
                          hipStream_t s);
int assemble_terms_per_launch(int dmax);  // how many terms one launch_assemble_block may carry (LDS)
int launch_rows_dot(const double* rows, long ld, long nrows, long nc, const double* zrow, double* sumsq,
                    double* dot, hipStream_t s);
int launch_zero_rows(double* A, long ld, long r0, long r1, long nc, hipStream_t s);
int launch_fill_pad(double* K, long ld, long N, long n_pad, long c0, long nc, long m_tot,
                    long row_lo, hipStream_t s);
int launch_border_rows(double* A, long ld, long n_pad, long N, long c0, long nc, const double* dY,
                       long ldy, long ncols, const double* d_mean, hipStream_t s);
int launch_diag_terms(double* out, long n, const DevTerm* d_terms, int nterms, hipStream_t s);
int launch_add_dense(double* K, long ld, const double* S, long lds, long N, int lower_only,
                     hipStream_t s);
int launch_add_dense_cols(double* P, long ldp, const double* S, long lds, long c0, long w, long N, hipStream_t s);
int launch_mirror_lower(double* K, long ld, long N, hipStream_t s);

// gemm_nt.hip : C = beta*C + alpha * A B'   (A: M x K, B: Nc x K, all column-major)
// tiles (tr, tc) with tr < tc + mask_off are skipped.  kcap_mode: per-tile K limited to
// (tc + 1) * 128 + kcap_off (used by L*Z products where only the lower triangle of B is valid).
int launch_gemm_nt(const double* A, long lda, const double* B, long ldb, double* C, long ldc,
                   long M, long Nc, long K, double alpha, double beta, long mask_off,
                   int kcap_mode, long kcap_off, hipStream_t s);
int launch_gemm_nt_update(const double* P, long ldp, double* C, long ldc, long M, long Nc, long K, hipStream_t s);
// the same lower update with potrf_diag of tile (0, 0) -- the next diagonal block -- fused into its workgroup
int launch_gemm_nt_potrf(const double* P, long ldp, double* C, long ldc, long M, long Nc, long K, int outer,
                         int handoff, double* d_invd, double* d_logdet_slot, int* d_info, long gcol0,
                         hipStream_t s);
// Structural zeros (round 4).  A Stheno programme with independent components has EXACT zero blocks in its covariance (no
// term connects the two processes: cross.jl gives zeros), and so has its Cholesky factor; the reference's dense LAPACK path
// multiplies them out.  capi.hip: sz_build derives the tile-level pattern of the factor (symbolic factorisation with
// fill-in) and the factorisation skips every tile product one of whose operands is structurally zero -- the skipped
// products are exact zeros, so the factor keeps its bits.  nz: one row of `words` 64-bit words per tile row of the
// bordered matrix, bit k = tile (row, k) of the factor may be non-zero.
typedef unsigned long long sz_word;
struct TileSkip {      // per launch of a lower update C -= P P': C's first tile row / column, the k tiles the panel covers
  const sz_word* nz = nullptr;
  int words = 0, tr0 = 0, tc0 = 0, kt0 = 0, kt1 = 0;
  // big launches: the live tiles of every XCD compacted to the front of its id sequence (tile_compact_kernel): cmap[x] =
  // live ids of XCD x, cmap[16 + x * cstride + pos] = the pos-th live id / 8 -- the live tiles keep the order, and with it
  // the lock-step operand sharing, of the dense enumeration
  const int* cmap = nullptr;
  int cstride = 0;
  // round 5 (the gradient's C^-1 = inv(L)' inv(L), launch_gemm_nt_uut): >= 0: pattern row need0 + tr says which tiles (tr, tc)
  // of the RESULT anyone reads -- the tiles of block pairs with terms (+ the diagonal); the others are not computed
  int need0 = -1;
};
// while set (chol_bordered's scope; per host thread), the lower updates launched on tiles of `base` skip dead tiles;
// scratch (optional, only for launches that are ordered on one stream): room for the compacted id maps
// tile0 (round 5): the global tile coordinate of base[0] -- 0 for the bordered matrix itself, c0 / 128 for a PACKED column
// panel of the sharded factorisation (its element [0] is (row c0, column c0) of the matrix), so that the launches of a
// panel's own factorisation find their tiles in the pattern too
// (scratch / scratch_ints: room for the compacted live-tile id map of a launch; stream2 / scratch2: a second map of the same
// size for the launches of that other stream -- launches of one stream are ordered, so a map per stream is enough)
// ncols: the columns of the matrix at `base` (pointers beyond base + ld * ncols are not part of it and get no pattern)
void gemm_set_structure(const double* base, long ld, const sz_word* d_nz, int words, long ncols = 0, int* scratch = nullptr,
                        long scratch_ints = 0, long tile0 = 0, hipStream_t stream2 = nullptr, int* scratch2 = nullptr);
// the panel solve X <- X inv(L_kk)' over rows of the structured matrix: 128-row tiles of X whose tile (row, k) is structurally
// zero are left alone (they hold the exact zeros the assembly wrote).  nz == nullptr: every row is solved.
struct StripSkip {
  const sz_word* nz = nullptr;
  int words = 0, tr0 = 0, kt = 0;   // tile row of X's first row, k tile of the block column
};
StripSkip strip_skip_for(const double* X, long ldx);   // from the per-thread structure record (gemm_nt.hip), or an empty one
// the fp32 instantiation (f32.hip, round 6): the same record over a float matrix; its update launches keep the dense tile
// enumeration (no compacted id maps) and skip / trim per workgroup
void gemm_set_structure_f32(const float* base, long ld, const sz_word* d_nz, int words, long ncols);
TileSkip gemm_skip_for_f32(const float* P, long ldp, const float* C, long ldc, long K);
StripSkip strip_skip_for_f32(const float* X, long ldx);
// chol_df.hip: the whole bordered factorisation (lower tiles of the n_pad columns + rows n_pad .. m_tot) in one launch of
// persistent workgroups; d_state: df_state_words(m_tot, 1) ints, d_invall: n_pad / 128 x 2048 doubles
constexpr int SGP_DF_TIMEOUT = -77;   // *info when a dependency wait inside the kernel ran into its bound
// One matrix of a launch (a batch holds up to DF_MAX_BATCH equally shaped ones: launch_chol_dataflow_batch)
constexpr int DF_MAX_BATCH = 16, DF_MAX_EXT = 2;
struct DfProb {
  double* A;        // m_tot x n_pad, column-major, lower tiles + bordered rows
  double* invall;   // n_pad / 128 x 2048 doubles: inverse 16x16 diagonal blocks of every 128-block
  double* slots;    // n_pad / 128 logdet contributions
  int* info;
};
// A factored column panel LEFT of the matrix a panel launch works on (round 6, the sharded factorisation): base points at the
// source's element (row of A's first row, first column of the source), kt0 = global tile column of that first column, kt =
// its tile columns.  Every task contracts the sources first, in the order given (k ascending).
struct DfExt {
  const double* base;
  long ld;
  int kt0, kt;
};
struct DfPanel {    // launch_chol_dataflow's optional extension: factor only the first n_fact columns (the others: update-only)
  long n_fact = 0;
  int n_ext = 0;
  DfExt ext[DF_MAX_EXT];
};
int launch_chol_dataflow(double* A, long ld, long n_pad, long m_tot, int* d_state, double* d_invall, double* d_slots,
                         int* d_info, int n_wg, double timeout_s, hipStream_t s, long long* d_stats = nullptr,
                         long long* d_cols = nullptr, int fat = 0, const sz_word* d_nz = nullptr, int nz_words = 0,
                         long gcol_base = 0, const DfPanel* px = nullptr);
int launch_chol_dataflow_batch(const DfProb* probs, int nb, long ld, long n_pad, long m_tot, int* d_state, int n_wg,
                               double timeout_s, int fat, hipStream_t s);
long df_state_words(long m_tot, int nb);   // ints of d_state a launch needs
constexpr long SGP_DF_STATE_WORDS = 16;   // state words ahead of the per-tile-row progress counters
int launch_gemm_nt_stamps(const double* P, long ldp, double* C, long ldc, long M, long Nc, long K, long long* dbg,
                          long* n_ids, hipStream_t s);   // bench: per-workgroup phase stamps of one lower update
// batched, segmented trailing update (gemm_nt.hip: gemm_nt_seg_kernel): destination panels, each with a range of source panels
constexpr int SEG_MAX_SRC = 8, SEG_MAX_DST = 16;
struct SegSrc {      // a factored column panel, packed: element (global row r, local column k) at base[(r - row0) + k * ld]
  const double* base;
  long ld, row0;
  int w;
  long k0 = -1;      // global column of the first column (a sub-panel: row0 + its offset); -1: row0.  Only the structural-
                     // zero test reads it.
};
struct SegDst {      // an owned column panel, packed: element (global row r, global column c) at C[(r - c0) + (c - c0) * ldc]
  double* C;
  long ldc, c0;
  int w, s_first, s_count;
  unsigned id0;      // (filled in by the launcher)
};
struct SegBatch {
  SegSrc src[SEG_MAX_SRC];
  SegDst dst[SEG_MAX_DST];
  int n_dst;
  long m_tot;
  const sz_word* nz = nullptr;   // structural zeros (above): a tile skips the sources all of whose k tiles are dead for it
  int nz_words = 0;
  // round 6: room for the compacted live-tile id map of a big structured launch (launches of ONE stream only: they are
  // ordered); nullptr: the dead ids leave where they stand.  cmap / cstride: filled in by the launcher.
  int* map_scratch = nullptr;
  long map_ints = 0, map_min_ids = 4096;   // (smaller launches are not worth the extra kernel; tests lower the bound)
  const int* cmap = nullptr;
  int cstride = 0;
};
int launch_gemm_nt_seg(SegBatch& b, hipStream_t s, long* n_ids = nullptr);
int launch_gemm_nt_cin(const double* A, long lda, const double* B, long ldb, const double* Cin, long ldcin,
                       double* C, long ldc, long M, long Nc, long K, double alpha, double beta,
                       hipStream_t s);
int launch_gemm_nt_lz(const double* L, long ldl, const double* Zt, long ldz, double* C, long ldc, long n,
                      long ns, double beta, hipStream_t s);
int launch_gemm_nt_lz_k(const double* L, long ldl, const double* Zt, long ldz, double* C, long ldc, long n,
                        long ns, long K, double beta, hipStream_t s);
int launch_gemm_nt_uut(const double* X, long ldx, double* C, long ldc, long n, hipStream_t s, const TileSkip* sk = nullptr);
int launch_gemm_nt_splitk(const double* A, long lda, const double* B, long ldb, double* Cpart, long ldc,
                          long M, long Nc, long K, int nsplit, long part_stride, int lower_only,
                          hipStream_t s);
int launch_splitk_reduce(const double* part, long part_stride, int nsplit, double* C, long ldc, long M,
                         long Nc, double alpha, double beta, int lower_only, hipStream_t s, long K = 0);
long splitk_sub(long M, long K);                 // further k split of the Gram product's leftover tiles (gemm_nt.hip)
long splitk_slabs(long M, long K, int nsplit);   // slabs launch_gemm_nt_splitk writes for this shape

// grad.hip
int launch_grad_block(const double* Kinv, long ldk, const double* alpha, long r0, long nr, long c0, long nc,
                      const DevTerm* d_terms, int nterms, int dmax, long trf, long tcf, long trc, long tcc,
                      double* partials, double* out_coef, double* out_scale, hipStream_t s, int accumulate = 0,
                      long clo = -(1L << 60), long chi = (1L << 60));   // accumulate into out_*; column window [clo, chi)
int launch_diag_grad(const double* w, long n, const DevTerm* d_terms, int nterms, double* out_coef,
                     double* out_scale, hipStream_t s);
int launch_diag_scale_grad(const double* w, long n, const DevTerm& T, double* out_rs, double* out_cs, hipStream_t s);
int launch_vfe_zs(const double* B, const double* Binv, const double* u, double* Z, double* S, long m,
                  hipStream_t s);
int launch_vfe_rowstats(const double* R, long ld, const double* RZ, long ldrz, const double* u,
                        const double* delta, const double* rsig, const double* var_x, long N, long m,
                        double* gy, double* gsy, hipStream_t s);
int launch_vfe_gxz(double* E, long ld, const double* delta, const double* ut, const double* rsig, long nrows,
                   long m, hipStream_t s);
int launch_grad_inputs(const double* Gm, long sr, long sc, const double* alpha, long r0, long nr, long c0, long nc,
                       const DevTerm& T, int dmax, double scale, double* gx, hipStream_t s, double* gsv = nullptr);
int launch_diag_grad_inputs(const double* w, long n, const DevTerm& T, double* gxr, double* gxc, hipStream_t s);
int launch_grad_border(double* A, long ld, long n_pad, long N, const double* y, const double* mean,
                       long nrows, hipStream_t s);
int launch_grad_noise(const double* Kinv, long ldk, const double* alpha, long N, int diag, double* out,
                      hipStream_t s);
int launch_grad_noise_dense(const double* Kinv, long ldk, const double* alpha, long N, double* out, hipStream_t s);
int launch_grad_noise_dense_cols(const double* Kinv, long ldk, const double* alpha, long N, long c0, long w, double* out, long ldo,
                                 hipStream_t s);

// potrf.hip
int launch_potrf_diag(double* A, long ld, double* d_invd, double* d_logdet_slot, int* d_info,
                      long gcol0, hipStream_t s);
int launch_potrf_diag_dbg(double* A, long ld, double* d_invd, double* d_logdet_slot, int* d_info, long long* dbg,
                          hipStream_t s);
int launch_panel_solve(double* X, long ldx, long rows, const double* L, long ldl, const double* inv,
                       long inv_cstride, long inv_kstride, hipStream_t s, const StripSkip* sk = nullptr, long div = 256);
// fp32 storage, fp64 arithmetic: the panel kernels of the fp32 instantiation (f32.hip)
int launch_potrf_diag_f32(float* A, long ld, double* d_invd, double* d_logdet_slot, int* d_info, long gcol0,
                          hipStream_t s);
int launch_panel_solve_f32(float* X, long ldx, long rows, const float* L, long ldl, const double* inv,
                           long inv_cstride, long inv_kstride, hipStream_t s, const StripSkip* sk = nullptr);

// reduce.hip
int launch_rowsumsq(const double* rows, long ld, long nc, long nrows, double* out, int accumulate,
                    hipStream_t s);
int launch_sum_array(const double* in, long n, double* out, hipStream_t s);
int launch_logpdf_final(const double* d_logdet, const double* d_sq, long N, long ncols,
                        double* d_out, hipStream_t s);
int launch_colsumsq_sub(const double* V, long ld, long nrows, long ncols, const double* prior,
                        double* out, double sign, hipStream_t s);
int launch_gemv_rows(const double* rows, long ld, long nrows, long nc, const double* z, long ldz,
                     const double* add, double* out, hipStream_t s, int upper_tri = 0);
// transposition + column sums in one pass (the ELBO's chunked pipeline): dst = (diag(row_scale) src)', dots[c] (+)= sum_r v delta[r],
// sq[c] (+)= sum_r v^2; scratch: transpose_colsum_scratch(nr, nc) doubles
long transpose_colsum_scratch(long nr, long nc);
int launch_transpose_colsum(const double* src, long lds, long nr, long nc, double* dst, long ldd, const double* row_scale,
                            const double* delta, double* dots, double* sq, int accumulate, double* scratch, hipStream_t s);
int launch_transpose_add(const double* src, long lds, long nr, long nc, double* dst, long ldd,
                         const double* add_vec, hipStream_t s);   // row_scale: src row r is multiplied by row_scale[r]
int launch_scale_rows(double* rows, long ld, long nrows, long nc, const double* scale,
                      hipStream_t s);

}  // namespace sgp
