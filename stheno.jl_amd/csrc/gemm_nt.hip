// fp64 MFMA GEMM kernels for gfx950.
//
// Instruction choice (measured on MI355X, profiles/r01_microbench.md): v_mfma_f64_16x16x4_f64
// issues only every ~96 cycles per SIMD (49 TF/s chip-wide, 62 % of the 78.6 TF/s datasheet
// rate) while the 4-block v_mfma_f64_4x4x4_4b_f64 issues every 16 cycles (71 TF/s, 90 %).
// The GEMM therefore runs on the 4x4x4 form.  Its lane maps (probed on hardware,
// tools/gpu_probe44.py; b = (l >> 2) & 3 is the block):
//   A-operand lane l holds A_b[i = l & 3][k = l >> 4]
//   B-operand lane l holds B_b[k = l >> 4][j = l & 3]
//   result    lane l holds D_b[i = l >> 4][j = l & 3]
// Feeding every block the same A (4 columns of C) and block b rows 4b..4b+3 of a 16-row
// fragment as B makes one instruction a 16(row) x 4(col) x 4(k) update whose result lanes run
// along the contiguous row dimension of C (row = l & 15, col = l >> 4).
//
// gemm_nt:  C = beta*C + alpha * A * B'   A: M x K, B: Nc x K, C: M x Nc, column-major.
//   This single kernel is the SYRK/GEMM trailing update of the blocked Cholesky
//   (alpha=-1, beta=1, lower tiles only), the panel TRSM (B = inv(L11), beta=0), the
//   row-bordered forward substitutions of logpdf / posterior / elbo, and L*Z for rand.
//   It replaces LAPACK dpotrf's dsyrk/dgemm/dtrsm calls under LinearAlgebra.cholesky on the
//   reference path.
//
// Tiling: one 256-thread workgroup (4 waves) per 128x128 tile of C, each wave a 64x64
// quadrant = 4x4 MFMA tiles (128 accumulator VGPRs).  K is consumed in chunks of 16 staged
// through LDS (double-buffered, register prefetch of the next chunk).  Both operands are
// "row index contiguous" so the staging copy is 1 KiB-per-wave coalesced, and the LDS leading
// dimension 144 makes every ds_read_b64 operand fetch conflict-free (common.h).
// The MFMA is issued as D = Bop' x Aop so that the result lane map (n = lane & 15) runs
// along the contiguous (row) dimension of C: every store instruction writes 4 x 128-byte runs.
#include "common.h"

namespace sgp {

constexpr int KB = 16;  // K chunk per LDS stage

// Shared inner product of one LDS stage: NJ column fragments (4 cols each) x 4 row fragments
// (16 rows each) per wave, K = 16 in four k-steps of 4.
#define SGP_COMPUTE(buf_, NJ_, COL0_)                                            \
  {                                                                              \
    const double* pa = &sA[buf_][wr * 64 + l15];                                 \
    const double* pb = &sB[buf_][(COL0_) + l3];                                  \
    _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                           \
      const int kk = ks * 4 + lq;                                                \
      double a_r[4], b_c[NJ_];                                                   \
      _Pragma("unroll") for (int i = 0; i < 4; ++i) a_r[i] = pa[kk * LDS_LD + i * 16]; \
      _Pragma("unroll") for (int j = 0; j < NJ_; ++j) b_c[j] = pb[kk * LDS_LD + j * 4]; \
      _Pragma("unroll") for (int j = 0; j < NJ_; ++j)                            \
          _Pragma("unroll") for (int i = 0; i < 4; ++i)                          \
              acc[j][i] = mfma44_f64(b_c[j], a_r[i], acc[j][i]);                 \
    }                                                                            \
  }

// WN = number of wave columns: 2 -> 4 waves (64x64 per wave), 4 -> 8 waves (64x32 per wave,
// half the registers, 4 waves per SIMD at 2 workgroups per CU: more latency hiding).
template <bool KCAP, int WN, int ABL = 0>
__global__ __launch_bounds__(128 * WN, WN) void gemm_nt_kernel(const double* A, long lda,
                                                               const double* B, long ldb,
                                                               double* C, long ldc, long K,
                                                               double alpha, double beta,
                                                               long mask_off, long kcap_off,
                                                               long n_tr, long n_tc) {
  constexpr int NT = 128 * WN;        // threads
  constexpr int NJ = (128 / WN) / 4;  // column fragments (4 cols each) per wave
  constexpr int WCOLS = 128 / WN;     // columns per wave
  constexpr int NU = 1024 / NT;       // double2 staging units per thread per operand
  long tr, tc;
  {
    // XCD-aware, load-balanced tile order.  Hardware places workgroup id on XCD id % 8.
    // XCD x owns the tile rows tr == x (mod 8): every A row-panel is read through exactly one
    // XCD's L2, and at any position in the grid all 8 XCDs see (almost) the same lower-
    // triangular mask, so the live work stays balanced however the dispatcher paces XCDs.
    // Inside an XCD the order is 8 owned rows x 8 tile columns, row fastest: 64 consecutive
    // workgroups form a patch sharing 8 A panels and 8 B panels in that XCD's L2.
    const long id = (long)blockIdx.x;
    const long xcd = id & 7, k = id >> 3;
    const long gs = 8 * n_tc;
    const long jgroup = k / gs, within = k % gs;
    const long j = jgroup * 8 + (within & 7);
    tr = 8 * j + xcd;
    tc = within >> 3;
    if (tr >= n_tr) return;
  }
  if (tr < tc + mask_off) return;
  long Keff = K;
  if (KCAP) {
    long cap = (tc + 1) * TILE + kcap_off;
    if (cap < Keff) Keff = cap;
    if (Keff <= 0) Keff = 0;
  }

  __shared__ __attribute__((aligned(16))) double sA[2][KB * LDS_LD];
  __shared__ __attribute__((aligned(16))) double sB[2][KB * LDS_LD];

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int w = t >> 6;
  const int wr = w / WN, wc = w % WN;
  const int l15 = lane & 15, lq = lane >> 4, l3 = lane & 3;

  const double* Ag = A + tr * TILE;
  const double* Bg = B + tc * TILE;

  // staging map: unit i of thread t -> column (t>>6) + (NT/64) i of the chunk, rows 2*(t&63), +1
  const int scol = t >> 6;
  const int srow = 2 * (t & 63);
  double2 ra[NU], rb[NU];

  // acc[j][i] = C[row = r0 + 16 i + l15][col = c0 + 4 j + lq].  The accumulators are seeded
  // with (beta / alpha) * C so that the old tile is read in the prologue (its latency hides
  // behind the first operand loads and the co-resident workgroup's MFMAs) and the epilogue is
  // store-only: C_new = alpha * (A B' + (beta / alpha) C).
  double* Cg = C + (tr * TILE + wr * 64 + l15) + (tc * TILE + wc * WCOLS + lq) * ldc;
  double acc[NJ][4];
  if (beta != 0.0) {
    const double seed = beta / alpha;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[j][i] = Cg[i * 16 + (long)(j * 4) * ldc];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[j][i] *= seed;
  } else {
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[j][i] = 0.0;
  }

#define SGP_GLOAD(k0_)                                                           \
  _Pragma("unroll") for (int i = 0; i < NU; ++i) {                               \
    long col = (k0_) + scol + (NT / 64) * i;                                     \
    ra[i] = *reinterpret_cast<const double2*>(Ag + srow + col * lda);            \
    rb[i] = *reinterpret_cast<const double2*>(Bg + srow + col * ldb);            \
  }
#define SGP_SSTORE(buf_)                                                         \
  _Pragma("unroll") for (int i = 0; i < NU; ++i) {                               \
    int col = scol + (NT / 64) * i;                                              \
    *reinterpret_cast<double2*>(&sA[buf_][col * LDS_LD + srow]) = ra[i];         \
    *reinterpret_cast<double2*>(&sB[buf_][col * LDS_LD + srow]) = rb[i];         \
  }

  if (Keff > 0) {
    SGP_GLOAD(0);
    SGP_SSTORE(0);
    __syncthreads();
    int buf = 0;
    // ABL != 0 only in the bench-only ablation builds (tools/gpu_gemm_abl.py): 1 = no global
    // loads / LDS stores, 2 = no LDS operand reads, 3 = no barrier, 4 = 1+2 (MFMA + barrier only)
    double inv_a[4], inv_b[NJ];
    if (ABL == 2 || ABL == 4) {
#pragma unroll
      for (int i = 0; i < 4; ++i) inv_a[i] = sA[0][wr * 64 + l15 + i * 16];
#pragma unroll
      for (int j = 0; j < NJ; ++j) inv_b[j] = sB[0][wc * WCOLS + l3 + j * 4];
    }
    for (long k0 = KB; k0 < Keff; k0 += KB) {
      if (ABL != 1 && ABL != 4) { SGP_GLOAD(k0); }
      if (ABL == 2 || ABL == 4) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
          for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[j][i] = mfma44_f64(inv_b[j], inv_a[i], acc[j][i]);
      } else {
        SGP_COMPUTE(buf, NJ, wc * WCOLS);
      }
      if (ABL != 1 && ABL != 4) { SGP_SSTORE(buf ^ 1); }
      if (ABL != 3) __syncthreads();
      buf ^= 1;
    }
    SGP_COMPUTE(buf, NJ, wc * WCOLS);
  }
#undef SGP_GLOAD
#undef SGP_SSTORE

  // epilogue (store only): every store instruction writes 4 columns x 16 consecutive rows
  // (4 x 128-byte runs).
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) Cg[i * 16 + (long)(j * 4) * ldc] = alpha * acc[j][i];
}

static int g_gemm_wn = 4;  // default wave layout; sgp_bench_gemm flips it for A/B comparisons
void set_gemm_wave_layout(int wn) { g_gemm_wn = (wn == 2) ? 2 : 4; }

int launch_gemm_nt(const double* A, long lda, const double* B, long ldb, double* C, long ldc,
                   long M, long Nc, long K, double alpha, double beta, long mask_off,
                   int kcap_mode, long kcap_off, hipStream_t s) {
  if (M <= 0 || Nc <= 0) return 0;
  if (M % TILE || Nc % TILE || K % KB) {
    set_error("gemm_nt: M, Nc must be multiples of 128 and K of 16");
    return -1;
  }
  long n_tr = M / TILE, n_tc = Nc / TILE;
  long groups = (n_tr + 7) / 8;
  long groups_pad = (groups + 7) / 8 * 8;
  long total = groups_pad * 8 * n_tc;
  dim3 grid((unsigned)total);
#define SGP_LAUNCH(KC, WNV)                                                                       \
  hipLaunchKernelGGL((gemm_nt_kernel<KC, WNV>), grid, dim3(128 * WNV), 0, s, A, lda, B, ldb, C, ldc, \
                     K, alpha, beta, mask_off, kcap_off, n_tr, n_tc)
  if (kcap_mode >= 16) {  // bench-only ablations of the 8-wave kernel
#define SGP_LAUNCH_ABL(AB)                                                                        \
  hipLaunchKernelGGL((gemm_nt_kernel<false, 4, AB>), grid, dim3(512), 0, s, A, lda, B, ldb, C, ldc, K, \
                     alpha, beta, mask_off, kcap_off, n_tr, n_tc)
    switch (kcap_mode >> 4) {
      case 1: SGP_LAUNCH_ABL(1); break;
      case 2: SGP_LAUNCH_ABL(2); break;
      case 3: SGP_LAUNCH_ABL(3); break;
      default: SGP_LAUNCH_ABL(4); break;
    }
#undef SGP_LAUNCH_ABL
  } else if (g_gemm_wn == 2) {
    if (kcap_mode) SGP_LAUNCH(true, 2); else SGP_LAUNCH(false, 2);
  } else {
    if (kcap_mode) SGP_LAUNCH(true, 4); else SGP_LAUNCH(false, 4);
  }
#undef SGP_LAUNCH
  SGP_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------------------
// gemm_tn: C(M x Nc) = beta*C + alpha * A' * B,  A: K x M, B: K x Nc  (contraction over the
// contiguous dimension).  Used for the VFE/ELBO Gram matrix A A' + I where A' is stored as
// bordered rows (N x M) and N is huge.  Staging transposes through
// LDS: chunk of 16 k-rows x 128 columns per operand.
// ---------------------------------------------------------------------------------------
constexpr int TN_LD = 130;  // LDS ld for [col][k] layout: element (k, c) at c*... see below

__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(const double* A, long lda,
                                                         const double* B, long ldb, double* C,
                                                         long ldc, long K, double alpha,
                                                         double beta, int lower_only,
                                                         long k_per_split, double* Cpart,
                                                         long part_stride) {
  const long tr = blockIdx.x, tc = blockIdx.y;
  const long ksplit = blockIdx.z;
  if (lower_only && tr < tc) return;
  // LDS layout [k][row] with ld LDS_LD, same as gemm_nt, so the compute loop is identical;
  // the global read is the transposing part: thread reads along k (contiguous) for one column.
  __shared__ __attribute__((aligned(16))) double sA[2][KB * LDS_LD];
  __shared__ __attribute__((aligned(16))) double sB[2][KB * LDS_LD];
  const int t = threadIdx.x;
  const int lane = t & 63, w = t >> 6;
  const int wr = w >> 1, wc = w & 1;
  const int l15 = lane & 15, lq = lane >> 4, l3 = lane & 3;
  const long kbeg = ksplit * k_per_split;
  long kend = kbeg + k_per_split;
  if (kend > K) kend = K;

  // staging: chunk = 16 (k) x 128 (cols) per operand = 2048 doubles; thread t handles
  // column c = t & 127 and k-half h = t >> 7 (8 consecutive k) -> 4 double2 loads.
  const int sc = t & 127, sh = t >> 7;
  const double* Ag = A + (tr * TILE + sc) * lda;
  const double* Bg = B + (tc * TILE + sc) * ldb;
  double2 ra[4], rb[4];
  double acc[16][4];  // acc[j][i] = C[row = r0 + 16 i + l15][col = c0 + 4 j + lq]
#pragma unroll
  for (int j = 0; j < 16; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[j][i] = 0.0;

#define SGP_TN_GLOAD(k0_)                                                        \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                \
    long k = (k0_) + sh * 8 + 2 * i;                                             \
    ra[i] = *reinterpret_cast<const double2*>(Ag + k);                           \
    rb[i] = *reinterpret_cast<const double2*>(Bg + k);                           \
  }
#define SGP_TN_SSTORE(buf_)                                                      \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                \
    int k = sh * 8 + 2 * i;                                                      \
    sA[buf_][k * LDS_LD + sc] = ra[i].x;                                         \
    sA[buf_][(k + 1) * LDS_LD + sc] = ra[i].y;                                   \
    sB[buf_][k * LDS_LD + sc] = rb[i].x;                                         \
    sB[buf_][(k + 1) * LDS_LD + sc] = rb[i].y;                                   \
  }
  if (kbeg < kend) {
    SGP_TN_GLOAD(kbeg);
    SGP_TN_SSTORE(0);
    __syncthreads();
    int buf = 0;
    for (long k0 = kbeg + KB; k0 < kend; k0 += KB) {
      SGP_TN_GLOAD(k0);
      SGP_COMPUTE(buf, 16, wc * 64);
      SGP_TN_SSTORE(buf ^ 1);
      __syncthreads();
      buf ^= 1;
    }
    SGP_COMPUTE(buf, 16, wc * 64);
  }
  if (Cpart) {
    // split-K: write the partial tile; a second kernel reduces in fixed order (deterministic)
    double* P = Cpart + ksplit * part_stride + (tr * TILE + wr * 64 + l15) +
                (tc * TILE + wc * 64 + lq) * ldc;
#pragma unroll
    for (int j = 0; j < 16; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) P[i * 16 + (long)(j * 4) * ldc] = acc[j][i];
    return;
  }
  double* Cg = C + (tr * TILE + wr * 64 + l15) + (tc * TILE + wc * 64 + lq) * ldc;
#pragma unroll
  for (int j = 0; j < 16; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      double* p = Cg + i * 16 + (long)(j * 4) * ldc;
      double old = (beta == 0.0) ? 0.0 : beta * (*p);
      *p = old + alpha * acc[j][i];
    }
}

__global__ void splitk_reduce_kernel(const double* part, long part_stride, int nsplit, double* C,
                                     long ldc, long M, long Nc, double alpha, double beta,
                                     int lower_only) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= M * Nc) return;
  long r = idx % M, c = idx / M;
  if (lower_only && (r / TILE) < (c / TILE)) return;
  double s = 0.0;
  for (int k = 0; k < nsplit; ++k) s += part[k * part_stride + r + c * ldc];
  double* p = C + r + c * ldc;
  double old = (beta == 0.0) ? 0.0 : beta * (*p);
  *p = old + alpha * s;
}

// workspace for split-K partials is provided by the caller through a static hook
static double* g_tn_ws = nullptr;
static size_t g_tn_ws_bytes = 0;
void set_gemm_tn_workspace(double* ws, size_t bytes) {
  g_tn_ws = ws;
  g_tn_ws_bytes = bytes;
}

int launch_gemm_tn(const double* A, long lda, const double* B, long ldb, double* C, long ldc,
                   long M, long Nc, long K, double alpha, double beta, int lower_only,
                   hipStream_t s) {
  if (M <= 0 || Nc <= 0) return 0;
  if (M % TILE || Nc % TILE || K % KB) {
    set_error("gemm_tn: M, Nc must be multiples of 128 and K of 16");
    return -1;
  }
  long n_tr = M / TILE, n_tc = Nc / TILE;
  long tiles = lower_only ? n_tr * (n_tr + 1) / 2 : n_tr * n_tc;
  // split K so that there are >= ~1024 workgroups when the output is small and K is huge
  int nsplit = 1;
  if (tiles < 1024 && K >= 8192) {
    nsplit = (int)((1024 + tiles - 1) / tiles);
    long maxsplit = K / 2048;
    if (nsplit > maxsplit) nsplit = (int)maxsplit;
    if (nsplit < 1) nsplit = 1;
    if (nsplit > 64) nsplit = 64;
  }
  long k_per = ((K + nsplit - 1) / nsplit + KB - 1) / KB * KB;
  nsplit = (int)((K + k_per - 1) / k_per);
  double* part = nullptr;
  long stride = 0;
  if (nsplit > 1) {
    if (ldc != M) {
      // partial slabs reuse C's layout (ldc x Nc)
    }
    stride = ldc * Nc;
    size_t need = (size_t)nsplit * stride * sizeof(double);
    if (need > g_tn_ws_bytes) {
      nsplit = 1;
      k_per = K;
    } else {
      part = g_tn_ws;
    }
  }
  dim3 grid((unsigned)n_tr, (unsigned)n_tc, (unsigned)nsplit), block(256);
  hipLaunchKernelGGL(gemm_tn_kernel, grid, block, 0, s, A, lda, B, ldb, C, ldc, K, alpha, beta,
                     lower_only, k_per, part, stride);
  SGP_HIP(hipGetLastError());
  if (part) {
    long tot = M * Nc;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s,
                       part, stride, nsplit, C, ldc, M, Nc, alpha, beta, lower_only);
    SGP_HIP(hipGetLastError());
  }
  return 0;
}

}  // namespace sgp
