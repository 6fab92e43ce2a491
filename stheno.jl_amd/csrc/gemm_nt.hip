// fp64 MFMA GEMM kernels for gfx950 (v_mfma_f64_16x16x4_f64).
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

template <bool KCAP>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(const double* A, long lda,
                                                         const double* B, long ldb, double* C,
                                                         long ldc, long K, double alpha,
                                                         double beta, long mask_off, long kcap_off,
                                                         int xcd_swizzle, long n_tr, long n_tc) {
  long tr, tc;
  {
    // XCD-aware remap: consecutive workgroup ids land on different XCDs (id % 8); give each
    // XCD a contiguous run of tiles in column-major tile order so that tiles sharing a B
    // row-panel (same tc) and neighbouring A panels hit the same L2.
    long id = (long)blockIdx.x;
    long total = n_tr * n_tc;
    if (xcd_swizzle) {
      long q = total / 8, r = total % 8;
      long xcd = id % 8, k = id / 8;
      long base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
      id = base + k;
    }
    // grouped order: 8 tile-rows x all tile-cols per group, row index fastest, so 64
    // consecutive ids form an 8x8 patch sharing 8 A panels and 8 B panels (L2 reuse).
    const long GM = 8;
    long group_size = GM * n_tc;
    long g = id / group_size;
    long first_tr = g * GM;
    long gm = n_tr - first_tr < GM ? n_tr - first_tr : GM;
    long rem = id % group_size;
    tr = first_tr + rem % gm;
    tc = rem / gm;
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
  const int wr = w >> 1, wc = w & 1;
  const int l15 = lane & 15, lq = lane >> 4;

  const double* Ag = A + tr * TILE;
  const double* Bg = B + tc * TILE;

  // staging map: unit i of thread t -> column (t>>6)+4i of the chunk, rows 2*(t&63), +1
  const int scol = t >> 6;
  const int srow = 2 * (t & 63);
  double2 ra[4], rb[4];

  d4 acc[4][4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[j][i] = (d4){0.0, 0.0, 0.0, 0.0};

#define SGP_GLOAD(k0_)                                                           \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                \
    long col = (k0_) + scol + 4 * i;                                             \
    ra[i] = *reinterpret_cast<const double2*>(Ag + srow + col * lda);            \
    rb[i] = *reinterpret_cast<const double2*>(Bg + srow + col * ldb);            \
  }
#define SGP_SSTORE(buf_)                                                         \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                \
    int col = scol + 4 * i;                                                      \
    *reinterpret_cast<double2*>(&sA[buf_][col * LDS_LD + srow]) = ra[i];         \
    *reinterpret_cast<double2*>(&sB[buf_][col * LDS_LD + srow]) = rb[i];         \
  }
#define SGP_COMPUTE(buf_)                                                        \
  {                                                                              \
    const double* pa = &sA[buf_][wr * 64 + l15];                                 \
    const double* pb = &sB[buf_][wc * 64 + l15];                                 \
    _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                           \
      const int kk = ks * 4 + lq;                                                \
      double a_r[4], b_c[4];                                                     \
      _Pragma("unroll") for (int i = 0; i < 4; ++i) a_r[i] = pa[kk * LDS_LD + i * 16]; \
      _Pragma("unroll") for (int j = 0; j < 4; ++j) b_c[j] = pb[kk * LDS_LD + j * 16]; \
      _Pragma("unroll") for (int j = 0; j < 4; ++j)                              \
          _Pragma("unroll") for (int i = 0; i < 4; ++i)                          \
              acc[j][i] = mfma_f64(b_c[j], a_r[i], acc[j][i]);                   \
    }                                                                            \
  }

  if (Keff > 0) {
    SGP_GLOAD(0);
    SGP_SSTORE(0);
    __syncthreads();
    int buf = 0;
    for (long k0 = KB; k0 < Keff; k0 += KB) {
      SGP_GLOAD(k0);
      SGP_COMPUTE(buf);
      SGP_SSTORE(buf ^ 1);
      __syncthreads();
      buf ^= 1;
    }
    SGP_COMPUTE(buf);
  }

  // epilogue: lane holds C[row = r0 + i*16 + l15][col = c0 + j*16 + lq + 4*reg]
  double* Cg = C + (tr * TILE + wr * 64 + l15) + (tc * TILE + wc * 64 + lq) * ldc;
  if (beta == 0.0) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) Cg[i * 16 + (long)(j * 16 + 4 * r) * ldc] = alpha * acc[j][i][r];
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        double cv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) cv[r] = Cg[i * 16 + (long)(j * 16 + 4 * r) * ldc];
#pragma unroll
        for (int r = 0; r < 4; ++r)
          Cg[i * 16 + (long)(j * 16 + 4 * r) * ldc] = beta * cv[r] + alpha * acc[j][i][r];
      }
  }
}

int launch_gemm_nt(const double* A, long lda, const double* B, long ldb, double* C, long ldc,
                   long M, long Nc, long K, double alpha, double beta, long mask_off,
                   int kcap_mode, long kcap_off, hipStream_t s) {
  if (M <= 0 || Nc <= 0) return 0;
  if (M % TILE || Nc % TILE || K % KB) {
    set_error("gemm_nt: M, Nc must be multiples of 128 and K of 16");
    return -1;
  }
  long n_tr = M / TILE, n_tc = Nc / TILE;
  long total = n_tr * n_tc;
  int swz = total >= 64 ? 1 : 0;
  dim3 grid((unsigned)total), block(256);
  if (kcap_mode)
    hipLaunchKernelGGL(gemm_nt_kernel<true>, grid, block, 0, s, A, lda, B, ldb, C, ldc, K, alpha,
                       beta, mask_off, kcap_off, swz, n_tr, n_tc);
  else
    hipLaunchKernelGGL(gemm_nt_kernel<false>, grid, block, 0, s, A, lda, B, ldb, C, ldc, K, alpha,
                       beta, mask_off, kcap_off, swz, n_tr, n_tc);
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
  const int l15 = lane & 15, lq = lane >> 4;
  const long kbeg = ksplit * k_per_split;
  long kend = kbeg + k_per_split;
  if (kend > K) kend = K;

  // staging: chunk = 16 (k) x 128 (cols) per operand = 2048 doubles; thread t handles
  // column c = t & 127 and k-half h = t >> 7 (8 consecutive k) -> 4 double2 loads.
  const int sc = t & 127, sh = t >> 7;
  const double* Ag = A + (tr * TILE + sc) * lda;
  const double* Bg = B + (tc * TILE + sc) * ldb;
  double2 ra[4], rb[4];
  d4 acc[4][4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[j][i] = (d4){0.0, 0.0, 0.0, 0.0};

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
      SGP_COMPUTE(buf);
      SGP_TN_SSTORE(buf ^ 1);
      __syncthreads();
      buf ^= 1;
    }
    SGP_COMPUTE(buf);
  }
  if (Cpart) {
    // split-K: write the partial tile; a second kernel reduces in fixed order (deterministic)
    double* P = Cpart + ksplit * part_stride + (tr * TILE + wr * 64 + l15) +
                (tc * TILE + wc * 64 + lq) * ldc;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) P[i * 16 + (long)(j * 16 + 4 * r) * ldc] = acc[j][i][r];
    return;
  }
  double* Cg = C + (tr * TILE + wr * 64 + l15) + (tc * TILE + wc * 64 + lq) * ldc;
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        double* p = Cg + i * 16 + (long)(j * 16 + 4 * r) * ldc;
        double old = (beta == 0.0) ? 0.0 : beta * (*p);
        *p = old + alpha * acc[j][i][r];
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
