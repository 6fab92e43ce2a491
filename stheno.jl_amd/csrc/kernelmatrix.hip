// Covariance assembly kernels: fused pairwise distance + kernel + scale + sum-of-terms + noise,
// written once into the (bordered) factor matrix in HBM.
//
// Replaces, on the reference path, KernelFunctions.kernelmatrix [EXT] (Distances.pairwise +
// map(kappa, D)) as reached from src/gp/atomic_gp.jl:30-33, together with the block assembly
// of src/affine_transformations/cross.jl:59-86 (_collect(_mortar(...)) copies), the nested
// broadcasts of addition.jl:28-47 / product.jl:25-70, and the `C + Sigma_y` pass of
// AbstractGPs.FiniteGP [EXT].  One pass, written once: the kernel is HBM-write/fp64-VALU bound.
//
// Work decomposition: the global matrix is cut into 128x128 tiles aligned to the global tile
// grid (the same grid the Cholesky uses).  One 256-thread workgroup per tile; thread t owns row
// (t & 127) and the 64 columns of half (t >> 7), so each store instruction of a wave writes 64
// consecutive doubles of one column (512 B, coalesced; ColVecs/column-major output).
// The 128 column points of every term are staged in LDS point-major ([point][DMAX], a straight
// copy of the ColVecs segment) and read as wave-uniform broadcasts; the row point lives in
// registers.  Distances are the direct sum_d (a_d - b_d)^2 (exactly 0 on coincident points, so
// SE diagonals are exactly 1: test/gp/atomic_gp.jl:34), not the GEMM trick.
#include "common.h"
#include <algorithm>

namespace sgp {

enum { K_SE = 0, K_M12 = 1, K_M32 = 2, K_M52 = 3, K_WHITE = 4, K_CONST = 5 };

template <int KIND>
__device__ __forceinline__ double kern_eval_t(double d2, double param) {
  if (KIND == K_SE) return exp_nonpos(-0.5 * d2);
  if (KIND == K_M12) return exp_nonpos(-sqrt_nonneg(d2));
  if (KIND == K_M32) {
    double l = 1.7320508075688772 * sqrt_nonneg(d2);
    return (1.0 + l) * exp_nonpos(-l);
  }
  if (KIND == K_M52) {
    double l = 2.23606797749979 * sqrt_nonneg(d2);
    return fma(l, fma(l, 0.3333333333333333, 1.0), 1.0) * exp_nonpos(-l);   // 1 + l + l^2 / 3
  }
  if (KIND == K_WHITE) return d2 == 0.0 ? 1.0 : 0.0;
  return param;
}

__device__ __forceinline__ double kern_eval(int kind, double d2, double param) {
  switch (kind) {
    case K_SE: return kern_eval_t<K_SE>(d2, param);
    case K_M12: return kern_eval_t<K_M12>(d2, param);
    case K_M32: return kern_eval_t<K_M32>(d2, param);
    case K_M52: return kern_eval_t<K_M52>(d2, param);
    case K_WHITE: return kern_eval_t<K_WHITE>(d2, param);
    default: return param;
  }
}

constexpr int CCHUNK = 8;  // columns processed per accumulator chunk

template <int DMAX, int KIND>
__device__ __forceinline__ void term_chunk(double (&acc)[CCHUNK], const double (&xi)[DMAX],
                                           const double* sp, const double (&cw)[CCHUNK], double param) {
#pragma unroll
  for (int q = 0; q < CCHUNK; ++q) {
    double d2 = 0.0;
#pragma unroll
    for (int d = 0; d < DMAX; ++d) {
      double df = xi[d] - sp[q * DMAX + d];
      d2 = fma(df, df, d2);
    }
    acc[q] = fma(kern_eval_t<KIND>(d2, param), cw[q], acc[q]);
  }
}

template <int DMAX>
__global__ __launch_bounds__(256) void assemble_block_kernel(
    double* K, long ld, long r0, long nr, long c0, long nc, const DevTerm* terms, int nterms,
    int lower_only, int accumulate, int noise_kind, double sigma2, const double* noise_diag,
    long tile_r_first, long tile_c_first) {
  const long gtr = tile_r_first + blockIdx.x;
  const long gtc = tile_c_first + blockIdx.y;
  if (lower_only && gtr < gtc) return;
  extern __shared__ __attribute__((aligned(16))) double smem[];  // [nterms][128][DMAX]
  const int t = threadIdx.x;
  const int trow = t & 127, th = t >> 7;

  // column range of this tile inside the block-pair rectangle
  long cbeg = gtc * TILE, cend = cbeg + TILE;
  if (cbeg < c0) cbeg = c0;
  if (cend > c0 + nc) cend = c0 + nc;
  long rbeg = gtr * TILE, rend = rbeg + TILE;
  if (rbeg < r0) rbeg = r0;
  if (rend > r0 + nr) rend = r0 + nr;
  if (cbeg >= cend || rbeg >= rend) return;

  // stage column points of every term: smem[(tm*128 + p)*DMAX + d], p relative to gtc*128
  for (int tm = 0; tm < nterms; ++tm) {
    const DevTerm T = terms[tm];
    const int D = T.dim;
    for (int idx = t; idx < TILE * DMAX; idx += 256) {
      int p = idx / DMAX, d = idx % DMAX;
      long gc = gtc * TILE + p;
      double v = 0.0;
      if (d < D && gc >= cbeg && gc < cend) v = T.xc[(gc - c0) * T.ldc + d];
      smem[(tm * TILE + p) * DMAX + d] = v;
    }
  }
  __syncthreads();

  const long grow = gtr * TILE + trow;
  const bool row_ok = grow >= rbeg && grow < rend;
  if (!row_ok) return;  // no further barriers below
  const long lrow = grow - r0;
  const bool diag_noise = noise_kind >= 0;
  double nval = 0.0;
  if (diag_noise) nval = (noise_kind == 0) ? sigma2 : noise_diag[grow];

  for (int jc = 0; jc < 64; jc += CCHUNK) {
    const int pbase = th * 64 + jc;  // point index within the tile
    if (gtc * TILE + pbase >= cend) break;
    double acc[CCHUNK];
#pragma unroll
    for (int q = 0; q < CCHUNK; ++q) acc[q] = 0.0;
    for (int tm = 0; tm < nterms; ++tm) {
      const DevTerm T = terms[tm];
      double xi[DMAX];
      {
        const double* xr = T.xr + lrow * T.ldr;
#pragma unroll
        for (int d = 0; d < DMAX; ++d) xi[d] = (d < T.dim) ? xr[d] : 0.0;
      }
      const double rsv = T.coef * (T.rs ? T.rs[lrow] : 1.0);
      const double* sp = &smem[(tm * TILE + pbase) * DMAX];
      double cw[CCHUNK];
#pragma unroll
      for (int q = 0; q < CCHUNK; ++q) {
        long gc = gtc * TILE + pbase + q;
        cw[q] = rsv;
        if (T.cs) cw[q] = (gc >= cbeg && gc < cend) ? rsv * T.cs[gc - c0] : 0.0;
      }
      // the kind switch is hoisted out of the element loop: each case is straight-line code over
      // the CCHUNK independent entries, so their exp/sqrt chains interleave (ILP)
      switch (T.kind) {
        case K_SE: term_chunk<DMAX, K_SE>(acc, xi, sp, cw, T.param); break;
        case K_M12: term_chunk<DMAX, K_M12>(acc, xi, sp, cw, T.param); break;
        case K_M32: term_chunk<DMAX, K_M32>(acc, xi, sp, cw, T.param); break;
        case K_M52: term_chunk<DMAX, K_M52>(acc, xi, sp, cw, T.param); break;
        case K_WHITE: term_chunk<DMAX, K_WHITE>(acc, xi, sp, cw, T.param); break;
        default: term_chunk<DMAX, K_CONST>(acc, xi, sp, cw, T.param); break;
      }
    }
#pragma unroll
    for (int q = 0; q < CCHUNK; ++q) {
      long gc = gtc * TILE + pbase + q;
      if (gc >= cbeg && gc < cend) {
        double v = acc[q];
        if (diag_noise && gc == grow) v += nval;
        double* p = K + grow + gc * ld;
        if (accumulate) v += *p;
        *p = v;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------
// Any input dimension (the templated kernels stop at 64): ONE term per launch, the squared distances of a
// thread's row to its 64 columns accumulated in registers while the input dimension is walked in chunks of
// 16 (column-point chunk in LDS, row-point chunk in registers).  Same direct sum_d (a_d - b_d)^2 as the
// other kernels (exact zeros on coincident points), 2 fp64 VALU operations per entry and dimension: at D = 256
// the assembly of an N x N block costs what ~0.4 N^3 / 3 MFMA flops do at N = 16 384 -- no GEMM trick needed
// to stay below the factorisation.  (KernelFunctions' kernelmatrix [EXT] takes ColVecs of any dimension.)
// ---------------------------------------------------------------------------------------
constexpr int BIGD_CHUNK = 16;
__global__ __launch_bounds__(256) void assemble_bigd_kernel(double* K, long ld, long r0, long nr, long c0, long nc,
                                                            const DevTerm* terms, int lower_only, int accumulate,
                                                            int noise_kind, double sigma2, const double* noise_diag,
                                                            long tile_r_first, long tile_c_first) {
  const long gtr = tile_r_first + blockIdx.x;
  const long gtc = tile_c_first + blockIdx.y;
  if (lower_only && gtr < gtc) return;
  __shared__ __attribute__((aligned(16))) double sx[TILE * BIGD_CHUNK];
  const int t = threadIdx.x;
  const int trow = t & 127, th = t >> 7;
  long cbeg = gtc * TILE, cend = cbeg + TILE;
  if (cbeg < c0) cbeg = c0;
  if (cend > c0 + nc) cend = c0 + nc;
  long rbeg = gtr * TILE, rend = rbeg + TILE;
  if (rbeg < r0) rbeg = r0;
  if (rend > r0 + nr) rend = r0 + nr;
  if (cbeg >= cend || rbeg >= rend) return;   // uniform over the workgroup
  const DevTerm T = terms[0];
  const int D = T.dim;
  const long grow = gtr * TILE + trow;
  const bool row_ok = grow >= rbeg && grow < rend;
  const long lrow = grow - r0;
  double acc[64];
#pragma unroll
  for (int q = 0; q < 64; ++q) acc[q] = 0.0;
  for (int d0 = 0; d0 < D; d0 += BIGD_CHUNK) {
    __syncthreads();
    for (int idx = t; idx < TILE * BIGD_CHUNK; idx += 256) {
      const int p = idx / BIGD_CHUNK, d = idx % BIGD_CHUNK;
      const long gc = gtc * TILE + p;
      sx[idx] = (d0 + d < D && gc >= cbeg && gc < cend) ? T.xc[(gc - c0) * T.ldc + d0 + d] : 0.0;
    }
    __syncthreads();
    if (row_ok) {
      double xi[BIGD_CHUNK];
      const double* xr = T.xr + lrow * T.ldr + d0;
#pragma unroll
      for (int d = 0; d < BIGD_CHUNK; ++d) xi[d] = (d0 + d < D) ? xr[d] : 0.0;
#pragma unroll
      for (int q = 0; q < 64; ++q) {
        const double* sp = &sx[(th * 64 + q) * BIGD_CHUNK];
        double s = 0.0;
#pragma unroll
        for (int d = 0; d < BIGD_CHUNK; ++d) {
          const double df = xi[d] - sp[d];
          s = fma(df, df, s);
        }
        acc[q] += s;
      }
    }
  }
  if (!row_ok) return;
  const double rsv = T.coef * (T.rs ? T.rs[lrow] : 1.0);
  double nval = 0.0;
  if (noise_kind >= 0) nval = (noise_kind == 0) ? sigma2 : noise_diag[grow];
#pragma unroll
  for (int q = 0; q < 64; ++q) {
    const long gc = gtc * TILE + th * 64 + q;
    if (gc < cbeg || gc >= cend) continue;
    double v = kern_eval(T.kind, acc[q], T.param) * rsv * (T.cs ? T.cs[gc - c0] : 1.0);
    if (noise_kind >= 0 && gc == grow) v += nval;
    double* p = K + grow + gc * ld;
    if (accumulate) v += *p;
    *p = v;
  }
}

// ---------------------------------------------------------------------------------------
// Two-rows-per-thread variant (input dimension <= 16): thread (lane, wave) owns rows 2 lane, 2 lane + 1
// of the tile and the 32 columns of quarter `wave`, so
//   * every store is 16 bytes per lane: one wave instruction writes 128 consecutive rows of a column
//     (1 KiB contiguous) -- stores are issue-bound, and 8-byte stores needed twice as many;
//   * each column point fetched from LDS serves two rows (half the LDS traffic per entry).
// Everything a term needs inside the column loop sits in LDS -- column points, ROW points, coef * row
// scale, column scale -- staged once per tile: the first version re-read the row points and scales
// from global memory in every column chunk (they cannot be hoisted: the term loop is inside, and K may
// alias them), and the counters showed the waves parked in s_waitcnt half of the time
// (SQ_WAIT_ANY / SQ_WAVE_CYCLES = 0.51, VALU 53 % busy).  Interior tiles take a path without any
// bounds checks.
// tri != 0: 1-D grid over the live tiles of a square diagonal window only (tile id -> (tr, tc) of the
// lower triangle); the 2-D grid of the general case launches the dead upper tiles just to exit.
// ---------------------------------------------------------------------------------------
constexpr int CC2 = 4;  // columns per chunk (x 2 rows = 8 independent exp / sqrt chains)

// LDS doubles per term: column points, row points, row weights (coef * rs), column weights (cs)
__host__ __device__ constexpr int asm2_term_doubles(int dmax) { return 2 * TILE * dmax + 2 * TILE; }
int assemble_terms_per_launch(int dmax) {
  if (dmax > 16) return std::max(1, 64 / dmax);                    // one-row kernel: column points only
  return std::max(1, (int)((60 * 1024) / (sizeof(double) * asm2_term_doubles(dmax))));
}

template <int DMAX, int KIND>
__device__ __forceinline__ void term_chunk2(double (&acc0)[CC2], double (&acc1)[CC2], const double (&xi0)[DMAX],
                                            const double (&xi1)[DMAX], const double* sp, const double* scq, double rs0,
                                            double rs1, double param) {
#pragma unroll
  for (int q = 0; q < CC2; ++q) {
    double d2a = 0.0, d2b = 0.0;
#pragma unroll
    for (int d = 0; d < DMAX; ++d) {
      const double c = sp[q * DMAX + d];
      const double da = xi0[d] - c, db = xi1[d] - c;
      d2a = fma(da, da, d2a);
      d2b = fma(db, db, d2b);
    }
    const double cq = scq[q];
    acc0[q] = fma(kern_eval_t<KIND>(d2a, param), rs0 * cq, acc0[q]);
    acc1[q] = fma(kern_eval_t<KIND>(d2b, param), rs1 * cq, acc1[q]);
  }
}

template <int DMAX>
__global__ __launch_bounds__(256) void assemble_block2_kernel(
    double* __restrict__ K, long ld, long r0, long nr, long c0, long nc, const DevTerm* __restrict__ terms,
    int nterms, int lower_only, int accumulate, int noise_kind, double sigma2,
    const double* __restrict__ noise_diag, long tile_r_first, long tile_c_first, int tri) {
  long gtr, gtc;
  if (tri) {
    const long id = blockIdx.x;
    long tr = (long)((sqrt(8.0 * (double)id + 1.0) - 1.0) * 0.5);
    while (tr * (tr + 1) / 2 > id) --tr;
    while ((tr + 1) * (tr + 2) / 2 <= id) ++tr;
    gtr = tile_r_first + tr;
    gtc = tile_c_first + (id - tr * (tr + 1) / 2);
  } else {
    gtr = tile_r_first + blockIdx.x;
    gtc = tile_c_first + blockIdx.y;
    if (lower_only && gtr < gtc) return;
  }
  extern __shared__ __attribute__((aligned(16))) double smem[];
  constexpr int TD = asm2_term_doubles(DMAX);  // per term: [128][DMAX] cols | [128][DMAX] rows | rw[128] | cw[128]
  const int t = threadIdx.x;
  const int lane = t & 63, wv = t >> 6;

  long cbeg = gtc * TILE, cend = cbeg + TILE;
  if (cbeg < c0) cbeg = c0;
  if (cend > c0 + nc) cend = c0 + nc;
  long rbeg = gtr * TILE, rend = rbeg + TILE;
  if (rbeg < r0) rbeg = r0;
  if (rend > r0 + nr) rend = r0 + nr;
  if (cbeg >= cend || rbeg >= rend) return;

  for (int tm = 0; tm < nterms; ++tm) {
    const DevTerm T = terms[tm];
    const int D = T.dim;
    double* sc = smem + tm * TD;
    double* sr = sc + TILE * DMAX;
    for (int idx = t; idx < TILE * DMAX; idx += 256) {
      const int p = idx / DMAX, d = idx % DMAX;
      const long gc = gtc * TILE + p, gr = gtr * TILE + p;
      sc[idx] = (d < D && gc >= cbeg && gc < cend) ? T.xc[(gc - c0) * T.ldc + d] : 0.0;
      sr[d * TILE + p] = (d < D && gr >= rbeg && gr < rend) ? T.xr[(gr - r0) * T.ldr + d] : 0.0;  // [d][row]
    }
    if (t < TILE) {
      const long gr = gtr * TILE + t;
      sr[TILE * DMAX + t] = (gr >= rbeg && gr < rend) ? T.coef * (T.rs ? T.rs[gr - r0] : 1.0) : 0.0;
    } else {
      const int p = t - TILE;
      const long gc = gtc * TILE + p;
      sr[TILE * DMAX + TILE + p] = (gc >= cbeg && gc < cend) ? (T.cs ? T.cs[gc - c0] : 1.0) : 0.0;
    }
  }
  __syncthreads();

  const long g0 = gtr * TILE + 2 * lane, g1 = g0 + 1;
  const bool ok0 = g0 >= rbeg && g0 < rend, ok1 = g1 >= rbeg && g1 < rend;
  const bool diag_noise = noise_kind >= 0 && gtr == gtc;  // only tiles on the diagonal carry Sigma_y entries
  double nv0 = 0.0, nv1 = 0.0;
  if (diag_noise) {
    nv0 = (noise_kind == 0) ? sigma2 : (ok0 ? noise_diag[g0] : 0.0);
    nv1 = (noise_kind == 0) ? sigma2 : (ok1 ? noise_diag[g1] : 0.0);
  }
  // interior tile: all 128 x 128 entries belong to the block pair, and 16-byte stores are aligned
  const bool full = (cend - cbeg == TILE) && (rend - rbeg == TILE) && !accumulate &&
                    ((((unsigned long long)(K + g0)) & 15ull) == 0) && ((ld & 1) == 0);

  for (int jc = 0; jc < 32; jc += CC2) {
    const int pbase = wv * 32 + jc;  // point index within the tile
    double acc0[CC2], acc1[CC2];
#pragma unroll
    for (int q = 0; q < CC2; ++q) acc0[q] = acc1[q] = 0.0;
    for (int tm = 0; tm < nterms; ++tm) {
      const DevTerm T = terms[tm];
      const double* sc = smem + tm * TD;
      const double* sr = sc + TILE * DMAX;
      double xi0[DMAX], xi1[DMAX];
#pragma unroll
      for (int d = 0; d < DMAX; ++d) {  // row points are stored [d][row]: one conflict-free 16-byte read per d
        const double2 x2 = *reinterpret_cast<const double2*>(sr + d * TILE + 2 * lane);
        xi0[d] = x2.x;
        xi1[d] = x2.y;
      }
      const double rs0 = sr[TILE * DMAX + 2 * lane], rs1 = sr[TILE * DMAX + 2 * lane + 1];
      const double* sp = sc + pbase * DMAX;
      const double* scq = sr + TILE * DMAX + TILE + pbase;
      switch (T.kind) {
        case K_SE: term_chunk2<DMAX, K_SE>(acc0, acc1, xi0, xi1, sp, scq, rs0, rs1, T.param); break;
        case K_M12: term_chunk2<DMAX, K_M12>(acc0, acc1, xi0, xi1, sp, scq, rs0, rs1, T.param); break;
        case K_M32: term_chunk2<DMAX, K_M32>(acc0, acc1, xi0, xi1, sp, scq, rs0, rs1, T.param); break;
        case K_M52: term_chunk2<DMAX, K_M52>(acc0, acc1, xi0, xi1, sp, scq, rs0, rs1, T.param); break;
        case K_WHITE: term_chunk2<DMAX, K_WHITE>(acc0, acc1, xi0, xi1, sp, scq, rs0, rs1, T.param); break;
        default: term_chunk2<DMAX, K_CONST>(acc0, acc1, xi0, xi1, sp, scq, rs0, rs1, T.param); break;
      }
    }
    if (full && !diag_noise) {
      double* kp = K + g0 + (gtc * TILE + pbase) * ld;
#pragma unroll
      for (int q = 0; q < CC2; ++q) *reinterpret_cast<double2*>(kp + q * ld) = make_double2(acc0[q], acc1[q]);
    } else if (full) {
#pragma unroll
      for (int q = 0; q < CC2; ++q) {
        const long gc = gtc * TILE + pbase + q;
        double v0 = acc0[q], v1 = acc1[q];
        if (gc == g0) v0 += nv0;
        if (gc == g1) v1 += nv1;
        *reinterpret_cast<double2*>(K + g0 + gc * ld) = make_double2(v0, v1);
      }
    } else {
#pragma unroll
      for (int q = 0; q < CC2; ++q) {
        const long gc = gtc * TILE + pbase + q;
        if (gc >= cbeg && gc < cend) {
          double v0 = acc0[q], v1 = acc1[q];
          if (diag_noise && gc == g0) v0 += nv0;
          if (diag_noise && gc == g1) v1 += nv1;
          double* p = K + g0 + gc * ld;
          if (ok0) p[0] = accumulate ? p[0] + v0 : v0;
          if (ok1) p[1] = accumulate ? p[1] + v1 : v1;
        }
      }
    }
  }
}

template <int DMAX>
static int launch_assemble_t(double* K, long ld, long r0, long nr, long c0, long nc,
                             const DevTerm* d_terms, int nterms, int lower_only, int accumulate,
                             int noise_kind, double sigma2, const double* d_noise_diag,
                             long tile_r_first, long tile_c_first, long tile_r_cnt,
                             long tile_c_cnt, hipStream_t s) {
  size_t lds = (size_t)nterms * TILE * DMAX * sizeof(double);
  if (DMAX <= 16) lds = (size_t)nterms * asm2_term_doubles(DMAX <= 16 ? DMAX : 16) * sizeof(double);
  if (lds == 0) lds = 16;
  dim3 grid((unsigned)tile_r_cnt, (unsigned)tile_c_cnt), block(256);
  if (DMAX <= 16) {
    // square window on the diagonal in lower mode: launch the live tiles only
    const int tri = lower_only && tile_r_first == tile_c_first && tile_r_cnt == tile_c_cnt;
    if (tri) grid = dim3((unsigned)(tile_r_cnt * (tile_r_cnt + 1) / 2));
    hipLaunchKernelGGL(assemble_block2_kernel<(DMAX <= 16 ? DMAX : 16)>, grid, block, lds, s, K, ld, r0, nr, c0, nc,
                       d_terms, nterms, lower_only, accumulate, noise_kind, sigma2, d_noise_diag, tile_r_first,
                       tile_c_first, tri);
  } else {
    hipLaunchKernelGGL(assemble_block_kernel<DMAX>, grid, block, lds, s, K, ld, r0, nr, c0, nc,
                       d_terms, nterms, lower_only, accumulate, noise_kind, sigma2, d_noise_diag,
                       tile_r_first, tile_c_first);
  }
  SGP_HIP(hipGetLastError());
  return 0;
}

int launch_assemble_block(double* K, long ld, long r0, long nr, long c0, long nc,
                          const DevTerm* d_terms, int nterms, int dmax, int lower_only,
                          int accumulate, int noise_kind, double sigma2,
                          const double* d_noise_diag, long tile_r_first, long tile_c_first,
                          long tile_r_cnt, long tile_c_cnt, hipStream_t s) {
  if (tile_r_cnt <= 0 || tile_c_cnt <= 0) return 0;
#define SGP_ASM(DM)                                                                            \
  return launch_assemble_t<DM>(K, ld, r0, nr, c0, nc, d_terms, nterms, lower_only, accumulate, \
                               noise_kind, sigma2, d_noise_diag, tile_r_first, tile_c_first,   \
                               tile_r_cnt, tile_c_cnt, s)
  if (dmax <= 1) SGP_ASM(1);
  if (dmax <= 2) SGP_ASM(2);
  if (dmax <= 4) SGP_ASM(4);
  if (dmax <= 8) SGP_ASM(8);
  if (dmax <= 16) SGP_ASM(16);
  if (dmax <= 32) SGP_ASM(32);
  if (dmax <= 64) SGP_ASM(64);
#undef SGP_ASM
  if (nterms != 1) {
    set_error("assemble: input dimension > 64 takes one term per launch");
    return -1;
  }
  hipLaunchKernelGGL(assemble_bigd_kernel, dim3((unsigned)tile_r_cnt, (unsigned)tile_c_cnt), dim3(256), 0, s, K, ld, r0,
                     nr, c0, nc, d_terms, lower_only, accumulate, noise_kind, sigma2, d_noise_diag, tile_r_first,
                     tile_c_first);
  SGP_HIP(hipGetLastError());
  return 0;
}

// identity padding: rows/cols in [N, n_pad) of the square part get delta(r, c); everything in a
// padded column below the square part (border rows) gets 0.
__global__ void fill_pad_rows_kernel(double* K, long ld, long N, long n_pad, long nc, long gc0) {
  // rows [N, n_pad) x local cols [0, nc)
  long h = n_pad - N;
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= h * nc) return;
  long r = N + idx % h, lc = idx / h;
  long gc = gc0 + lc;
  K[r + lc * ld] = (r == gc) ? 1.0 : 0.0;
}
__global__ void fill_pad_cols_kernel(double* K, long ld, long N, long n_pad, long m_tot, long nc,
                                     long gc0, long row_lo) {
  // local cols whose global index >= N, rows [row_lo, m_tot)
  long h = m_tot - row_lo;
  long first = N > gc0 ? N - gc0 : 0;  // first padded local col
  long w = nc - first;
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (w <= 0 || idx >= h * w) return;
  long r = row_lo + idx % h, lc = first + idx / h;
  long gc = gc0 + lc;
  K[r + lc * ld] = (r == gc) ? 1.0 : 0.0;
}

int launch_fill_pad(double* K, long ld, long N, long n_pad, long c0, long nc, long m_tot,
                    long row_lo, hipStream_t s) {
  // K points at local column 0 (global column c0), row index global.
  if (n_pad > N) {
    long tot = (n_pad - N) * nc;
    hipLaunchKernelGGL(fill_pad_rows_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s,
                       K, ld, N, n_pad, nc, c0);
    SGP_HIP(hipGetLastError());
    long first = N > c0 ? N - c0 : 0;
    long w = nc - first;
    if (w > 0) {
      long tot2 = (m_tot - row_lo) * w;
      hipLaunchKernelGGL(fill_pad_cols_kernel, dim3((unsigned)((tot2 + 255) / 256)), dim3(256), 0,
                         s, K, ld, N, n_pad, m_tot, nc, c0, row_lo);
      SGP_HIP(hipGetLastError());
    }
  }
  return 0;
}

// bordered rows: A[n_pad + s, lc] = Y[gc, s] - mean[gc] for gc < N, s < ncols; 0 otherwise
__global__ void border_rows_kernel(double* A, long ld, long n_pad, long N, long gc0, long nc,
                                   const double* Y, long ldy, long ncols, const double* mean,
                                   long r_pad) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= r_pad * nc) return;
  long s = idx % r_pad, lc = idx / r_pad;
  long gc = gc0 + lc;
  double v = 0.0;
  if (s < ncols && gc < N) v = Y[gc + s * ldy] - (mean ? mean[gc] : 0.0);
  A[n_pad + s + lc * ld] = v;
}

int launch_border_rows(double* A, long ld, long n_pad, long N, long c0, long nc, const double* dY,
                       long ldy, long ncols, const double* d_mean, hipStream_t s) {
  if (ncols <= 0) return 0;
  long r_pad = (ncols + TILE - 1) / TILE * TILE;
  long tot = r_pad * nc;
  hipLaunchKernelGGL(border_rows_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, A,
                     ld, n_pad, N, c0, nc, dY, ldy, ncols, d_mean, r_pad);
  SGP_HIP(hipGetLastError());
  return 0;
}

// diag of a block: out[i] = sum_t coef rs[i] cs[i] k(xr_i, xc_i)   (kernelmatrix_diag [EXT],
// src/gp/util.jl:5-7, cross.jl:64-77, addition.jl:31-40, product.jl:32-47)
__global__ void diag_terms_kernel(double* out, long n, const DevTerm* terms, int nterms) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double acc = 0.0;
  for (int tm = 0; tm < nterms; ++tm) {
    const DevTerm T = terms[tm];
    double d2 = 0.0;
    for (int d = 0; d < T.dim; ++d) {
      double df = T.xr[i * T.ldr + d] - T.xc[i * T.ldc + d];
      d2 = fma(df, df, d2);
    }
    // same operation order as assemble_block_kernel, so var(f, x) == diag(cov(f, x)) bit for bit
    double cw = T.coef * (T.rs ? T.rs[i] : 1.0);
    if (T.cs) cw = cw * T.cs[i];
    acc = fma(kern_eval(T.kind, d2, T.param), cw, acc);
  }
  out[i] = acc;
}

int launch_diag_terms(double* out, long n, const DevTerm* d_terms, int nterms, hipStream_t s) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(diag_terms_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, out, n,
                     d_terms, nterms);
  SGP_HIP(hipGetLastError());
  return 0;
}

// dense Sigma_y: K[r, c] += S[r, c] on the tiles the factorisation reads
__global__ void add_dense_kernel(double* K, long ld, const double* S, long lds, long N,
                                 int lower_only) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= N * N) return;
  long r = idx % N, c = idx / N;
  if (lower_only && (r / TILE) < (c / TILE)) return;
  K[r + c * ld] += S[r + c * lds];
}

int launch_add_dense(double* K, long ld, const double* S, long lds, long N, int lower_only,
                     hipStream_t s) {
  long tot = N * N;
  if (tot <= 0) return 0;
  hipLaunchKernelGGL(add_dense_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, K, ld,
                     S, lds, N, lower_only);
  SGP_HIP(hipGetLastError());
  return 0;
}

// dense Sigma_y onto ONE packed column panel of a sharded factorisation (multi.hip): columns c0 .. c0 + w, rows c0 .. N;
// P[(r - c0) + lc * ldp] += S[(r - c0) + lc * lds] on the tiles the factorisation reads (tile row >= tile column)
__global__ void add_dense_cols_kernel(double* P, long ldp, const double* S, long lds, long c0, long w, long N) {
  const long rows = N - c0;
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * w) return;
  const long rr = idx % rows, lc = idx / rows;
  if (c0 + lc >= N) return;
  if ((rr / TILE) < (lc / TILE)) return;   // (c0 is a multiple of 128: tile indices relative to the panel)
  P[rr + lc * ldp] += S[rr + lc * lds];
}
int launch_add_dense_cols(double* P, long ldp, const double* S, long lds, long c0, long w, long N, hipStream_t s) {
  const long tot = (N - c0) * w;
  if (tot <= 0) return 0;
  hipLaunchKernelGGL(add_dense_cols_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, P, ldp, S, lds, c0, w, N);
  SGP_HIP(hipGetLastError());
  return 0;
}

// K[r, c] = K[c, r] for r < c: makes a symmetric covariance bit-exactly symmetric (multi-term
// blocks sum their cross terms in a different order above and below the diagonal).
__global__ void mirror_lower_kernel(double* K, long ld, long N) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= N * N) return;
  long r = idx % N, c = idx / N;
  if (r < c) K[r + c * ld] = K[c + r * ld];
}

int launch_mirror_lower(double* K, long ld, long N, hipStream_t s) {
  long tot = N * N;
  if (tot <= 0) return 0;
  hipLaunchKernelGGL(mirror_lower_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, K, ld, N);
  SGP_HIP(hipGetLastError());
  return 0;
}

}  // namespace sgp
