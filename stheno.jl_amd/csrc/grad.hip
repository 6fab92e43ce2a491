// Reverse-mode gradient of logpdf: the contraction
//   d logpdf / d theta = sum_ij G_ij dC_ij / d theta,   G = (alpha alpha' - C^-1) / 2
// against the flattened covariance terms.  On the reference path this is what Zygote derives
// through cholesky / kernelmatrix for hyper-parameter learning
// (examples/getting_started/script.jl:154-213; AD glue sites SURVEY.md section 2 #11).
//
// For every term t of a block pair the kernel accumulates
//   gc[t] = sum_ij G_ij rs_i k_t(x_i, x_j) cs_j                     (d / d coef_t)
//   gs[t] = sum_ij G_ij coef_t rs_i cs_j  d k_t(g x_i, g x_j)/dg |_{g=1}   (d / d input scale)
// with the same tiling as the assembly kernel (kernelmatrix.hip): 128x128 tiles, thread = one row
// x 64 columns, column points broadcast from LDS.  Partials per workgroup are reduced in fixed
// order by a second kernel (deterministic).
#include "common.h"
#include <algorithm>

namespace sgp {

enum { G_SE = 0, G_M12 = 1, G_M32 = 2, G_M52 = 3, G_WHITE = 4, G_CONST = 5 };
constexpr int GRAD_MAXT = 8;  // terms per launch

// k and d k / d g (input scale, at g = 1) from the squared distance
__device__ __forceinline__ void kern_and_dscale(int kind, double d2, double param, double& k, double& dk) {
  switch (kind) {
    case G_SE:
      k = exp(-0.5 * d2);
      dk = -d2 * k;
      return;
    case G_M12: {
      double d = sqrt(d2);
      k = exp(-d);
      dk = -d * k;
      return;
    }
    case G_M32: {
      double l = 1.7320508075688772 * sqrt(d2);
      double e = exp(-l);
      k = (1.0 + l) * e;
      dk = -3.0 * d2 * e;
      return;
    }
    case G_M52: {
      double l = 2.23606797749979 * sqrt(d2);
      double e = exp(-l);
      k = (1.0 + l + l * l / 3.0) * e;
      dk = -(5.0 * d2 / 3.0) * (1.0 + l) * e;
      return;
    }
    case G_WHITE:
      k = d2 == 0.0 ? 1.0 : 0.0;
      dk = 0.0;
      return;
    default:
      k = param;
      dk = 0.0;
  }
}

template <int DMAX>
__global__ __launch_bounds__(256) void grad_block_kernel(const double* Kinv, long ldk, const double* alpha,
                                                         long r0, long nr, long c0, long nc,
                                                         const DevTerm* terms, int nterms,
                                                         long tile_r_first, long tile_c_first,
                                                         double* partials /*[blocks][GRAD_MAXT][2]*/,
                                                         long clo, long chi) {   // column window inside the block (absolute)
  // terms per launch: nterms * DMAX <= 64 (the launcher groups them so), at most GRAD_MAXT
  constexpr int TMAX = (64 / DMAX < GRAD_MAXT) ? 64 / DMAX : GRAD_MAXT;
  const long gtr = tile_r_first + blockIdx.x;
  const long gtc = tile_c_first + blockIdx.y;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* sx = smem;                           // [nterms][128][DMAX] column points
  double* scs = smem + TMAX * TILE * DMAX;     // [nterms][128] column scales
  const int t = threadIdx.x;
  const int trow = t & 127, th = t >> 7;
  long cbeg = gtc * TILE, cend = cbeg + TILE;
  if (cbeg < c0) cbeg = c0;
  if (cend > c0 + nc) cend = c0 + nc;
  if (cbeg < clo) cbeg = clo;   // (a rank of the sharded gradient contracts the columns of ITS panels only: multi.hip)
  if (cend > chi) cend = chi;
  long rbeg = gtr * TILE, rend = rbeg + TILE;
  if (rbeg < r0) rbeg = r0;
  if (rend > r0 + nr) rend = r0 + nr;
  const bool live_tile = cbeg < cend && rbeg < rend;

  for (int tm = 0; tm < nterms; ++tm) {
    const DevTerm T = terms[tm];
    for (int idx = t; idx < TILE * DMAX; idx += 256) {
      int p = idx / DMAX, d = idx % DMAX;
      long gc = gtc * TILE + p;
      double v = 0.0;
      if (live_tile && d < T.dim && gc >= cbeg && gc < cend) v = T.xc[(gc - c0) * T.ldc + d];
      sx[(tm * TILE + p) * DMAX + d] = v;
    }
    if (t < TILE) {
      long gc = gtc * TILE + t;
      scs[tm * TILE + t] = (live_tile && T.cs && gc >= cbeg && gc < cend) ? T.cs[gc - c0] : 1.0;
    }
  }
  __syncthreads();

  double gc_acc[GRAD_MAXT], gs_acc[GRAD_MAXT];
#pragma unroll
  for (int q = 0; q < GRAD_MAXT; ++q) gc_acc[q] = gs_acc[q] = 0.0;

  const long grow = gtr * TILE + trow;
  if (live_tile && grow >= rbeg && grow < rend) {
    const long lrow = grow - r0;
    const double ai = alpha ? alpha[grow] : 0.0;
    // per-term row data in registers: the row point, its scale, the term constants
    double xr[TMAX * DMAX], rsv[TMAX], coef[TMAX], param[TMAX];
    int kind[TMAX];
#pragma unroll
    for (int tm = 0; tm < TMAX; ++tm) {
      kind[tm] = G_CONST;
      rsv[tm] = coef[tm] = param[tm] = 0.0;
#pragma unroll
      for (int d = 0; d < DMAX; ++d) xr[tm * DMAX + d] = 0.0;
      if (tm < nterms) {
        const DevTerm T = terms[tm];
        kind[tm] = T.kind;
        coef[tm] = T.coef;
        param[tm] = T.param;
        rsv[tm] = T.rs ? T.rs[lrow] : 1.0;
        const double* xp = T.xr + lrow * T.ldr;
#pragma unroll
        for (int d = 0; d < DMAX; ++d) xr[tm * DMAX + d] = (d < T.dim) ? xp[d] : 0.0;
      }
    }
    const long pbeg = th * 64, gcol0 = gtc * TILE;
    for (int p = (int)pbeg; p < (int)pbeg + 64; ++p) {
      const long gc = gcol0 + p;
      if (gc < cbeg || gc >= cend) continue;
      // alpha == nullptr: `Kinv` holds the cotangent matrix G itself (ELBO gradient)
      const double g = alpha ? 0.5 * (ai * alpha[gc] - Kinv[grow + gc * ldk]) : Kinv[grow + gc * ldk];
#pragma unroll
      for (int tm = 0; tm < TMAX; ++tm) {
        if (tm >= nterms) break;
        const double* sp = &sx[(tm * TILE + p) * DMAX];
        double d2 = 0.0;
#pragma unroll
        for (int d = 0; d < DMAX; ++d) {
          double df = xr[tm * DMAX + d] - sp[d];
          d2 = fma(df, df, d2);
        }
        double k, dk;
        kern_and_dscale(kind[tm], d2, param[tm], k, dk);
        double w = g * rsv[tm] * scs[tm * TILE + p];
        gc_acc[tm] = fma(w, k, gc_acc[tm]);
        gs_acc[tm] = fma(w * coef[tm], dk, gs_acc[tm]);
      }
    }
  }
  // block reduction (fixed order): wave shuffle, then 4 partials through LDS
  __syncthreads();
  double* red = smem;  // reuse: [4 waves][GRAD_MAXT][2]
#pragma unroll
  for (int tm = 0; tm < GRAD_MAXT; ++tm) {
    double a = gc_acc[tm], b = gs_acc[tm];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      a += __shfl_xor(a, off, 64);
      b += __shfl_xor(b, off, 64);
    }
    if ((t & 63) == 0) {
      red[((t >> 6) * GRAD_MAXT + tm) * 2 + 0] = a;
      red[((t >> 6) * GRAD_MAXT + tm) * 2 + 1] = b;
    }
  }
  __syncthreads();
  if (t < GRAD_MAXT * 2) {
    double s = 0.0;
    for (int wv = 0; wv < 4; ++wv) s += red[wv * GRAD_MAXT * 2 + t];
    const long blk = (long)blockIdx.y * gridDim.x + blockIdx.x;
    partials[blk * GRAD_MAXT * 2 + t] = s;
  }
}

// out[t*2 + c] = sum_b partials[b][t][c]: one workgroup per output, 256 strided partial sums
// combined by a fixed tree (deterministic)
__global__ __launch_bounds__(256) void grad_reduce_kernel(const double* partials, long nblocks, int nterms,
                                                          double* out_coef, double* out_scale, int accumulate) {
  __shared__ double sh[256];
  const int idx = blockIdx.x;  // term * 2 + component
  double s = 0.0;
  for (long b = threadIdx.x; b < nblocks; b += 256) s += partials[b * GRAD_MAXT * 2 + idx];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int off = 128; off >= 1; off >>= 1) {
    if ((int)threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    double* dst = (idx & 1) ? out_scale + (idx >> 1) : out_coef + (idx >> 1);
    *dst = accumulate ? *dst + sh[0] : sh[0];
  }
}

// Any input dimension (> 64): ONE term per launch, the dimension walked in chunks of 16 -- the thread's 64 squared
// distances accumulate in registers (d ascending, fma: the order of the small-dimension kernel), the column chunk sits in LDS,
// the row chunk comes from global memory; then the same kernel / derivative evaluation and the same reductions.
constexpr int GTB_CHUNK = 16;
__global__ __launch_bounds__(256) void grad_block_bigd_kernel(const double* Kinv, long ldk, const double* alpha, long r0,
                                                              long nr, long c0, long nc, const DevTerm* terms,
                                                              long tile_r_first, long tile_c_first, double* partials,
                                                              long clo, long chi) {
  __shared__ __attribute__((aligned(16))) double sx[TILE * GTB_CHUNK];
  __shared__ double scs[TILE];
  __shared__ double red[4 * GRAD_MAXT * 2];
  const DevTerm T = terms[0];
  const long gtr = tile_r_first + blockIdx.x;
  const long gtc = tile_c_first + blockIdx.y;
  const int t = threadIdx.x;
  const int trow = t & 127, th = t >> 7;
  long cbeg = gtc * TILE, cend = cbeg + TILE;
  if (cbeg < c0) cbeg = c0;
  if (cend > c0 + nc) cend = c0 + nc;
  if (cbeg < clo) cbeg = clo;
  if (cend > chi) cend = chi;
  long rbeg = gtr * TILE, rend = rbeg + TILE;
  if (rbeg < r0) rbeg = r0;
  if (rend > r0 + nr) rend = r0 + nr;
  const bool live_tile = cbeg < cend && rbeg < rend;
  const long grow = gtr * TILE + trow;
  const bool live = live_tile && grow >= rbeg && grow < rend;
  const long lrow = grow - r0;
  const int D = (int)T.dim;
  if (t < TILE) {
    const long gc = gtc * TILE + t;
    scs[t] = (live_tile && T.cs && gc >= cbeg && gc < cend) ? T.cs[gc - c0] : 1.0;
  }
  double d2[64];
#pragma unroll
  for (int q = 0; q < 64; ++q) d2[q] = 0.0;
  for (int d0 = 0; d0 < D; d0 += GTB_CHUNK) {
    __syncthreads();
    for (int idx = t; idx < TILE * GTB_CHUNK; idx += 256) {
      const int p = idx / GTB_CHUNK, d = idx % GTB_CHUNK;
      const long gc = gtc * TILE + p;
      double v = 0.0;
      if (live_tile && d0 + d < D && gc >= cbeg && gc < cend) v = T.xc[(gc - c0) * T.ldc + d0 + d];
      sx[p * GTB_CHUNK + d] = v;
    }
    __syncthreads();
    double xr[GTB_CHUNK];
#pragma unroll
    for (int d = 0; d < GTB_CHUNK; ++d) xr[d] = (live && d0 + d < D) ? T.xr[lrow * T.ldr + d0 + d] : 0.0;
#pragma unroll
    for (int q = 0; q < 64; ++q) {
      const double* sp = &sx[(th * 64 + q) * GTB_CHUNK];
#pragma unroll
      for (int d = 0; d < GTB_CHUNK; ++d) {
        const double df = xr[d] - sp[d];
        d2[q] = fma(df, df, d2[q]);
      }
    }
  }
  double a = 0.0, b = 0.0;
  if (live) {
    const double ai = alpha ? alpha[grow] : 0.0;
    const double rsv = T.rs ? T.rs[lrow] : 1.0;
#pragma unroll
    for (int q = 0; q < 64; ++q) {
      const int p = th * 64 + q;
      const long gc = gtc * TILE + p;
      if (gc < cbeg || gc >= cend) continue;
      const double g = alpha ? 0.5 * (ai * alpha[gc] - Kinv[grow + gc * ldk]) : Kinv[grow + gc * ldk];
      double k, dk;
      kern_and_dscale(T.kind, d2[q], T.param, k, dk);
      const double w = g * rsv * scs[p];
      a = fma(w, k, a);
      b = fma(w * T.coef, dk, b);
    }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    a += __shfl_xor(a, off, 64);
    b += __shfl_xor(b, off, 64);
  }
  if ((t & 63) == 0) {
    red[(t >> 6) * 2 + 0] = a;
    red[(t >> 6) * 2 + 1] = b;
  }
  __syncthreads();
  if (t < GRAD_MAXT * 2) {
    double s = 0.0;
    if (t < 2)
      for (int wv = 0; wv < 4; ++wv) s += red[wv * 2 + t];
    const long blk = (long)blockIdx.y * gridDim.x + blockIdx.x;
    partials[blk * GRAD_MAXT * 2 + t] = s;   // terms 1 .. GRAD_MAXT - 1: zeros (the reduction reads nterms = 1 only)
  }
}

template <int DMAX>
static int launch_grad_t(const double* Kinv, long ldk, const double* alpha, long r0, long nr, long c0,
                         long nc, const DevTerm* d_terms, int nterms, long trf, long tcf, long trc, long tcc,
                         double* partials, hipStream_t s, long clo, long chi) {
  constexpr int TMAX = (64 / DMAX < GRAD_MAXT) ? 64 / DMAX : GRAD_MAXT;
  if (nterms > TMAX) {
    set_error("grad: too many terms in one launch for this input dimension");
    return -1;
  }
  size_t lds = (size_t)(TMAX * TILE * DMAX + TMAX * TILE) * sizeof(double);
  dim3 grid((unsigned)trc, (unsigned)tcc), block(256);
  SGP_LDS_ATTR_ONCE(grad_block_kernel<DMAX>, lds);
  hipLaunchKernelGGL(grad_block_kernel<DMAX>, grid, block, lds, s, Kinv, ldk, alpha, r0, nr, c0, nc, d_terms,
                     nterms, trf, tcf, partials, clo, chi);
  SGP_HIP(hipGetLastError());
  return 0;
}

int launch_grad_block(const double* Kinv, long ldk, const double* alpha, long r0, long nr, long c0, long nc,
                      const DevTerm* d_terms, int nterms, int dmax, long trf, long tcf, long trc, long tcc,
                      double* partials, double* out_coef, double* out_scale, hipStream_t s, int accumulate, long clo,
                      long chi) {
  if (nterms <= 0 || trc <= 0 || tcc <= 0) return 0;
  if (nterms > GRAD_MAXT) {
    set_error("grad: too many terms in one launch");
    return -1;
  }
  int rc = -1;
#define SGP_GR(DM) rc = launch_grad_t<DM>(Kinv, ldk, alpha, r0, nr, c0, nc, d_terms, nterms, trf, tcf, trc, tcc, partials, s, clo, chi)
  if (dmax <= 1) SGP_GR(1);
  else if (dmax <= 2) SGP_GR(2);
  else if (dmax <= 4) SGP_GR(4);
  else if (dmax <= 8) SGP_GR(8);
  else if (dmax <= 16) SGP_GR(16);
  else if (dmax <= 32) SGP_GR(32);
  else if (dmax <= 64) SGP_GR(64);
  else {
    if (nterms != 1) {
      set_error("grad: one term per launch beyond input dimension 64");
      return -1;
    }
    hipLaunchKernelGGL(grad_block_bigd_kernel, dim3((unsigned)trc, (unsigned)tcc), dim3(256), 0, s, Kinv, ldk, alpha, r0, nr,
                       c0, nc, d_terms, trf, tcf, partials, clo, chi);
    SGP_HIP(hipGetLastError());
    rc = 0;
  }
#undef SGP_GR
  if (rc) return rc;
  hipLaunchKernelGGL(grad_reduce_kernel, dim3((unsigned)(nterms * 2)), dim3(256), 0, s, partials, trc * tcc, nterms, out_coef,
                     out_scale, accumulate);
  SGP_HIP(hipGetLastError());
  return 0;
}

// bordered rows of the gradient factorisation: row n_pad = (y - m)' (+ 127 zero rows),
// rows [n_pad + 128, 2 n_pad + 128) = identity (for i < N)
__global__ void grad_border_kernel(double* A, long ld, long n_pad, long N, const double* y,
                                   const double* mean, long nrows) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= nrows * n_pad) return;
  long r = idx % nrows, c = idx / nrows;
  double v = 0.0;
  if (r >= TILE)
    v = (r - TILE == c && c < N) ? 1.0 : 0.0;
  else if (r == 0 && c < N)
    v = y[c] - (mean ? mean[c] : 0.0);
  A[n_pad + r + c * ld] = v;
}

int launch_grad_border(double* A, long ld, long n_pad, long N, const double* y, const double* mean,
                       long nrows, hipStream_t s) {
  long tot = nrows * n_pad;
  hipLaunchKernelGGL(grad_border_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, A, ld, n_pad, N,
                     y, mean, nrows);
  SGP_HIP(hipGetLastError());
  return 0;
}

// noise gradients: scalar -> out[0] = (alpha'alpha - tr Kinv)/2 ; diag -> out[i] = (alpha_i^2 - Kinv_ii)/2
__global__ void grad_noise_kernel(const double* Kinv, long ldk, const double* alpha, long N, int diag,
                                  double* out) {
  __shared__ double sh[4];
  double acc = 0.0;
  for (long i = threadIdx.x; i < N; i += blockDim.x) {
    double g = 0.5 * (alpha[i] * alpha[i] - Kinv[i + i * ldk]);
    if (diag) out[i] = g;
    acc += g;
  }
  for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0 && !diag) out[0] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// dense Sigma_y: out[r + c * N] = (alpha_r alpha_c - Kinv[r, c]) / 2 = d logpdf / d Sigma_y[r, c] (the cotangent G itself)
__global__ void grad_noise_dense_kernel(const double* Kinv, long ldk, const double* alpha, long N, double* out) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= N * N) return;
  const long r = idx % N, c = idx / N;
  out[idx] = 0.5 * (alpha[r] * alpha[c] - Kinv[r + c * ldk]);
}
int launch_grad_noise_dense(const double* Kinv, long ldk, const double* alpha, long N, double* out, hipStream_t s) {
  const long tot = N * N;
  if (tot <= 0) return 0;
  hipLaunchKernelGGL(grad_noise_dense_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, Kinv, ldk, alpha, N, out);
  SGP_HIP(hipGetLastError());
  return 0;
}

// ... its columns [c0, c0 + w) only, into an N x w slab (ld ldo): the sharded gradient, where a rank holds the columns of C^-1 of
// its own panels (multi.hip)
__global__ void grad_noise_dense_cols_kernel(const double* Kinv, long ldk, const double* alpha, long N, long c0, long w,
                                             double* out, long ldo) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= N * w) return;
  const long r = idx % N, c = idx / N;
  out[r + c * ldo] = 0.5 * (alpha[r] * alpha[c0 + c] - Kinv[r + (c0 + c) * ldk]);
}
int launch_grad_noise_dense_cols(const double* Kinv, long ldk, const double* alpha, long N, long c0, long w, double* out, long ldo,
                                 hipStream_t s) {
  const long tot = N * w;
  if (tot <= 0) return 0;
  hipLaunchKernelGGL(grad_noise_dense_cols_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, Kinv, ldk, alpha, N, c0, w,
                     out, ldo);
  SGP_HIP(hipGetLastError());
  return 0;
}

int launch_grad_noise(const double* Kinv, long ldk, const double* alpha, long N, int diag, double* out,
                      hipStream_t s) {
  hipLaunchKernelGGL(grad_noise_kernel, dim3(1), dim3(256), 0, s, Kinv, ldk, alpha, N, diag, out);
  SGP_HIP(hipGetLastError());
  return 0;
}

// sum_i w[i] d var_i / d theta for the diagonal of a block (kernelmatrix_diag): per term
//   out_coef[t] = sum_i w_i rs_i cs_i k_t(x_i, x'_i),  out_scale[t] = sum_i w_i coef rs_i cs_i dk_t/dg
__global__ __launch_bounds__(256) void diag_grad_kernel(const double* w, long n, const DevTerm* terms,
                                                        double* out_coef, double* out_scale) {
  __shared__ double sh[2][256];
  const DevTerm T = terms[blockIdx.x];
  double a = 0.0, b = 0.0;
  for (long i = threadIdx.x; i < n; i += 256) {
    double d2 = 0.0;
    for (int d = 0; d < T.dim; ++d) {
      double df = T.xr[i * T.ldr + d] - T.xc[i * T.ldc + d];
      d2 = fma(df, df, d2);
    }
    double k, dk;
    kern_and_dscale(T.kind, d2, T.param, k, dk);
    double ww = w[i] * (T.rs ? T.rs[i] : 1.0) * (T.cs ? T.cs[i] : 1.0);
    a = fma(ww, k, a);
    b = fma(ww * T.coef, dk, b);
  }
  sh[0][threadIdx.x] = a;
  sh[1][threadIdx.x] = b;
  __syncthreads();
  for (int off = 128; off >= 1; off >>= 1) {
    if ((int)threadIdx.x < off) {
      sh[0][threadIdx.x] += sh[0][threadIdx.x + off];
      sh[1][threadIdx.x] += sh[1][threadIdx.x + off];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out_coef[blockIdx.x] = sh[0][0];
    out_scale[blockIdx.x] = sh[1][0];
  }
}

// per point i of a diagonal term (function-valued scales, product.jl:25-48): out_rs[i] += w_i coef cs_i k(x_i, x'_i),
// out_cs[i] += w_i coef rs_i k(x_i, x'_i)
__global__ void diag_scale_grad_kernel(const double* w, long n, DevTerm T, double* out_rs, double* out_cs) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double d2 = 0.0;
  for (int d = 0; d < T.dim; ++d) {
    const double df = T.xr[i * T.ldr + d] - T.xc[i * T.ldc + d];
    d2 = fma(df, df, d2);
  }
  double k, dk;
  kern_and_dscale(T.kind, d2, T.param, k, dk);
  k *= w[i] * T.coef;
  if (out_rs) out_rs[i] += k * (T.cs ? T.cs[i] : 1.0);
  if (out_cs) out_cs[i] += k * (T.rs ? T.rs[i] : 1.0);
}

int launch_diag_scale_grad(const double* w, long n, const DevTerm& T, double* out_rs, double* out_cs, hipStream_t s) {
  if (n <= 0 || (!out_rs && !out_cs)) return 0;
  hipLaunchKernelGGL(diag_scale_grad_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, w, n, T, out_rs, out_cs);
  SGP_HIP(hipGetLastError());
  return 0;
}

int launch_diag_grad(const double* w, long n, const DevTerm* d_terms, int nterms, double* out_coef,
                     double* out_scale, hipStream_t s) {
  if (nterms <= 0) return 0;
  hipLaunchKernelGGL(diag_grad_kernel, dim3((unsigned)nterms), dim3(256), 0, s, w, n, d_terms, out_coef, out_scale);
  SGP_HIP(hipGetLastError());
  return 0;
}

// ELBO gradient, M x M stage:  Z = I - B^-1 - u u',  S = B + B^-1 - 2 I + u u'  (all full, ld = m)
__global__ void vfe_zs_kernel(const double* B, const double* Binv, const double* u, double* Z, double* S,
                              long m) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= m * m) return;
  long r = idx % m, c = idx / m;
  double uu = u[r] * u[c], id = (r == c) ? 1.0 : 0.0, bi = Binv[idx];
  Z[idx] = id - bi - uu;
  S[idx] = B[idx] + bi - 2.0 * id + uu;
}

int launch_vfe_zs(const double* B, const double* Binv, const double* u, double* Z, double* S, long m,
                  hipStream_t s) {
  long tot = m * m;
  hipLaunchKernelGGL(vfe_zs_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, B, Binv, u, Z, S, m);
  SGP_HIP(hipGetLastError());
  return 0;
}

// ELBO gradient, per data point i (rows of R = A', RZ = A' Z):
//   ru = R[i,:] u,  ddelta = -delta + ru,  diagdot = R[i,:] . RZ[i,:] + delta ru
//   gy[i] = ddelta rsig ;  gsy[i] = rsig^2 (-1/2 + var rsig^2 / 2 - (ddelta delta + diagdot) / 2)
__global__ void vfe_rowstats_kernel(const double* R, long ld, const double* RZ, long ldrz, const double* u,
                                    const double* delta, const double* rsig, const double* var_x, long N,
                                    long m, double* gy, double* gsy) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  double ru = 0.0, dd = 0.0;
  for (long k = 0; k < m; ++k) {
    double r = R[i + k * ld];
    ru = fma(r, u[k], ru);
    dd = fma(r, RZ[i + k * ldrz], dd);
  }
  double d = delta[i], rs = rsig[i], is2 = rs * rs;
  double ddelta = ru - d;
  dd = fma(d, ru, dd);
  gy[i] = ddelta * rs;
  gsy[i] = is2 * (-0.5 + 0.5 * var_x[i] * is2 - 0.5 * (ddelta * d + dd));
}

int launch_vfe_rowstats(const double* R, long ld, const double* RZ, long ldrz, const double* u,
                        const double* delta, const double* rsig, const double* var_x, long N, long m,
                        double* gy, double* gsy, hipStream_t s) {
  hipLaunchKernelGGL(vfe_rowstats_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, R, ld, RZ, ldrz, u,
                     delta, rsig, var_x, N, m, gy, gsy);
  SGP_HIP(hipGetLastError());
  return 0;
}

// G_xz[i, j] = rsig_i (E[i, j] + delta_i ut_j), in place on E (nrows x m, ld)
__global__ void vfe_gxz_kernel(double* E, long ld, const double* delta, const double* ut, const double* rsig,
                               long nrows, long m) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= nrows * m) return;
  long i = idx % nrows, j = idx / nrows;
  E[i + j * ld] = rsig[i] * fma(delta[i], ut[j], E[i + j * ld]);
}

int launch_vfe_gxz(double* E, long ld, const double* delta, const double* ut, const double* rsig, long nrows,
                   long m, hipStream_t s) {
  long tot = nrows * m;
  hipLaunchKernelGGL(vfe_gxz_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, E, ld, delta, ut, rsig,
                     nrows, m);
  SGP_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------------------
// Gradient w.r.t. the input points of a term's row input ("inputs"):
//   gx[d, i] += scale * sum_j G_ij coef rs_i cs_j kappa'(|x_i - x'_j|^2) 2 (x_i - x'_j)[d]
// for the stationary kernels kappa(d^2) of kern_and_dscale (dk/dg = 2 d^2 kappa').  One workgroup
// owns one 128-row tile of the block pair and walks over all its column tiles, so every output
// row is written by exactly one workgroup -- no atomics, deterministic; launches that add into the
// same input array are ordered on one stream.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ double kern_dd2(int kind, double d2, double param) {
  switch (kind) {
    case G_SE:
      return -0.5 * exp(-0.5 * d2);
    case G_M12: {
      double d = sqrt(d2);
      return d > 0.0 ? -0.5 * exp(-d) / d : 0.0;  // not differentiable at coincident points: subgradient 0
    }
    case G_M32:
      return -1.5 * exp(-1.7320508075688772 * sqrt(d2));
    case G_M52: {
      double l = 2.23606797749979 * sqrt(d2);
      return -(5.0 / 6.0) * (1.0 + l) * exp(-l);
    }
    default:
      return 0.0;  // white noise (a.e.), constant
  }
}

// kappa(d^2) itself, for the row-scale gradient (the value next to kern_dd2's derivative)
__device__ __forceinline__ double kern_val(int kind, double d2, double param) {
  double k, dk;
  kern_and_dscale(kind, d2, param, k, dk);
  return k;
}

// WS: also accumulate the gradient w.r.t. the term's ROW SCALE vector (function-scaled processes,
// /root/reference/src/affine_transformations/product.jl:25-48):  gsv[i] += scale * sum_j G_ij coef k_ij cs_j
// (K_ij = coef rs_i k_ij cs_j) -- the per-row sums this kernel forms anyway.  gx may then be NULL.
template <int DMAX, bool WS>
__global__ __launch_bounds__(256) void grad_inputs_kernel(const double* Gm, long sr, long sc, const double* alpha,
                                                          long r0, long nr, long c0, long nc, DevTerm T,
                                                          double scale, double* gx /* T.dim x nr, packed */,
                                                          double* gsv /* nr */) {
  // G(i, j) of this launch's (row point i, column point j) lives at Gm[(r0 + i) * sr + (c0 + j) * sc]:
  // (sr, sc) = (1, ld) for the matrix as stored, (ld, 1) to contract its transpose (column-side
  // gradients of a rectangular block: the caller swaps the term's row / column data).
  __shared__ double sx[TILE * DMAX];
  __shared__ double scs[TILE];
  __shared__ double comb[TILE * DMAX];
  const int t = threadIdx.x;
  const int trow = t & 127, th = t >> 7;
  const long lrow = (long)blockIdx.x * TILE + trow;  // row within the block
  const bool live = lrow < nr;
  const long grow = r0 + lrow;
  double xr[DMAX], acc[DMAX];
  double accs = 0.0;
#pragma unroll
  for (int d = 0; d < DMAX; ++d) {
    xr[d] = (live && d < T.dim) ? T.xr[lrow * T.ldr + d] : 0.0;
    acc[d] = 0.0;
  }
  const double ai = (alpha && live) ? alpha[grow] : 0.0;
  const double wrow = live ? T.coef * (T.rs ? T.rs[lrow] : 1.0) : 0.0;
  for (long ct = 0; ct * TILE < nc; ++ct) {
    __syncthreads();
    for (int idx = t; idx < TILE * DMAX; idx += 256) {
      int p = idx / DMAX, d = idx % DMAX;
      long lc = ct * TILE + p;
      sx[idx] = (lc < nc && d < T.dim) ? T.xc[lc * T.ldc + d] : 0.0;
    }
    if (t < TILE) {
      long lc = ct * TILE + t;
      scs[t] = (lc < nc) ? (T.cs ? T.cs[lc] : 1.0) : 0.0;  // 0 kills the padding columns
    }
    __syncthreads();
    if (live) {
      for (int p = th * 64; p < th * 64 + 64; ++p) {
        const long lc = ct * TILE + p;
        if (lc >= nc) break;
        const long gc = c0 + lc;
        const double gm = Gm[grow * sr + gc * sc];
        const double g = alpha ? 0.5 * (ai * alpha[gc] - gm) : gm;
        double df[DMAX], d2 = 0.0;
#pragma unroll
        for (int d = 0; d < DMAX; ++d) {
          df[d] = xr[d] - sx[p * DMAX + d];
          d2 = fma(df[d], df[d], d2);
        }
        const double w = 2.0 * g * wrow * scs[p] * kern_dd2(T.kind, d2, T.param);
#pragma unroll
        for (int d = 0; d < DMAX; ++d) acc[d] = fma(w, df[d], acc[d]);
        if (WS) accs = fma(g * T.coef * scs[p], kern_val(T.kind, d2, T.param), accs);
      }
    }
  }
  // combine the two column halves in fixed order, then add into the input's gradient
  __syncthreads();
  if (th == 1) {
#pragma unroll
    for (int d = 0; d < DMAX; ++d) comb[trow * DMAX + d] = acc[d];
    if (WS) scs[trow] = accs;   // the column scales are no longer needed
  }
  __syncthreads();
  if (th == 0 && live) {
    if (gx) {
#pragma unroll
      for (int d = 0; d < DMAX; ++d)
        if (d < T.dim) gx[lrow * T.dim + d] += scale * (acc[d] + comb[trow * DMAX + d]);
    }
    if (WS) gsv[lrow] += scale * (accs + scs[trow]);
  }
}

// Any input dimension (> 16): the same sums with the dimension walked in chunks of 16, twice per column tile -- pass 1
// forms the 64 squared distances of a thread's row (registers; column chunk in LDS, as assemble_bigd_kernel) and turns
// them into the weights w_j = 2 G_ij coef rs_i cs_j kappa'(d2_ij); pass 2 forms sum_j w_j (x_i - x'_j)[d] chunk by
// chunk and adds it into the gradient, which this workgroup owns for its 128 rows (no atomics, deterministic).
constexpr int GBD_CHUNK = 16;
template <bool WS>
__global__ __launch_bounds__(256) void grad_inputs_bigd_kernel(const double* Gm, long sr, long sc, const double* alpha,
                                                               long r0, long nr, long c0, long nc, DevTerm T,
                                                               double scale, double* gx, double* gsv) {
  __shared__ __attribute__((aligned(16))) double sx[TILE * GBD_CHUNK];
  __shared__ double scs[TILE];
  __shared__ double comb[TILE * GBD_CHUNK];
  const int t = threadIdx.x;
  const int trow = t & 127, th = t >> 7;
  const long lrow = (long)blockIdx.x * TILE + trow;
  const bool live = lrow < nr;
  const long grow = r0 + lrow;
  const int D = T.dim;
  const double ai = (alpha && live) ? alpha[grow] : 0.0;
  const double wrow = live ? T.coef * (T.rs ? T.rs[lrow] : 1.0) : 0.0;
  double accs = 0.0;
  for (long ct = 0; ct * TILE < nc; ++ct) {
    double w[64];
#pragma unroll
    for (int q = 0; q < 64; ++q) w[q] = 0.0;
    // ---- pass 1: squared distances
    for (int d0 = 0; d0 < D; d0 += GBD_CHUNK) {
      __syncthreads();
      for (int idx = t; idx < TILE * GBD_CHUNK; idx += 256) {
        const int p = idx / GBD_CHUNK, d = idx % GBD_CHUNK;
        const long lc = ct * TILE + p;
        sx[idx] = (lc < nc && d0 + d < D) ? T.xc[lc * T.ldc + d0 + d] : 0.0;
      }
      if (d0 == 0 && t < TILE) {
        const long lc = ct * TILE + t;
        scs[t] = (lc < nc) ? (T.cs ? T.cs[lc] : 1.0) : 0.0;   // 0 kills the padding columns
      }
      __syncthreads();
      if (live) {
        double xi[GBD_CHUNK];
#pragma unroll
        for (int d = 0; d < GBD_CHUNK; ++d) xi[d] = (d0 + d < D) ? T.xr[lrow * T.ldr + d0 + d] : 0.0;
#pragma unroll
        for (int q = 0; q < 64; ++q) {
          const double* sp = &sx[(th * 64 + q) * GBD_CHUNK];
          double s = 0.0;
#pragma unroll
          for (int d = 0; d < GBD_CHUNK; ++d) {
            const double df = xi[d] - sp[d];
            s = fma(df, df, s);
          }
          w[q] += s;
        }
      }
    }
    // ---- weights
    if (live) {
#pragma unroll
      for (int q = 0; q < 64; ++q) {
        const int p = th * 64 + q;
        const long lc = ct * TILE + p;
        double wq = 0.0;
        if (lc < nc) {
          const long gc = c0 + lc;
          const double gm = Gm[grow * sr + gc * sc];
          const double g = alpha ? 0.5 * (ai * alpha[gc] - gm) : gm;
          const double d2 = w[q];
          wq = 2.0 * g * wrow * scs[p] * kern_dd2(T.kind, d2, T.param);
          if (WS) accs = fma(g * T.coef * scs[p], kern_val(T.kind, d2, T.param), accs);
        }
        w[q] = wq;
      }
    }
    // ---- pass 2: sum_j w_j (x_i - x'_j)[d]
    if (gx)
      for (int d0 = 0; d0 < D; d0 += GBD_CHUNK) {
        __syncthreads();
        for (int idx = t; idx < TILE * GBD_CHUNK; idx += 256) {
          const int p = idx / GBD_CHUNK, d = idx % GBD_CHUNK;
          const long lc = ct * TILE + p;
          sx[idx] = (lc < nc && d0 + d < D) ? T.xc[lc * T.ldc + d0 + d] : 0.0;
        }
        __syncthreads();
        double part[GBD_CHUNK];
#pragma unroll
        for (int d = 0; d < GBD_CHUNK; ++d) part[d] = 0.0;
        if (live) {
          double xi[GBD_CHUNK];
#pragma unroll
          for (int d = 0; d < GBD_CHUNK; ++d) xi[d] = (d0 + d < D) ? T.xr[lrow * T.ldr + d0 + d] : 0.0;
#pragma unroll
          for (int q = 0; q < 64; ++q) {
            const double* sp = &sx[(th * 64 + q) * GBD_CHUNK];
#pragma unroll
            for (int d = 0; d < GBD_CHUNK; ++d) part[d] = fma(w[q], xi[d] - sp[d], part[d]);
          }
        }
        if (th == 1) {
#pragma unroll
          for (int d = 0; d < GBD_CHUNK; ++d) comb[trow * GBD_CHUNK + d] = part[d];
        }
        __syncthreads();
        if (th == 0 && live) {
#pragma unroll
          for (int d = 0; d < GBD_CHUNK; ++d)
            if (d0 + d < D) gx[lrow * D + d0 + d] += scale * (part[d] + comb[trow * GBD_CHUNK + d]);
        }
      }
  }
  if (WS) {
    __syncthreads();
    if (th == 1) scs[trow] = accs;
    __syncthreads();
    if (th == 0 && live) gsv[lrow] += scale * (accs + scs[trow]);
  }
}

int launch_grad_inputs(const double* Gm, long sr, long sc, const double* alpha, long r0, long nr, long c0, long nc,
                       const DevTerm& T, int dmax, double scale, double* gx, hipStream_t s, double* gsv) {
  if (nr <= 0 || nc <= 0) return 0;
  dim3 grid((unsigned)((nr + TILE - 1) / TILE)), block(256);
#define SGP_GI(DM)                                                                                                    \
  do {                                                                                                                \
    if (gsv)                                                                                                          \
      hipLaunchKernelGGL((grad_inputs_kernel<DM, true>), grid, block, 0, s, Gm, sr, sc, alpha, r0, nr, c0, nc, T, scale, \
                         gx, gsv);                                                                                    \
    else                                                                                                              \
      hipLaunchKernelGGL((grad_inputs_kernel<DM, false>), grid, block, 0, s, Gm, sr, sc, alpha, r0, nr, c0, nc, T,      \
                         scale, gx, gsv);                                                                             \
  } while (0)
  if (dmax <= 1) SGP_GI(1);
  else if (dmax <= 2) SGP_GI(2);
  else if (dmax <= 4) SGP_GI(4);
  else if (dmax <= 8) SGP_GI(8);
  else if (dmax <= 16) SGP_GI(16);
  else if (gsv)   // any dimension: chunked two-pass kernel
    hipLaunchKernelGGL((grad_inputs_bigd_kernel<true>), grid, block, 0, s, Gm, sr, sc, alpha, r0, nr, c0, nc, T, scale, gx, gsv);
  else
    hipLaunchKernelGGL((grad_inputs_bigd_kernel<false>), grid, block, 0, s, Gm, sr, sc, alpha, r0, nr, c0, nc, T, scale, gx, gsv);
#undef SGP_GI
  SGP_HIP(hipGetLastError());
  return 0;
}

// d / d(input points) of sum_i w_i var_i for one diagonal term: the row input gets
// +w_i coef rs_i cs_i kappa'(d2_i) 2 (xr_i - xc_i), the column input the negative (zero whenever the
// term reads one input on both sides).  One thread per point; launches into one array are ordered
// on the stream.
__global__ void diag_grad_inputs_kernel(const double* w, long n, DevTerm T, double* gxr, double* gxc) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double d2 = 0.0;
  for (int d = 0; d < T.dim; ++d) {
    double df = T.xr[i * T.ldr + d] - T.xc[i * T.ldc + d];
    d2 = fma(df, df, d2);
  }
  double c = 2.0 * w[i] * T.coef * (T.rs ? T.rs[i] : 1.0) * (T.cs ? T.cs[i] : 1.0) * kern_dd2(T.kind, d2, T.param);
  for (int d = 0; d < T.dim; ++d) {
    double df = T.xr[i * T.ldr + d] - T.xc[i * T.ldc + d];
    gxr[i * T.dim + d] += c * df;
    if (gxc != gxr) gxc[i * T.dim + d] -= c * df;
  }
}

int launch_diag_grad_inputs(const double* w, long n, const DevTerm& T, double* gxr, double* gxc, hipStream_t s) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(diag_grad_inputs_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, w, n, T, gxr, gxc);
  SGP_HIP(hipGetLastError());
  return 0;
}

}  // namespace sgp
