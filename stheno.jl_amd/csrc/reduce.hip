// Small HBM-bound reductions / vector kernels around the factorisation: all deterministic
// (fixed-order tree reductions, no floating-point atomics) so results are bit-identical
// run to run.
#include "common.h"

namespace sgp {

__device__ __forceinline__ double block_sum_256(double v, double* sh) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) sh[w] = v;
  __syncthreads();
  double s = 0.0;
  if (threadIdx.x == 0) {
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) s += sh[i];
  }
  return s;  // valid in thread 0
}

// out[s] (+)= sum_{c < nc} rows[s + c*ld]^2     (|L^-1 (y - m)|^2 from the bordered rows)
__global__ void rowsumsq_kernel(const double* rows, long ld, long nc, double* out, int accumulate) {
  __shared__ double sh[4];
  const long s = blockIdx.x;
  double acc = 0.0;
  for (long c = threadIdx.x; c < nc; c += blockDim.x) {
    double v = rows[s + c * ld];
    acc = fma(v, v, acc);
  }
  double tot = block_sum_256(acc, sh);
  if (threadIdx.x == 0) out[s] = accumulate ? out[s] + tot : tot;
}

int launch_rowsumsq(const double* rows, long ld, long nc, long nrows, double* out, int accumulate,
                    hipStream_t s) {
  if (nrows <= 0) return 0;
  hipLaunchKernelGGL(rowsumsq_kernel, dim3((unsigned)nrows), dim3(256), 0, s, rows, ld, nc, out,
                     accumulate);
  SGP_HIP(hipGetLastError());
  return 0;
}

// Bordered rows after the factorisation: rows r < nrows of a local panel hold V' = K(x*, x) L^-T (columns =
// this panel's nc columns), zrow the transformed observation row z' = (L^-1 (y - m))'.  One thread per row,
// columns in order (coalesced across rows, deterministic):  sumsq[r] += sum_c V'[r,c]^2,
// dot[r] += sum_c V'[r,c] z[c]  -- the posterior variance reduction and mean shift of test point r.
__global__ void rows_dot_kernel(const double* rows, long ld, long nrows, long nc, const double* zrow,
                                double* sumsq, double* dot) {
  const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= nrows) return;
  double a = 0.0, b = 0.0;
  for (long c = 0; c < nc; ++c) {
    const double v = rows[r + c * ld];
    a = fma(v, v, a);
    b = fma(v, zrow[c * ld], b);
  }
  sumsq[r] += a;
  dot[r] += b;
}

int launch_rows_dot(const double* rows, long ld, long nrows, long nc, const double* zrow, double* sumsq,
                    double* dot, hipStream_t s) {
  if (nrows <= 0 || nc <= 0) return 0;
  hipLaunchKernelGGL(rows_dot_kernel, dim3((unsigned)((nrows + 127) / 128)), dim3(128), 0, s, rows, ld, nrows, nc,
                     zrow, sumsq, dot);
  SGP_HIP(hipGetLastError());
  return 0;
}

// rows [r0, r1) x nc columns of a column-major matrix = 0
__global__ void zero_rows_kernel(double* A, long ld, long r0, long r1, long nc) {
  const long h = r1 - r0;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= h * nc) return;
  A[r0 + idx % h + (idx / h) * ld] = 0.0;
}

int launch_zero_rows(double* A, long ld, long r0, long r1, long nc, hipStream_t s) {
  if (r1 <= r0 || nc <= 0) return 0;
  const long tot = (r1 - r0) * nc;
  hipLaunchKernelGGL(zero_rows_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, A, ld, r0, r1, nc);
  SGP_HIP(hipGetLastError());
  return 0;
}

__global__ void sum_array_kernel(const double* in, long n, double* out) {
  __shared__ double sh[4];
  double acc = 0.0;
  for (long i = threadIdx.x; i < n; i += blockDim.x) acc += in[i];
  double tot = block_sum_256(acc, sh);
  if (threadIdx.x == 0) out[0] = tot;
}

int launch_sum_array(const double* in, long n, double* out, hipStream_t s) {
  hipLaunchKernelGGL(sum_array_kernel, dim3(1), dim3(256), 0, s, in, n, out);
  SGP_HIP(hipGetLastError());
  return 0;
}

// out[s] = -(N log 2pi + logdet + sq[s]) / 2      (AbstractGPs.logpdf [EXT], App. A.3)
__global__ void logpdf_final_kernel(const double* logdet, const double* sq, long N, long ncols,
                                    double* out) {
  long s = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= ncols) return;
  out[s] = -0.5 * ((double)N * 1.8378770664093453 + logdet[0] + sq[s]);
}

int launch_logpdf_final(const double* d_logdet, const double* d_sq, long N, long ncols,
                        double* d_out, hipStream_t s) {
  hipLaunchKernelGGL(logpdf_final_kernel, dim3((unsigned)((ncols + 63) / 64)), dim3(64), 0, s,
                     d_logdet, d_sq, N, ncols, d_out);
  SGP_HIP(hipGetLastError());
  return 0;
}

// out[j] = prior[j] + sign * sum_{c<ncols} V[j + c*ld]^2   (posterior variance, App. A.5)
__global__ void rowsumsq_axpy_kernel(const double* V, long ld, long ncols, const double* prior,
                                     double* out, double sign) {
  __shared__ double sh[4];
  const long j = blockIdx.x;
  double acc = 0.0;
  for (long c = threadIdx.x; c < ncols; c += blockDim.x) {
    double v = V[j + c * ld];
    acc = fma(v, v, acc);
  }
  double tot = block_sum_256(acc, sh);
  if (threadIdx.x == 0) out[j] = prior[j] + sign * tot;
}

int launch_colsumsq_sub(const double* V, long ld, long nrows, long ncols, const double* prior,
                        double* out, double sign, hipStream_t s) {
  if (nrows <= 0) return 0;
  hipLaunchKernelGGL(rowsumsq_axpy_kernel, dim3((unsigned)nrows), dim3(256), 0, s, V, ld, ncols,
                     prior, out, sign);
  SGP_HIP(hipGetLastError());
  return 0;
}

// out[j] = add[j] + sum_{c<nc} rows[j + c*ld] * z[c*ldz]   (posterior mean m* + V' z; alpha = inv(L)' z)
// Column-oriented: a workgroup owns 32 consecutive rows, its 8 k-lanes stride over the columns, so
// every wave load is two contiguous 256-byte row segments of the column-major matrix; the k-lanes
// are combined through LDS in fixed order (deterministic).  upper_tri: row j is zero left of column
// 128 * floor(j / 128) (rows of inv(L)'), those columns are skipped.
__global__ __launch_bounds__(256) void gemv_rows_kernel(const double* rows, long ld, long nrows, long nc,
                                                        const double* z, long ldz, const double* add,
                                                        double* out, int upper_tri) {
  __shared__ double sh[8][33];
  const int r = threadIdx.x & 31, kq = threadIdx.x >> 5;
  const long j0 = (long)blockIdx.x * 32, j = j0 + r;
  const long cbeg = upper_tri ? (j0 / TILE) * TILE : 0;
  double acc = 0.0;
  if (j < nrows) {
    const double* p = rows + j;
    long c = cbeg + kq;
    for (; c + 24 < nc; c += 32) {
      double a0 = p[c * ld], a1 = p[(c + 8) * ld], a2 = p[(c + 16) * ld], a3 = p[(c + 24) * ld];
      double z0 = z[c * ldz], z1 = z[(c + 8) * ldz], z2 = z[(c + 16) * ldz], z3 = z[(c + 24) * ldz];
      acc = fma(a0, z0, acc);
      acc = fma(a1, z1, acc);
      acc = fma(a2, z2, acc);
      acc = fma(a3, z3, acc);
    }
    for (; c < nc; c += 8) acc = fma(p[c * ld], z[c * ldz], acc);
  }
  sh[kq][r] = acc;
  __syncthreads();
  if (kq == 0 && j < nrows) {
    double tot = 0.0;
#pragma unroll
    for (int q = 0; q < 8; ++q) tot += sh[q][r];
    out[j] = (add ? add[j] : 0.0) + tot;
  }
}

int launch_gemv_rows(const double* rows, long ld, long nrows, long nc, const double* z, long ldz,
                     const double* add, double* out, hipStream_t s, int upper_tri) {
  if (nrows <= 0) return 0;
  hipLaunchKernelGGL(gemv_rows_kernel, dim3((unsigned)((nrows + 31) / 32)), dim3(256), 0, s, rows, ld, nrows, nc,
                     z, ldz, add, out, upper_tri);
  SGP_HIP(hipGetLastError());
  return 0;
}

// dst[c + r*ldd] = src[r + c*lds] (+ add_vec[c])   -- tiled transpose, nr x nc -> nc x nr
__global__ void transpose_add_kernel(const double* src, long lds, long nr, long nc, double* dst,
                                     long ldd, const double* add_vec) {
  __shared__ double tile[32][33];
  long r0 = (long)blockIdx.x * 32, c0 = (long)blockIdx.y * 32;
  int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: ty 0..7
  for (int k = ty; k < 32; k += 8) {
    long r = r0 + tx, c = c0 + k;
    tile[k][tx] = (r < nr && c < nc) ? src[r + c * lds] : 0.0;
  }
  __syncthreads();
  for (int k = ty; k < 32; k += 8) {
    long c = c0 + tx, r = r0 + k;
    if (r < nr && c < nc) dst[c + r * ldd] = tile[tx][k] + (add_vec ? add_vec[c] : 0.0);
  }
}

int launch_transpose_add(const double* src, long lds, long nr, long nc, double* dst, long ldd,
                         const double* add_vec, hipStream_t s) {
  if (nr <= 0 || nc <= 0) return 0;
  dim3 grid((unsigned)((nr + 31) / 32), (unsigned)((nc + 31) / 32));
  hipLaunchKernelGGL(transpose_add_kernel, grid, dim3(256), 0, s, src, lds, nr, nc, dst, ldd,
                     add_vec);
  SGP_HIP(hipGetLastError());
  return 0;
}

// The transposition and the column sums of the ELBO's chunked pipeline in ONE pass over the solved rows (round 6: they were two,
// 2 x 2.1 GB per chunk at M = 4096): dst[c + r ldd] = v, part_dot[b][c] = sum_r v delta[r], part_sq[b][c] = sum_r v^2 over the
// rows of row block b (TC_RB rows), v = src[r + c lds] rs[r].  One workgroup: 32 columns x TC_RB rows, tile by tile;
// colsum_finish_kernel adds the blocks' partial sums in block order (deterministic).
constexpr long TC_RB = 2048;
__global__ __launch_bounds__(256) void transpose_colsum_kernel(const double* src, long lds, long nr, long nc, double* dst,
                                                               long ldd, const double* row_scale, const double* delta,
                                                               double* part_dot, double* part_sq) {
  __shared__ double tile[32][33];
  const long rb = blockIdx.x, c0 = (long)blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 256 threads: ty 0 .. 7, this thread's columns c0 + ty + 8 q
  double a[4] = {0.0, 0.0, 0.0, 0.0}, b[4] = {0.0, 0.0, 0.0, 0.0};
  const long r_end = min(nr, (rb + 1) * TC_RB);
  for (long r0 = rb * TC_RB; r0 < r_end; r0 += 32) {
    const long r = r0 + tx;
    const double rs = r < nr ? row_scale[r] : 0.0, dl = r < nr ? delta[r] : 0.0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int k = ty + 8 * q;
      const long c = c0 + k;
      const double v = (r < nr && c < nc) ? src[r + c * lds] * rs : 0.0;
      tile[k][tx] = v;
      a[q] = fma(v, dl, a[q]);
      b[q] = fma(v, v, b[q]);
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int k = ty + 8 * q;
      const long c = c0 + tx, rr = r0 + k;
      if (rr < nr && c < nc) dst[c + rr * ldd] = tile[tx][k];
    }
    __syncthreads();
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
#pragma unroll
    for (int off = 16; off >= 1; off >>= 1) {   // over tx: the 32 lanes of a half wave
      a[q] += __shfl_xor(a[q], off, 64);
      b[q] += __shfl_xor(b[q], off, 64);
    }
    const long c = c0 + ty + 8 * q;
    if (tx == 0 && c < nc) {
      part_dot[rb * nc + c] = a[q];
      part_sq[rb * nc + c] = b[q];
    }
  }
}
__global__ void colsum_finish_kernel(const double* part_dot, const double* part_sq, long nb, long nc, double* dots, double* sq,
                                     int accumulate) {
  const long c = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= nc) return;
  double a = 0.0, b = 0.0;
  for (long q = 0; q < nb; ++q) {
    a += part_dot[q * nc + c];
    b += part_sq[q * nc + c];
  }
  dots[c] = accumulate ? dots[c] + a : a;
  sq[c] = accumulate ? sq[c] + b : b;
}
long transpose_colsum_scratch(long nr, long nc) { return 2 * ((nr + TC_RB - 1) / TC_RB) * nc; }
int launch_transpose_colsum(const double* src, long lds, long nr, long nc, double* dst, long ldd, const double* row_scale,
                            const double* delta, double* dots, double* sq, int accumulate, double* scratch, hipStream_t s) {
  if (nr <= 0 || nc <= 0) return 0;
  const long nb = (nr + TC_RB - 1) / TC_RB;
  double* pd = scratch;
  double* ps = scratch + nb * nc;
  hipLaunchKernelGGL(transpose_colsum_kernel, dim3((unsigned)nb, (unsigned)((nc + 31) / 32)), dim3(256), 0, s, src, lds, nr, nc,
                     dst, ldd, row_scale, delta, pd, ps);
  hipLaunchKernelGGL(colsum_finish_kernel, dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, s, pd, ps, nb, nc, dots, sq,
                     accumulate);
  SGP_HIP(hipGetLastError());
  return 0;
}

// rows[j + c*ld] *= scale[j]   (Lambda_y^-1 scaling of K(x, z) rows for the ELBO, App. A.6)
__global__ void scale_rows_kernel(double* rows, long ld, long nrows, long nc, const double* scale) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= nrows * nc) return;
  long j = idx % nrows, c = idx / nrows;
  rows[j + c * ld] *= scale[j];
}

int launch_scale_rows(double* rows, long ld, long nrows, long nc, const double* scale,
                      hipStream_t s) {
  long tot = nrows * nc;
  if (tot <= 0) return 0;
  hipLaunchKernelGGL(scale_rows_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, rows,
                     ld, nrows, nc, scale);
  SGP_HIP(hipGetLastError());
  return 0;
}

}  // namespace sgp
