// Micro-benchmarks that pin the roofline denominators on the box:
// fp64 MFMA issue rate (and a lane-map self check with asymmetric data), HBM write / copy rate.
#include "common.h"
#include <cstdio>

namespace sgp {

// NACC independent accumulators per wave, 256 threads per block.  Also records the shader
// cycle count of the loop (s_memtime) so cycles/MFMA is known independently of the clock.
template <int NACC>
__global__ __launch_bounds__(256) void mfma_peak_kernel(double* out, int iters, double seed,
                                                        unsigned long long* cyc) {
  d4 a[NACC];
#pragma unroll
  for (int q = 0; q < NACC; ++q) a[q] = (d4){0, 0, 0, 0};
  double x = seed + threadIdx.x * 1e-3, y = seed - threadIdx.x * 2e-3;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int q = 0; q < NACC; ++q) a[q] = mfma_f64((q & 1) ? y : x, (q & 2) ? y : x, a[q]);
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  d4 s = a[0];
#pragma unroll
  for (int q = 1; q < NACC; ++q) s += a[q];
  out[(long)blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
  if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

// one wave: D = A(16x4) * B(4x16) with asymmetric integer data; writes D row-major [16][16]
// according to the documented lane map.  Host compares with the exact product.
__global__ void mfma_layout_kernel(const double* A /*16x4 row-major*/, const double* B /*4x16*/,
                                   double* D /*16x16 row-major*/) {
  int l = threadIdx.x;
  double a = A[(l & 15) * 4 + (l >> 4)];
  double b = B[(l >> 4) * 16 + (l & 15)];
  d4 acc = {0, 0, 0, 0};
  acc = mfma_f64(a, b, acc);
  for (int r = 0; r < 4; ++r) D[((l >> 4) + 4 * r) * 16 + (l & 15)] = acc[r];
}

__global__ void hbm_write_kernel(double2* dst, long n2) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long stride = (long)gridDim.x * blockDim.x;
  double2 v = {1.0, 2.0};
  for (; i < n2; i += stride) dst[i] = v;
}
__global__ void hbm_copy_kernel(const double2* src, double2* dst, long n2) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long stride = (long)gridDim.x * blockDim.x;
  for (; i < n2; i += stride) dst[i] = src[i];
}

template <int NACC>
static int mfma_variant(hipStream_t s, int blocks, int iters, double* d_out, unsigned long long* d_cyc,
                        double* tf, double* cyc_per_mfma, double* ghz) {
  hipEvent_t e0, e1;
  SGP_HIP(hipEventCreate(&e0));
  SGP_HIP(hipEventCreate(&e1));
  hipLaunchKernelGGL(mfma_peak_kernel<NACC>, dim3(blocks), dim3(256), 0, s, d_out, 16, 1.0, d_cyc);
  SGP_HIP(hipEventRecord(e0, s));
  hipLaunchKernelGGL(mfma_peak_kernel<NACC>, dim3(blocks), dim3(256), 0, s, d_out, iters, 1.0, d_cyc);
  SGP_HIP(hipEventRecord(e1, s));
  SGP_HIP(hipEventSynchronize(e1));
  float ms = 0;
  SGP_HIP(hipEventElapsedTime(&ms, e0, e1));
  unsigned long long cyc = 0;
  SGP_HIP(hipMemcpy(&cyc, d_cyc, sizeof(cyc), hipMemcpyDeviceToHost));
  double flops = (double)blocks * 4.0 * iters * NACC * 2048.0;
  *tf = flops / (ms * 1e-3) / 1e12;
  *cyc_per_mfma = (double)cyc / ((double)iters * NACC);
  *ghz = (double)cyc / (ms * 1e-3) / 1e9;  // only meaningful when the grid is one wave of blocks
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  return 0;
}

int run_mfma_bench(hipStream_t s, int iters, double* tflops_out, double* layout_maxerr_out) {
  const int blocks = 256 * 8;
  double* d_out = nullptr;
  unsigned long long* d_cyc = nullptr;
  SGP_HIP(hipMalloc(&d_out, sizeof(double) * blocks * 256));
  SGP_HIP(hipMalloc(&d_cyc, 8));
  double best = 0;
  // variants: (blocks per CU, accumulators).  Printed to stderr for the bring-up log.
  {
    double tf, c, g;
    int rc;
#define SGP_VAR(NACC, BPC)                                                              \
  rc = mfma_variant<NACC>(s, 256 * BPC, iters, d_out, d_cyc, &tf, &c, &g);             \
  if (rc) return rc;                                                                    \
  fprintf(stderr, "[mfma_f64] acc=%d blocks/CU=%d : %.2f TF/s, %.1f cycles/MFMA/wave, ~%.2f GHz*\n", \
          NACC, BPC, tf, c, g);                                                         \
  if (tf > best) best = tf;
    SGP_VAR(1, 1)
    SGP_VAR(2, 1)
    SGP_VAR(4, 1)
    SGP_VAR(8, 1)
    SGP_VAR(4, 2)
    SGP_VAR(8, 2)
    SGP_VAR(8, 8)
    SGP_VAR(16, 1)
#undef SGP_VAR
  }
  *tflops_out = best;

  // layout check
  double hA[64], hB[64], hD[256], ref[256];
  for (int m = 0; m < 16; ++m)
    for (int k = 0; k < 4; ++k) hA[m * 4 + k] = 1.0 + m * 7 + k * 3;
  for (int k = 0; k < 4; ++k)
    for (int n = 0; n < 16; ++n) hB[k * 16 + n] = 2.0 + k * 11 - n * 5;
  for (int m = 0; m < 16; ++m)
    for (int n = 0; n < 16; ++n) {
      double acc = 0;
      for (int k = 0; k < 4; ++k) acc += hA[m * 4 + k] * hB[k * 16 + n];
      ref[m * 16 + n] = acc;
    }
  double *dA, *dB, *dD;
  SGP_HIP(hipMalloc(&dA, sizeof(hA)));
  SGP_HIP(hipMalloc(&dB, sizeof(hB)));
  SGP_HIP(hipMalloc(&dD, sizeof(hD)));
  SGP_HIP(hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice));
  SGP_HIP(hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(mfma_layout_kernel, dim3(1), dim3(64), 0, s, dA, dB, dD);
  SGP_HIP(hipStreamSynchronize(s));
  SGP_HIP(hipMemcpy(hD, dD, sizeof(hD), hipMemcpyDeviceToHost));
  double me = 0;
  for (int i = 0; i < 256; ++i) {
    double e = hD[i] - ref[i];
    if (e < 0) e = -e;
    if (e > me) me = e;
  }
  *layout_maxerr_out = me;
  hipFree(dA);
  hipFree(dB);
  hipFree(dD);
  hipFree(d_out);
  hipFree(d_cyc);
  return 0;
}

int run_hbm_bench(hipStream_t s, long bytes, int iters, double* write_gbs, double* copy_gbs) {
  double2 *a = nullptr, *b = nullptr;
  long n2 = bytes / 16;
  SGP_HIP(hipMalloc(&a, n2 * 16));
  SGP_HIP(hipMalloc(&b, n2 * 16));
  hipEvent_t e0, e1;
  SGP_HIP(hipEventCreate(&e0));
  SGP_HIP(hipEventCreate(&e1));
  hipLaunchKernelGGL(hbm_write_kernel, dim3(4096), dim3(256), 0, s, a, n2);
  SGP_HIP(hipEventRecord(e0, s));
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(hbm_write_kernel, dim3(4096), dim3(256), 0, s, a, n2);
  SGP_HIP(hipEventRecord(e1, s));
  SGP_HIP(hipEventSynchronize(e1));
  float ms = 0;
  SGP_HIP(hipEventElapsedTime(&ms, e0, e1));
  *write_gbs = (double)n2 * 16.0 * iters / (ms * 1e-3) / 1e9;
  hipLaunchKernelGGL(hbm_copy_kernel, dim3(4096), dim3(256), 0, s, a, b, n2);
  SGP_HIP(hipEventRecord(e0, s));
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(hbm_copy_kernel, dim3(4096), dim3(256), 0, s, a, b, n2);
  SGP_HIP(hipEventRecord(e1, s));
  SGP_HIP(hipEventSynchronize(e1));
  SGP_HIP(hipEventElapsedTime(&ms, e0, e1));
  *copy_gbs = 2.0 * (double)n2 * 16.0 * iters / (ms * 1e-3) / 1e9;
  hipFree(a);
  hipFree(b);
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  return 0;
}

}  // namespace sgp
