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

// ---- alternative issue forms, to find the fastest fp64 path on gfx950 -------------------------
// MODE 0: 16x16x4 with AGPR accumulators (inline asm)      2048 flop / instr
// MODE 1: 4x4x4 (4 blocks)                                   512 flop / instr
// MODE 2: VALU v_fma_f64, 16 independent chains              128 flop / instr
// MODE 3: even waves MFMA (builtin), odd waves VALU -- do the two pipes overlap?
template <int MODE>
__global__ __launch_bounds__(256) void alt_peak_kernel(double* out, int iters, double seed,
                                                       unsigned long long* cyc) {
  double x = seed + threadIdx.x * 1e-3, y = seed - threadIdx.x * 2e-3;
  double res = 0.0;
  unsigned long long t0 = __builtin_readcyclecounter();
  const int wave = threadIdx.x >> 6;
  if (MODE == 0) {
    d4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0, a4 = a0, a5 = a0, a6 = a0, a7 = a0;
    for (int i = 0; i < iters; ++i) {
      asm volatile("v_mfma_f64_16x16x4_f64 %0, %8, %9, %0\n\t"
                   "v_mfma_f64_16x16x4_f64 %1, %9, %8, %1\n\t"
                   "v_mfma_f64_16x16x4_f64 %2, %8, %8, %2\n\t"
                   "v_mfma_f64_16x16x4_f64 %3, %9, %9, %3\n\t"
                   "v_mfma_f64_16x16x4_f64 %4, %8, %9, %4\n\t"
                   "v_mfma_f64_16x16x4_f64 %5, %9, %8, %5\n\t"
                   "v_mfma_f64_16x16x4_f64 %6, %8, %8, %6\n\t"
                   "v_mfma_f64_16x16x4_f64 %7, %9, %9, %7\n\t"
                   : "+a"(a0), "+a"(a1), "+a"(a2), "+a"(a3), "+a"(a4), "+a"(a5), "+a"(a6), "+a"(a7)
                   : "v"(x), "v"(y));
    }
    d4 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    res = s[0] + s[1] + s[2] + s[3];
  } else if (MODE == 1) {
    double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int q = 0; q < 8; ++q) a[q] = __builtin_amdgcn_mfma_f64_4x4x4f64((q & 1) ? y : x, (q & 2) ? y : x, a[q], 0, 0, 0);
    }
    for (int q = 0; q < 8; ++q) res += a[q];
  } else if (MODE == 4) {
    // 4x4x4 with per-lane pseudo-random operands (realistic bit toggling -> realistic power)
    double xs[8], ys[8], a[8];
    unsigned long long h = 0x9E3779B97F4A7C15ULL * (threadIdx.x + 1 + 256ULL * blockIdx.x);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ULL; h ^= h >> 32;
      xs[q] = (double)(h >> 11) * (1.0 / 9007199254740992.0) - 0.5;
      h *= 0x94D049BB133111EBULL; h ^= h >> 31;
      ys[q] = (double)(h >> 11) * (1.0 / 9007199254740992.0) - 0.5;
      a[q] = 0.0;
    }
    for (int i = 0; i < iters; i += 8) {
#pragma unroll
      for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int q = 0; q < 8; ++q)
          a[q] = __builtin_amdgcn_mfma_f64_4x4x4f64(xs[(q + 3 * r) & 7], ys[(q + r) & 7], a[q], 0, 0, 0);
    }
    for (int q = 0; q < 8; ++q) res += a[q];
  } else if (MODE == 2 || (MODE == 3 && (wave & 1))) {
    double a[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) a[q] = q;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int q = 0; q < 16; ++q) a[q] = fma(a[q], x, y);
    }
    for (int q = 0; q < 16; ++q) res += a[q];
  } else {
    d4 a[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) a[q] = (d4){0, 0, 0, 0};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int q = 0; q < 8; ++q) a[q] = mfma_f64((q & 1) ? y : x, (q & 2) ? y : x, a[q]);
    }
    d4 s = a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7];
    res = s[0] + s[1] + s[2] + s[3];
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  out[(long)blockIdx.x * blockDim.x + threadIdx.x] = res;
  if (blockIdx.x == 0 && (threadIdx.x == 0 || threadIdx.x == 64)) cyc[threadIdx.x >> 6] = t1 - t0;
}

template <int MODE>
static int alt_variant(hipStream_t s, int bpc, int iters, double* d_out, unsigned long long* d_cyc) {
  hipEvent_t e0, e1;
  SGP_HIP(hipEventCreate(&e0));
  SGP_HIP(hipEventCreate(&e1));
  int blocks = 256 * bpc;
  hipLaunchKernelGGL(alt_peak_kernel<MODE>, dim3(blocks), dim3(256), 0, s, d_out, 16, 1.0, d_cyc);
  SGP_HIP(hipEventRecord(e0, s));
  hipLaunchKernelGGL(alt_peak_kernel<MODE>, dim3(blocks), dim3(256), 0, s, d_out, iters, 1.0, d_cyc);
  SGP_HIP(hipEventRecord(e1, s));
  SGP_HIP(hipEventSynchronize(e1));
  float ms = 0;
  SGP_HIP(hipEventElapsedTime(&ms, e0, e1));
  unsigned long long cyc[2] = {0, 0};
  SGP_HIP(hipMemcpy(cyc, d_cyc, sizeof(cyc), hipMemcpyDeviceToHost));
  double waves = (double)blocks * 4.0;
  double fl;
  if (MODE == 0) fl = waves * iters * 8 * 2048.0;
  else if (MODE == 1 || MODE == 4) fl = waves * iters * 8 * 512.0;
  else if (MODE == 2) fl = waves * iters * 64 * 128.0;
  else fl = waves / 2 * iters * 8 * 2048.0 + waves / 2 * iters * 64 * 128.0;
  fprintf(stderr, "[alt mode %d] blocks/CU=%d: %.2f TF/s total, %.3f ms, wave0 %llu cyc, wave1 %llu cyc\n",
          MODE, bpc, fl / (ms * 1e-3) / 1e12, ms, cyc[0], cyc[1]);
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  return 0;
}

int run_alt_bench(hipStream_t s, int iters) {
  double* d_out = nullptr;
  unsigned long long* d_cyc = nullptr;
  SGP_HIP(hipMalloc(&d_out, sizeof(double) * 256 * 8 * 256));
  SGP_HIP(hipMalloc(&d_cyc, 16));
  for (int bpc : {1, 2, 4}) {
    int rc;
    if ((rc = alt_variant<0>(s, bpc, iters, d_out, d_cyc))) return rc;
    if ((rc = alt_variant<1>(s, bpc, iters, d_out, d_cyc))) return rc;
    if ((rc = alt_variant<2>(s, bpc, iters, d_out, d_cyc))) return rc;
    if ((rc = alt_variant<3>(s, bpc, iters, d_out, d_cyc))) return rc;
    if ((rc = alt_variant<4>(s, bpc, iters * 4, d_out, d_cyc))) return rc;
  }
  hipFree(d_out);
  hipFree(d_cyc);
  return 0;
}

// lane-map discovery for v_mfma_f64_4x4x4_4b_f64: one-hot A lane (la) x one-hot B lane (lb)
__global__ void mfma44_probe_kernel(double* out /*[64][64][64]*/) {
  const int la = blockIdx.x, lane = threadIdx.x;
  for (int lb = 0; lb < 64; ++lb) {
    double a = (lane == la) ? 1.0 : 0.0;
    double b = (lane == lb) ? 1.0 : 0.0;
    double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
    out[((long)la * 64 + lb) * 64 + lane] = d;
  }
}

int run_mfma44_probe(hipStream_t s) {
  double* d = nullptr;
  SGP_HIP(hipMalloc(&d, sizeof(double) * 64 * 64 * 64));
  hipLaunchKernelGGL(mfma44_probe_kernel, dim3(64), dim3(64), 0, s, d);
  SGP_HIP(hipStreamSynchronize(s));
  static double h[64 * 64 * 64];
  SGP_HIP(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
  hipFree(d);
  // print, for every A lane, the B lanes it pairs with and the output lane
  for (int la = 0; la < 64; ++la) {
    fprintf(stderr, "[mfma44] A lane %2d:", la);
    for (int lb = 0; lb < 64; ++lb)
      for (int l = 0; l < 64; ++l)
        if (h[((long)la * 64 + lb) * 64 + l] != 0.0) fprintf(stderr, " (B%d->D%d)", lb, l);
    fprintf(stderr, "\n");
  }
  return 0;
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
  if (iters >= 1000) { int rc2 = run_alt_bench(s, iters / 4); if (rc2) return rc2; }
  if (iters == 7) return run_mfma44_probe(s);

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
