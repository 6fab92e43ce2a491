// Diagonal-block kernels of the blocked Cholesky (the latency-bound critical path):
//   potrf_diag_kernel : Cholesky of one 128x128 diagonal block, entirely in LDS, one workgroup.
//   trtri_kernel      : W = inv(L11) (128x128 lower) so that the panel TRSM
//                       L21 = A21 * L11^-T becomes a plain MFMA GEMM (gemm_nt, B = W).
// Together they replace LAPACK dpotf2/dtrsm inside dpotrf on the reference path
// (LinearAlgebra.cholesky under AbstractGPs.logpdf/posterior/rand/elbo [EXT]).
//
// potrf_diag: left-looking over 16-column sub-panels.  The rank-k updates and the sub-panel
// solves run on v_mfma_f64_16x16x4_f64 straight out of LDS (ld 144: conflict-free operand
// reads); the 16x16 micro-Cholesky and its inverse run in the registers of one wave with
// v_readlane broadcasts (no LDS round trips, no barriers) -- "wavefront shuffles for the
// diagonal panel" in north-star terms.
#include "common.h"

namespace sgp {

__device__ __forceinline__ double bcast_lane(double v, int srclane) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, srclane);
  hi = __builtin_amdgcn_readlane(hi, srclane);
  return __hiloint2double(hi, lo);
}

constexpr int PD_THREADS = 512;
constexpr size_t PD_LDS = (size_t)(TILE * LDS_LD + 256) * sizeof(double);

__global__ __launch_bounds__(PD_THREADS) void potrf_diag_kernel(double* A, long ld, double* invd,
                                                                double* logdet_slot, int* info,
                                                                long gcol0) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* T = smem;                      // [128 cols][144]
  double* sInv = smem + TILE * LDS_LD;   // 16x16 col-major
  const int t = threadIdx.x;
  const int lane = t & 63, w = t >> 6;
  const int l15 = lane & 15, lq = lane >> 4;

  for (int idx = t; idx < TILE * TILE; idx += PD_THREADS) {
    int r = idx & 127, c = idx >> 7;
    T[c * LDS_LD + r] = (r >= c) ? A[r + (long)c * ld] : 0.0;
  }
  __syncthreads();

  double logacc = 0.0;
  int firstbad = -1;

  for (int cb = 0; cb < 8; ++cb) {
    // (1) left-looking update of sub-panel cb, row-blocks rb >= cb (one per wave)
    if (cb > 0) {
      int rb = cb + w;
      if (rb < 8) {
        d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
        for (int p = 0; p < cb; ++p) {
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) {
            int kcol = p * 16 + ks * 4 + lq;
            double aop = T[kcol * LDS_LD + cb * 16 + l15];  // L[cb16+m][k]
            double bop = T[kcol * LDS_LD + rb * 16 + l15];  // L[rb16+n][k]
            acc = mfma_f64(aop, bop, acc);
          }
        }
        // acc[r] = sum_k L[cb16 + lq+4r][k] * L[rb16 + l15][k]
#pragma unroll
        for (int r = 0; r < 4; ++r) T[(cb * 16 + lq + 4 * r) * LDS_LD + rb * 16 + l15] -= acc[r];
      }
    }
    __syncthreads();

    // (2) wave 0: 16x16 micro-Cholesky in registers; lanes 16..31 carry the rows of the identity
    // through the same right-looking updates (X <- X L^-T), so inv(L)^T falls out for free.
    if (w == 0) {
      const int i = l15;
      const bool lrow = lane < 16;       // lanes holding rows of the block itself
      double row[16];
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        double a = T[(cb * 16 + c) * LDS_LD + cb * 16 + i];
        row[c] = lrow ? a : ((c == i) ? 1.0 : 0.0);
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        double djj = bcast_lane(row[j], j);
        if (!(djj > 0.0) && firstbad < 0) firstbad = cb * 16 + j;
        // sqrt and reciprocal from one v_rsq_f64 + Newton steps: this chain is the serial
        // critical path of the whole factorisation (128 dependent pivots per diagonal block)
        double r = __builtin_amdgcn_rsq(djj);
        r = r * fma(-0.5 * djj, r * r, 1.5);
        r = r * fma(-0.5 * djj, r * r, 1.5);
        double d = djj * r;
        d = fma(0.5 * r, fma(-d, d, djj), d);      // d = sqrt(djj)
        double rinv = fma(r, fma(-d, r, 1.0), r);  // 1 / d
        double lij = row[j] * rinv;
        row[j] = (lrow && i == j) ? d : lij;
#pragma unroll
        for (int c2 = j + 1; c2 < 16; ++c2) {
          double lcj = bcast_lane(row[j], c2);  // L[c2][j] lives in lane c2
          row[c2] = fma(-lij, lcj, row[c2]);
        }
      }
      double dii = 0.0;
#pragma unroll
      for (int c = 0; c < 16; ++c) dii = (c == i) ? row[c] : dii;
      if (lrow) {
#pragma unroll
        for (int c = 0; c < 16; ++c) T[(cb * 16 + c) * LDS_LD + cb * 16 + i] = (c <= i) ? row[c] : 0.0;
      } else if (lane < 32) {
        // lane 16 + i holds row i of inv(L)^T, i.e. column i of inv(L): Inv[c][i] = row[c], c >= i
#pragma unroll
        for (int c = 0; c < 16; ++c) {
          double v = (c >= i) ? row[c] : 0.0;
          sInv[i * 16 + c] = v;  // col-major: Inv[r = c][col = i]
          invd[cb * 256 + i * 16 + c] = v;
        }
      }
      double lg = lrow ? log(dii) : 0.0;
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) lg += __shfl_xor(lg, off, 64);
      logacc += lg;
    }
    __syncthreads();

    // (3) sub-panel solve: X = T[rb][cb] * inv(Lcc)^T for rb > cb
    {
      int rb = cb + 1 + w;
      if (rb < 8) {
        d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          int kk = ks * 4 + lq;
          double aop = sInv[kk * 16 + l15];                         // Inv[m=l15][k]
          double bop = T[(cb * 16 + kk) * LDS_LD + rb * 16 + l15];  // T[rb16+n][cb16+k]
          acc = mfma_f64(aop, bop, acc);
        }
        // acc[r] = sum_k Inv[lq+4r][k] * T[rb16+l15][cb16+k] = X[rb16+l15][cb16 + lq+4r]
        // A product with an explicit inverse is not backward stable (covariances of smooth kernels
        // cancel massively here), so one step of iterative refinement against Lcc itself follows:
        //   R = B - X Lcc',  X += R inv(Lcc)'
        // which restores substitution-level (LAPACK dtrsm) accuracy.  The accumulator lane map of one
        // product is the B-operand map of the next, so both extra products stay in registers.
        d4 nres;  // -(R)[rb16+l15][cb16 + lq+4r]
#pragma unroll
        for (int r = 0; r < 4; ++r) nres[r] = -T[(cb * 16 + lq + 4 * r) * LDS_LD + rb * 16 + l15];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          double aop = T[(cb * 16 + ks * 4 + lq) * LDS_LD + cb * 16 + l15];  // Lcc[m=l15][k]
          nres = mfma_f64(aop, acc[ks], nres);                               // acc[ks] == X[n=l15][k=4ks+lq]
        }
        d4 nx;
#pragma unroll
        for (int r = 0; r < 4; ++r) nx[r] = -acc[r];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          double aop = sInv[(ks * 4 + lq) * 16 + l15];  // Inv[m=l15][k]
          nx = mfma_f64(aop, nres[ks], nx);             // -(X + R Inv')
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) T[(cb * 16 + lq + 4 * r) * LDS_LD + rb * 16 + l15] = -nx[r];
      }
    }
    __syncthreads();
  }

  for (int idx = t; idx < TILE * TILE; idx += PD_THREADS) {
    int r = idx & 127, c = idx >> 7;
    A[r + (long)c * ld] = (r >= c) ? T[c * LDS_LD + r] : 0.0;
  }
  if (t == 0) {
    *logdet_slot = 2.0 * logacc;
    if (firstbad >= 0 && *info == 0) *info = (int)(gcol0 + firstbad + 1);
  }
}

int launch_potrf_diag(double* A, long ld, double* d_invd, double* d_logdet_slot, int* d_info,
                      long gcol0, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    SGP_HIP(hipFuncSetAttribute((const void*)potrf_diag_kernel,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)PD_LDS));
    attr_set = true;
  }
  hipLaunchKernelGGL(potrf_diag_kernel, dim3(1), dim3(PD_THREADS), PD_LDS, s, A, ld, d_invd,
                     d_logdet_slot, d_info, gcol0);
  SGP_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------------------
// W = inv(L11).  Wave cb owns block-column cb of W, kept transposed in a 16 x 128 LDS strip:
//   strip[b*16 + j] = W[b][cb16 + j].
// Block recurrence: W[cb][cb] = invD[cb];  W[rb][cb] = -invD[rb] * sum_{p=cb}^{rb-1} L[rb][p] W[p][cb].
// The f64 MFMA result lane map (m = lq + 4r, n = l15) is exactly the B-operand map of four
// successive k-steps, so the second product consumes the first one's accumulator directly.
// ---------------------------------------------------------------------------------------
constexpr size_t TT_LDS = (size_t)8 * TILE * 16 * sizeof(double);

__global__ __launch_bounds__(512) void trtri_kernel(const double* L, long ld, const double* invd,
                                                    double* Wg) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int t = threadIdx.x;
  const int lane = t & 63, cb = t >> 6;
  const int l15 = lane & 15, lq = lane >> 4;
  double* strip = smem + cb * (TILE * 16);
  for (int idx = lane; idx < TILE * 16; idx += 64) strip[idx] = 0.0;
  for (int idx = lane; idx < 256; idx += 64) {
    int j = idx & 15, i = idx >> 4;
    strip[(cb * 16 + i) * 16 + j] = invd[cb * 256 + j * 16 + i];  // W[cb16+i][cb16+j]
  }
  for (int rb = cb + 1; rb < 8; ++rb) {
    d4 S = (d4){0.0, 0.0, 0.0, 0.0};
    for (int p = cb; p < rb; ++p) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        int k = p * 16 + ks * 4 + lq;
        double aop = L[(rb * 16 + l15) + (long)k * ld];  // L[rb16+m][k]
        double bop = strip[k * 16 + l15];                // W[k][cb16+n]
        S = mfma_f64(aop, bop, S);
      }
    }
    d4 T2 = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      double aop = invd[rb * 256 + (ks * 4 + lq) * 16 + l15];  // invD_rb[m=l15][k=4ks+lq]
      T2 = mfma_f64(aop, S[ks], T2);                           // S[ks] == S[k=4ks+lq][n=l15]
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) strip[(rb * 16 + lq + 4 * r) * 16 + l15] = -T2[r];
  }
  // Wg[b + (cb16+j)*128] = W[b][cb16+j]
  for (int idx = lane; idx < TILE * 16; idx += 64) {
    int b = idx & 127, j = idx >> 7;
    Wg[b + (cb * 16 + j) * TILE] = strip[b * 16 + j];
  }
}

int launch_trtri(const double* L, long ld, const double* d_invd, double* d_w, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    SGP_HIP(hipFuncSetAttribute((const void*)trtri_kernel,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)TT_LDS));
    attr_set = true;
  }
  hipLaunchKernelGGL(trtri_kernel, dim3(1), dim3(512), TT_LDS, s, L, ld, d_invd, d_w);
  SGP_HIP(hipGetLastError());
  return 0;
}

}  // namespace sgp
