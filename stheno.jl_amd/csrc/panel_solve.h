// Device code of the panel TRSM X <- X inv(L11)' (potrf.hip: panel_solve_kernel), shared with the dataflow
// factorisation (chol_df.hip): the fill of L11's 36 lower 16x16 blocks into LDS and the blocked substitution of one
// 16-row strip in the registers of one wave.
#pragma once
#include "common.h"
#include "potrf_diag.h"

namespace sgp {

// one wave per packed block of L11 (blocks wu, wu + NW, ...), four elements per lane: every load of a wave is independent
template <typename TS, int NW>
__device__ __forceinline__ void panel_solve_fill(double* sL, const TS* L, long ldl, int wu, int lane) {
  for (int blk = wu; blk < 36; blk += NW) {
    int c, p;
    block_rc(blk, c, p);
    double v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int e = lane + 64 * q, k = e >> 4, m = e & 15;
      v[q] = (double)L[(16 * c + m) + (long)(16 * p + k) * ldl];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) sL[blk * 256 + lane + 64 * q] = v[q];
  }
}

// Blocked substitution over the eight 16-column blocks of L11 for the 16 rows whose element (row l15, column lq) sits at
// X[loff]:   T_c = B_c - sum_{p<c} X_p L_cp',   X_c = T_c inv(L_cc)' + one refinement step against L_cc.
// sL: the packed blocks of L11 in LDS (block (c, p) at (c (c + 1) / 2 + p) * 256, k-major); inv: the eight inverse
// diagonal blocks (block c at inv + c * inv_cstride, element [m][k] at + k * inv_kstride + m).
template <typename TS>
__device__ __forceinline__ void panel_solve_strip(TS* X, long ldx, int loff, const double* sL, const double* inv,
                                                  long inv_cstride, long inv_kstride, int lane) {
  const int l15 = lane & 15, lq = lane >> 4;
  const int aoff = lq * 16 + l15;  // A operand of k-step ks: [k = 4 ks + lq][m = l15]
  const int ioff = (int)(lq * inv_kstride + l15);
  d4 nb[8];  // nb[c][r] = -(B - sum X_p L_cp')[row l15][col 16 c + lq + 4 r]
#pragma unroll
  for (int c = 0; c < 8; ++c)
#pragma unroll
    for (int r = 0; r < 4; ++r) nb[c][r] = -(double)(X + (long)(16 * c + 4 * r) * ldx)[loff];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const double* Lcc = sL + (c * (c + 1) / 2 + c) * 256 + aoff;
    // inverse diagonal block: operand registers straight from global (16 KB in all, L1/L2-resident)
    double icc[4];  // inv_c[m = l15][k = 4 ks + lq]
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) icc[ks] = (inv + c * inv_cstride + ks * 4 * inv_kstride)[ioff];
    d4 nx1 = (d4){0.0, 0.0, 0.0, 0.0};  // -(T inv(L_cc)')
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) nx1 = mfma_f64(icc[ks], nb[c][ks], nx1);
    d4 rr = -nb[c];  // T - X1 L_cc'
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) rr = mfma_f64(Lcc[ks * 64], nx1[ks], rr);
    d4 x = -nx1;  // X1 + R inv(L_cc)'
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) x = mfma_f64(icc[ks], rr[ks], x);
#pragma unroll
    for (int r = 0; r < 4; ++r) (X + (long)(16 * c + 4 * r) * ldx)[loff] = (TS)x[r];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int c2 = c + 1; c2 < 8; ++c2)
        nb[c2] = mfma_f64(sL[(c2 * (c2 + 1) / 2 + c) * 256 + aoff + ks * 64], x[ks], nb[c2]);
  }
}

}  // namespace sgp
