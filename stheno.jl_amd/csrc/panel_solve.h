// Device code of the panel TRSM X <- X inv(L11)' (potrf.hip: panel_solve_kernel), shared with the dataflow
// factorisation (chol_df.hip): the fill of L11's 36 lower 16x16 blocks into LDS and the blocked substitution of one
// 16-row strip in the registers of one wave.
#pragma once
#include "common.h"
#include "potrf_diag.h"

namespace sgp {

// one wave per packed block of L11 (blocks wu, wu + NW, ...), four elements per lane: every load of a wave is independent
template <typename TS, int NW, typename LP = const TS*>   // LP: the pointer type of L (a global-address-space pointer where the caller has one)
__device__ __forceinline__ void panel_solve_fill(double* sL, LP L, long ldl, int wu, int lane) {
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
__device__ __forceinline__ void panel_solve_strip(TS* X_, long ldx, int loff, const double* sL, const double* inv_,
                                                  long inv_cstride, long inv_kstride, int lane) {
  // global address space, stated: reached through a noinline function or a struct the pointers are generic, FLAT loads and
  // stores return out of order, and the compiler can then only wait with vmcnt(0) -- see the note at the loads below
  __attribute__((address_space(1))) TS* X = (__attribute__((address_space(1))) TS*)X_;
  const __attribute__((address_space(1))) double* inv = (const __attribute__((address_space(1))) double*)inv_;
  const int l15 = lane & 15, lq = lane >> 4;
  const int aoff = lq * 16 + l15;  // A operand of k-step ks: [k = 4 ks + lq][m = l15]
  const int ioff = (int)(lq * inv_kstride + l15);
  d4 nb[8];  // nb[c][r] = -(B - sum X_p L_cp')[row l15][col 16 c + lq + 4 r]
#pragma unroll
  for (int c = 0; c < 8; ++c)
#pragma unroll
    for (int r = 0; r < 4; ++r) nb[c][r] = -(double)(X + (long)(16 * c + 4 * r) * ldx)[loff];
  // The inverse diagonal blocks come straight from global memory (16 KB in all, L1 / L2-resident): block c + 1's operand
  // registers are requested at the TOP of step c, before step c's results are stored.  gfx9's vmcnt counts loads and stores
  // in order, so a load issued AFTER the four stores of the previous step can only be waited for with vmcnt(0) -- which
  // also waits for those stores to drain (~1 us each step: the substitution of one tile took 13 us of which 8 were that);
  // issued before them it is vmcnt(4) and the stores stay in flight (round 5).
  double icc2[2][4];  // inv_c[m = l15][k = 4 ks + lq], double-buffered over c
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) icc2[0][ks] = (inv + ks * 4 * inv_kstride)[ioff];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const double* Lcc = sL + (c * (c + 1) / 2 + c) * 256 + aoff;
    if (c + 1 < 8) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) icc2[(c + 1) & 1][ks] = (inv + (c + 1) * inv_cstride + ks * 4 * inv_kstride)[ioff];
    }
    const double(&icc)[4] = icc2[c & 1];
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
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int c2 = c + 1; c2 < 8; ++c2)
        nb[c2] = mfma_f64(sL[(c2 * (c2 + 1) / 2 + c) * 256 + aoff + ks * 64], x[ks], nb[c2]);
    nb[c] = x;   // block c's right-hand side is spent: its registers keep the solution until the stores at the end
  }
  // all stores after the last load: inside the loop a store ahead of a load would put that load's wait behind the store's
  // completion (vmcnt is one in-order counter for both)
#pragma unroll
  for (int c = 0; c < 8; ++c)
#pragma unroll
    for (int r = 0; r < 4; ++r) (X + (long)(16 * c + 4 * r) * ldx)[loff] = (TS)nb[c][r];
}

}  // namespace sgp
