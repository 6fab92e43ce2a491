// Device code of the 128x128 diagonal-block Cholesky (potrf_diag_body), shared by potrf_diag_kernel
// (potrf.hip) and the fused trailing-update + next-diagonal-block kernel (gemm_nt.hip).
#pragma once
#include "common.h"
#include <cstdlib>

namespace sgp {

__device__ __forceinline__ double bcast_lane(double v, int srclane) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, srclane);
  hi = __builtin_amdgcn_readlane(hi, srclane);
  return __hiloint2double(hi, lo);
}

// s_setprio level of the latency-critical panel kernels' waves.  Under the look-ahead they share CUs with
// trailing-update GEMM waves issuing MFMAs back to back; a raised wave priority wins the issue arbitration for the
// serial pivot / substitution chains (measured with the fused launches, 0 vs 3: N = 16384 33.2 -> 32.7 ms,
// 4096 2.73 -> 2.71, 2048 1.093 -> 1.087).
inline int panel_prio() { return 3; }
__device__ __forceinline__ void set_wave_prio(int p) {
  if (p == 1) __builtin_amdgcn_s_setprio(1);
  if (p == 2) __builtin_amdgcn_s_setprio(2);
  if (p >= 3) __builtin_amdgcn_s_setprio(3);
}

constexpr int PD_THREADS = 512;
// LDS: the 36 lower 16x16 blocks of the tile, block (rb, cb) at boff(rb, cb), element (m, k) of a
// block at [k * 16 + m] (k-major: one MFMA operand k-step = 64 consecutive doubles, conflict-free),
// plus the 16x16 inverse of the current diagonal block.  75.8 KB and <= 128 VGPRs: the workgroup
// fits into the slot one trailing-update GEMM workgroup (73.7 KB, 8 waves x 128 VGPRs) leaves on a
// CU, so the high-priority panel stream gets onto the chip while the update of the previous panel
// is still running (a 149 KB tile had to wait for an entirely idle CU, i.e. for the update's tail).
constexpr size_t PD_LDS = (size_t)(36 * 256 + 256) * sizeof(double);
__device__ __forceinline__ int boff(int rb, int cb) { return (rb * (rb + 1) / 2 + cb) * 256; }

// wave-uniform (rb, cb) of packed lower block `blk` (blk = rb (rb + 1) / 2 + cb)
__device__ __forceinline__ void block_rc(int blk, int& rb, int& cb) {
  rb = 0;
#pragma unroll
  for (int q = 1; q < 8; ++q) rb += (q * (q + 1) / 2 <= blk) ? 1 : 0;
  cb = blk - rb * (rb + 1) / 2;
}

// 16x16 micro-Cholesky of the packed block Dcc in the registers of ONE wave.  128 dependent pivots per
// diagonal block are the serial critical path of the whole factorisation, and a wave issues in order, so
// what counts is the number of instructions per pivot.  Layout: lane i of every 16-lane DPP row holds row i
// of the block (L part, computed redundantly in each row) AND row i of the identity (inverse part: carried
// through the same right-looking updates it becomes inv(L)').  The rank-1 update of column c2 needs
// L[c2][j] in every lane: `v_fmac_f64_dpp ... row_newbcast:c2` reads it from lane c2 of the own 16-lane row
// inside the FMA itself (DPP on 64-bit VALU ops is legal on gfx90a+ exactly for row_newbcast) -- one
// instruction per updated element instead of two v_readlane + s_nop + FMA through SGPRs.  sqrt and
// reciprocal come from one v_rsq_f64 + two Newton steps; the correctly rounded root is finished off the
// dependency chain.  (`s_nop`: a DPP read of a VGPR written by the previous VALU instruction needs two wait states, a DPP op
// after an SALU write of EXEC -- the divergent column store of the previous pivot -- five; a DPP read needs two
// wait states, and the hazard recogniser does not see inside inline asm.)
#define SGP_DPP_BCAST(dst, src, N) \
  asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:" #N " row_mask:0xf bank_mask:0xf" : "=v"(dst) : "v"(src))
// rank-1 update of column N by a finished pivot column (lx: its L part, ix: its inverse part; both in the
// lane's registers): L[i][N] -= L[N][j] L[i][j] and Inv[i][N] -= L[N][j] Inv[i][j], L[N][j] broadcast from lane N
#define SGP_DPP_UPD1(lx, ix, N)                                                                                \
  asm volatile("v_fmac_f64_dpp %0, %2, -%2 row_newbcast:" #N " row_mask:0xf bank_mask:0xf\n\t"                 \
               "v_fmac_f64_dpp %1, %2, -%3 row_newbcast:" #N " row_mask:0xf bank_mask:0xf"                      \
               : "+v"(Lr[N]), "+v"(Ir[N])                                                                      \
               : "v"(lx), "v"(ix));
// Slot K (0..6) of pivot J: the updates pivot J - 1 deferred (columns N >= J + 1 with N % 7 == K).  A wave
// issues in order, so the 2 (14 - J) independent FMAs of a rank-1 update would sit between one pivot's
// dependency chain and the next; spread over the gaps of the NEXT pivot's chain (each of its ten steps waits
// ~20 cycles for its operand) they cost nothing.  Only column J + 1 -- the next pivot -- is updated at once.
#ifndef SGP_POTRF_NEWTON
#define SGP_POTRF_NEWTON 2   /* Newton steps on 1 / sqrt(pivot) inside the dependency chain (A/B: -DSGP_POTRF_NEWTON=1) */
#endif
#if SGP_POTRF_NEWTON == 2
#define SGP_NSLOT 7
#else
#define SGP_NSLOT 5
#endif
#define SGP_SLOT_N(J, K, N) \
  if (J >= 1 && N >= J + 1 && (N % SGP_NSLOT) == K) SGP_DPP_UPD1(lp, ip, N)
#define SGP_SLOT(J, K)                                                                                   \
  SGP_SLOT_N(J, K, 1) SGP_SLOT_N(J, K, 2) SGP_SLOT_N(J, K, 3) SGP_SLOT_N(J, K, 4) SGP_SLOT_N(J, K, 5)   \
  SGP_SLOT_N(J, K, 6) SGP_SLOT_N(J, K, 7) SGP_SLOT_N(J, K, 8) SGP_SLOT_N(J, K, 9) SGP_SLOT_N(J, K, 10)  \
  SGP_SLOT_N(J, K, 11) SGP_SLOT_N(J, K, 12) SGP_SLOT_N(J, K, 13) SGP_SLOT_N(J, K, 14) SGP_SLOT_N(J, K, 15)
#define SGP_PIN(x) asm volatile("" : "+v"(x))   /* keeps x's computation between the neighbouring slots */
#define SGP_NEXT_N(J, N) \
  if (N == J + 1) SGP_DPP_UPD1(lij, iij, N)
#if SGP_POTRF_NEWTON == 2
#define SGP_NEWTON2(J)                                                                   \
    t = h * r;                                                                           \
    SGP_PIN(t);                                                                          \
    SGP_SLOT(J, 5)                                                                       \
    t = fma(t, r, 1.5);                                                                  \
    SGP_PIN(t);                                                                          \
    SGP_SLOT(J, 6)                                                                       \
    r = r * t; /* 1 / sqrt(djj) to rounding level */
#else
/* one Newton step: 1 / sqrt(djj) with a relative error of ~2^-51 (v_rsq_f64 seeds ~2^-26) scales the whole column --
   a componentwise backward error of two units in the last place, inside the factorisation's own rounding; the
   diagonal entry itself is still the corrected root */
#define SGP_NEWTON2(J)
#endif
#define SGP_PIVOT(J)                                                                     \
  {                                                                                      \
    double djj;                                                                          \
    SGP_DPP_BCAST(djj, Lr[J], J);                                                        \
    SGP_SLOT(J, 0)                                                                       \
    firstbad = (djj > 0.0) ? firstbad : min(firstbad, cb * 16 + J);                      \
    double h = -0.5 * djj;                                                               \
    double r = __builtin_amdgcn_rsq(djj);                                                \
    SGP_PIN(r);                                                                          \
    SGP_SLOT(J, 1)                                                                       \
    double t = h * r;                                                                    \
    SGP_PIN(t);                                                                          \
    SGP_SLOT(J, 2)                                                                       \
    t = fma(t, r, 1.5);                                                                  \
    SGP_PIN(t);                                                                          \
    SGP_SLOT(J, 3)                                                                       \
    r = r * t;                                                                           \
    SGP_PIN(r);                                                                          \
    SGP_SLOT(J, 4)                                                                       \
    SGP_NEWTON2(J)                                                                       \
    const double lij = Lr[J] * r, iij = Ir[J] * r;                                       \
    double d = djj * r;                                                                  \
    d = fma(0.5 * r, fma(-d, d, djj), d); /* sqrt(djj) */                                \
    asm volatile("s_nop 1" : "+v"(d)); /* DPP read of lij two wait states after its write */ \
    SGP_NEXT_N(J, 1) SGP_NEXT_N(J, 2) SGP_NEXT_N(J, 3) SGP_NEXT_N(J, 4) SGP_NEXT_N(J, 5)     \
    SGP_NEXT_N(J, 6) SGP_NEXT_N(J, 7) SGP_NEXT_N(J, 8) SGP_NEXT_N(J, 9) SGP_NEXT_N(J, 10)    \
    SGP_NEXT_N(J, 11) SGP_NEXT_N(J, 12) SGP_NEXT_N(J, 13) SGP_NEXT_N(J, 14) SGP_NEXT_N(J, 15) \
    /* column J is final: store it now (frees its registers).  One ds_write for the whole wave, no   \
       divergence (an SALU write of EXEC would cost the next DPP op five wait states): half 0 of each \
       32-lane group writes L[i][J] into the block, half 1 writes Inv[J][i] (lane 16 + i carries column \
       i of inv(L)); lanes 32..63 write the same values to the same addresses. */                    \
    {                                                                                    \
      double ls = lij, is = iij;                                                         \
      asm volatile("" : "+v"(ls), "+v"(is), "+v"(d)); /* the selects below stay behind the update above */ \
      const double vl = (i == J) ? d : ((i > J) ? ls : 0.0);                             \
      const double vi = (J >= i) ? is : 0.0;                                             \
      (hi ? sInv + i * 16 : Dcc + i)[hi ? J : J * 16] = hi ? vi : vl;                    \
    }                                                                                    \
    lp = lij;                                                                            \
    ip = iij;                                                                            \
  }

__device__ __forceinline__ void micro_cholesky(double* Dcc, double* sInv, int cb, int lane, int& firstbad) {
  const int i = lane & 15;
  const bool hi = (lane & 16) != 0;
  double Lr[16], Ir[16];
  double lp = 0.0, ip = 0.0;   // the previous pivot's column (its deferred updates run in this pivot's slots)
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    Lr[c] = Dcc[c * 16 + i];
    Ir[c] = (c == i) ? 1.0 : 0.0;
  }
  SGP_PIVOT(0) SGP_PIVOT(1) SGP_PIVOT(2) SGP_PIVOT(3) SGP_PIVOT(4) SGP_PIVOT(5) SGP_PIVOT(6) SGP_PIVOT(7)
  SGP_PIVOT(8) SGP_PIVOT(9) SGP_PIVOT(10) SGP_PIVOT(11) SGP_PIVOT(12) SGP_PIVOT(13) SGP_PIVOT(14) SGP_PIVOT(15)
}
#undef SGP_NEXT_N
#undef SGP_PIN
#undef SGP_SLOT
#undef SGP_SLOT_N
#undef SGP_DPP_UPD1
#undef SGP_PIVOT
#undef SGP_NEWTON2
#undef SGP_DPP_BCAST

// Right-looking over the eight 16-column sub-panels, software-pipelined so that the serial pivot chain
// of sub-panel cb + 1 (wave 0) runs under the trailing updates of sub-panel cb (waves 1..7):
//   A: wave 0: micro-Cholesky of block (cb, cb) + its inverse             | barrier
//   B: waves:  X = T[rb][cb] inv(L_cc)' (+ one refinement step), rb > cb  | barrier
//   C: wave 0: T[cb+1][cb+1] -= X X'  and straight on to A of cb + 1;
//      waves 1..7: every other trailing block T[rb][cc] -= X_rb X_cc', cb < cc <= rb
// Every MFMA chain on the critical path is 4 (update) or 12 (refined solve) instructions deep; the
// left-looking form this replaces accumulated up to 28 dependent MFMAs per sub-panel (44 -> ~27 us).
// TS = storage type of the tile in global memory: double, or float for the fp32 instantiation (f32.hip), whose
// diagonal blocks are factored HERE in fp64 -- converted on the way into and out of LDS -- so the serial chain
// is this kernel's, not a second, slower fp32 one.
// PRELOADED: the tile already sits in LDS in the packed layout (put there by the fused update's workgroup,
// gemm_nt.hip) and the caller has synchronised; the load phase is skipped.
template <bool DBG, typename TS, bool PRELOADED = false>
__device__ __forceinline__ void potrf_diag_body(TS* A, long ld, double* invd, double* logdet_slot, int* info,
                                                long gcol0, int prio, long long* dbg) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  set_wave_prio(prio);
  int nst = 0;
#define SGP_STAMP()                                                           \
  if (DBG && threadIdx.x == 0) dbg[nst] = (long long)__builtin_readcyclecounter(); \
  if (DBG) ++nst;
  SGP_STAMP()
  double* T = smem;               // packed lower blocks
  double* sInv = smem + 36 * 256; // [k][m]
  const int t = threadIdx.x;
  const int lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int l15 = lane & 15, lq = lane >> 4;
  const int aoff = lq * 16 + l15;  // operand element of k-step ks: + ks * 64

  // one wave per packed block (blocks w, w + 8, ...), four elements per lane; every load of the wave is
  // issued before the first LDS store, so the tile arrives in ONE memory round trip
  if (!PRELOADED) {
    double v[5][4];
#pragma unroll
    for (int it = 0; it < 5; ++it) {
      const int blk = w + 8 * it;
      int rb, cb;
      block_rc(blk < 36 ? blk : 35, rb, cb);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int e = lane + 64 * q, k = e >> 4, m = e & 15;
        const int r = rb * 16 + m, c = cb * 16 + k;
        v[it][q] = (blk < 36 && r >= c) ? (double)A[r + (long)c * ld] : 0.0;
      }
    }
#pragma unroll
    for (int it = 0; it < 5; ++it) {
      const int blk = w + 8 * it;
      if (blk < 36) {
#pragma unroll
        for (int q = 0; q < 4; ++q) T[blk * 256 + lane + 64 * q] = v[it][q];
      }
    }
    __syncthreads();
  }

  SGP_STAMP()
  int firstbad = 1 << 20;   // index of the first non-positive pivot (1 << 20: none)
  if (w == 0) micro_cholesky(T + boff(0, 0), sInv, 0, lane, firstbad);
  SGP_STAMP()
  __syncthreads();
  SGP_STAMP()

  for (int cb = 0; cb < 8; ++cb) {
    double* Dcc = T + boff(cb, cb);
    // the inverse diagonal block goes to global memory here, by the last 256 threads (one coalesced
    // 2 KB store off the critical path; later solves against the factor read it)
    if (t >= 256) invd[cb * 256 + (t - 256)] = sInv[t - 256];
    // (B) sub-panel solve: X = T[rb][cb] * inv(Lcc)^T for rb > cb, one row block per wave
    {
      const int rb = cb + 1 + w;
      if (rb < 8) {
        double* pc = T + boff(rb, cb) + aoff;  // element [row rb16 + l15][col cb16 + 4 ks + lq] at pc[ks * 64]
        const double* pi = sInv + aoff;        // Inv[m = l15][k = 4 ks + lq]
        const double* pl = Dcc + aoff;         // Lcc[m = l15][k = 4 ks + lq]
        d4 b;
#pragma unroll
        for (int r = 0; r < 4; ++r) b[r] = pc[r * 64];
        d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) acc = mfma_f64(pi[ks * 64], b[ks], acc);
        // acc[r] = sum_k Inv[lq+4r][k] * T[rb16+l15][cb16+k] = X[rb16+l15][cb16 + lq+4r]
        // A product with an explicit inverse is not backward stable (covariances of smooth kernels
        // cancel massively here), so one step of iterative refinement against Lcc itself follows:
        //   R = B - X Lcc',  X += R inv(Lcc)'
        // which restores substitution-level (LAPACK dtrsm) accuracy.  The accumulator lane map of one
        // product is the B-operand map of the next, so both extra products stay in registers.
        d4 nres = -b;  // -(R)[rb16+l15][cb16 + lq+4r]
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) nres = mfma_f64(pl[ks * 64], acc[ks], nres);
        d4 nx = -acc;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) nx = mfma_f64(pi[ks * 64], nres[ks], nx);  // -(X + R Inv')
#pragma unroll
        for (int r = 0; r < 4; ++r) pc[r * 64] = -nx[r];
      }
    }
    SGP_STAMP()
    __syncthreads();
    SGP_STAMP()
    if (cb == 7) break;
    // (C) trailing updates T[rb][cc] -= X_rb X_cc' (cb < cc <= rb).  Enumeration e = 0 is the next
    // diagonal block: wave 0 takes it alone and goes on to the next micro-Cholesky; blocks e >= 1 are
    // dealt round-robin to waves 1..7.
    {
      const int nrem = 7 - cb;                  // remaining block rows / columns
      const int nblk = nrem * (nrem + 1) / 2;
      for (int e = (w == 0 ? 0 : w); e < (w == 0 ? 1 : nblk); e += 7) {
        int rr, cc;
        block_rc(e, rr, cc);                    // lower-triangular enumeration of the trailing blocks
        const int rb = cb + 1 + rr, cn = cb + 1 + cc;
        const double* pa = T + boff(cn, cb) + aoff;  // A operand: X_cc[m = l15][k]
        const double* pb = T + boff(rb, cb) + aoff;  // B operand: X_rb[n = l15][k]
        d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) acc = mfma_f64(pa[ks * 64], pb[ks * 64], acc);
        // acc[r] = sum_k X[cn16 + lq + 4r][k] X[rb16 + l15][k]  ->  block (rb, cn), row l15, col lq + 4r
        double* pc = T + boff(rb, cn) + aoff;
#pragma unroll
        for (int r = 0; r < 4; ++r) pc[r * 64] -= acc[r];
      }
      // (A) of the next sub-panel: block (cb + 1, cb + 1) was updated by this very wave
      SGP_STAMP()
      if (w == 0) micro_cholesky(T + boff(cb + 1, cb + 1), sInv, cb + 1, lane, firstbad);
    }
    SGP_STAMP()
    __syncthreads();
    SGP_STAMP()
  }

  SGP_STAMP()
  // all 64 16x16 blocks of the tile, eight per wave: the 36 lower ones from LDS, the strictly upper ones
  // as zeros (products with the whole diagonal tile -- L Z of rand -- read them)
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int b = w + 8 * it, rb = b & 7, cbk = b >> 3;
    double v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = (rb >= cbk) ? T[boff(rb, cbk) + lane + 64 * q] : 0.0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int e = lane + 64 * q, k = e >> 4, m = e & 15;
      A[(rb * 16 + m) + (long)(cbk * 16 + k) * ld] = (TS)v[q];
    }
  }
  SGP_STAMP()
  // log-determinant from the finished diagonal (kept off the per-sub-panel critical path)
  double lg = 0.0;
  if (t < TILE) lg = log(T[boff(t >> 4, t >> 4) + (t & 15) * 17]);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) lg += __shfl_xor(lg, off, 64);
  __syncthreads();  // sInv is reused below
  if (t < TILE && lane == 0) sInv[w] = lg;
  __syncthreads();
  if (t == 0) {
    *logdet_slot = 2.0 * (sInv[0] + sInv[1]);
    if (firstbad < (1 << 20) && *info == 0) *info = (int)(gcol0 + firstbad + 1);
  }
  SGP_STAMP()
#undef SGP_STAMP
}

}  // namespace sgp
