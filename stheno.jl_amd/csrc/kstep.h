// One k-step (K = 4) of the 128 x 128 tile program: 12 explicit ds_read_b64 operand fetches (256 B/clk; hipcc fuses plain
// loads into ds_read2_b64 at half that rate) and the 32 v_mfma_f64_4x4x4_4b of a wave's 64 x 32 share (gemm_nt.hip explains
// the lane maps).  Shared by the dataflow factorisation (chol_df.hip) and the batched panel update (gemm_nt.hip:
// gemm_nt_seg_kernel); gemm_nt_dma_tile keeps its own unrolled copy (its ISA is pinned by the bench).
#pragma once
#include "common.h"

namespace sgp {

template <int KS>
__device__ __forceinline__ void tile_kstep(double (&acc)[8][4], unsigned a_addr, unsigned b_addr) {
  double a_r[4], b_c[8];
  constexpr int O = KS * 4 * LDS_LD * 8;   // byte offset of k-step KS inside a stage
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(a_r[0]) : "v"(a_addr), "i"(O + 0));
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(a_r[1]) : "v"(a_addr), "i"(O + 128));
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(a_r[2]) : "v"(a_addr), "i"(O + 256));
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(a_r[3]) : "v"(a_addr), "i"(O + 384));
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(b_c[0]) : "v"(b_addr), "i"(O + 0));
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(b_c[1]) : "v"(b_addr), "i"(O + 32));
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(b_c[2]) : "v"(b_addr), "i"(O + 64));
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(b_c[3]) : "v"(b_addr), "i"(O + 96));
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(b_c[4]) : "v"(b_addr), "i"(O + 128));
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(b_c[5]) : "v"(b_addr), "i"(O + 160));
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(b_c[6]) : "v"(b_addr), "i"(O + 192));
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(b_c[7]) : "v"(b_addr), "i"(O + 224));
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);  // keep the MFMAs below the wait (asm is opaque to hipcc)
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[j][i] = mfma44_f64(b_c[j], a_r[i], acc[j][i]);
}

// The same k-step for a wave whose 64 x 32 share of a DIAGONAL tile is only partly needed (chol_df.hip: the lower 16 x 16
// blocks feed the diagonal-block routine, the strictly upper ones are never read).  With rb = 4 wr + i the 16-row block and
// cb = 2 wc + (j >> 2) the 16-column block of fragment (j, i), the fragment is needed iff rb >= cb, i.e. i - (j >> 2) + D >= 0
// with D = 4 wr - 2 wc.  D >= 1: every fragment (use tile_kstep); D = 0: all but (i = 0, j >= 4) -- 28 of 32; D = -2:
// (i = 2, j < 4) and (i = 3, all j) -- 12 of 32; D <= -3: none.  Same per-element operation sequence as tile_kstep: the bits of
// the needed entries do not depend on which form computed them.
template <int KS, int D>
__device__ __forceinline__ void tile_kstep_lower(double (&acc)[8][4], unsigned a_addr, unsigned b_addr) {
  static_assert(D == 0 || D == -2, "tile_kstep_lower: D = 0 or -2");
  double a_r[4], b_c[8];
  constexpr int O = KS * 4 * LDS_LD * 8;
  if (D == 0) {
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(a_r[0]) : "v"(a_addr), "i"(O + 0));
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(a_r[1]) : "v"(a_addr), "i"(O + 128));
  }
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(a_r[2]) : "v"(a_addr), "i"(O + 256));
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(a_r[3]) : "v"(a_addr), "i"(O + 384));
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(b_c[0]) : "v"(b_addr), "i"(O + 0));
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(b_c[1]) : "v"(b_addr), "i"(O + 32));
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(b_c[2]) : "v"(b_addr), "i"(O + 64));
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(b_c[3]) : "v"(b_addr), "i"(O + 96));
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(b_c[4]) : "v"(b_addr), "i"(O + 128));
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(b_c[5]) : "v"(b_addr), "i"(O + 160));
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(b_c[6]) : "v"(b_addr), "i"(O + 192));
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(b_c[7]) : "v"(b_addr), "i"(O + 224));
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (i - (j >> 2) + D >= 0) acc[j][i] = mfma44_f64(b_c[j], a_r[i], acc[j][i]);
}

// Which 64 x 32 share of a DIAGONAL tile wave w takes when only the lower 16 x 16 blocks are needed, and what it executes of
// it (chol_df.hip: df_contract_diag explains the deal: 32 / 32 / 40 / 40 fragments per SIMD instead of 60 / 44 / 28 / 12):
//   wave 0 -> share (1, 0) all 32     wave 4 -> share (0, 2) none
//   wave 1 -> share (1, 1) all 32     wave 5 -> share (0, 3) none
//   wave 2 -> share (0, 0) 28         wave 6 -> share (0, 1) 12
//   wave 3 -> share (1, 2) 28         wave 7 -> share (1, 3) 12
// shape: 1 = every fragment (tile_kstep), 0 / -2 = tile_kstep_lower<., 0 / -2>, -9 = none
__device__ __forceinline__ void diag_share(int w, int& wr, int& wc, int& shape) {
  wr = (0x8B >> w) & 1;            // {1, 1, 0, 1, 0, 0, 0, 1}
  wc = (0xDE84 >> (2 * w)) & 3;    // {0, 1, 0, 2, 2, 3, 1, 3}
  const int d = 4 * wr - 2 * wc;
  shape = d >= 1 ? 1 : (d == 0 ? 0 : (d == -2 ? -2 : -9));
}

}  // namespace sgp
