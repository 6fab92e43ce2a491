// Dataflow blocked Cholesky: the whole bordered factorisation in ONE launch (replaces LAPACK dpotrf under
// LinearAlgebra.cholesky on the reference path -- SURVEY.md section 8a A2-A5 -- for the sizes where the launch-per-step
// schedules of capi.hip: chol_bordered are bound by their serial chain of ~3 launches per 128 columns).
//
// Left-looking by 128 x 128 tile.  Task (i, j), i >= j: tile row i, tile column j of the m_tot x n_pad factor matrix
//   acc  = -A_ij + sum_{k < j} L_ik L_jk'          the production GEMM tile program (gemm_nt.hip), K = 128 j
//   i == j : L_jj = chol(-acc)                     potrf_diag_body, straight from the accumulators (LDS hand-off)
//   i >  j : L_ij = (-acc) inv(L_jj)'              panel_solve_strip, eight waves x 16 rows
// Workgroups are persistent: each takes the next task id from a device counter (column-major task order = a topological
// order of the dependency graph, so whoever holds the earliest unfinished task can always finish it: no residency
// requirement, no deadlock) and walks its contraction as far as the two operand tile rows are final.  Progress is one
// counter per tile row (row i's tiles become final in increasing column order): prog[i] = number of final tiles of row
// i.  A workgroup that runs ahead of the diagonal chain simply accumulates the k blocks that exist and polls for the
// next: the look-ahead of the launch-based schedules falls out of the data dependencies, at tile granularity and with
// no kernel boundary, stream or event on the chain -- and the trailing work never waits for a whole panel.
//
// Arithmetic: every entry sees exactly the operations of the launch-based path in the same order (k ascending in steps
// of 4 through v_mfma_f64_4x4x4_4b; a stored and re-read fp64 accumulator is the same number), the same diagonal-block
// routine and the same refined substitution: the factor is BIT-IDENTICAL to chol_bordered's (tests/test_gpu_dataflow.py).
//
// Inter-workgroup visibility (per-XCD L2s are not coherent, a CU's L1 is never refreshed by other CUs' stores): the
// producer's waves drain their stores (s_waitcnt vmcnt(0)), barrier, one lane issues an agent-scope release fence
// (buffer_wbl2 sc1) + a second drain, then the relaxed agent-scope store of prog[]; a consumer polls prog[] with
// relaxed agent-scope loads from one lane, issues ONE agent-scope acquire (buffer_inv sc1) after the match, barrier,
// plain loads.  No tile is read by another workgroup before it is final, so no cache can hold a stale copy of it.
// Every wait is bounded: a poll that exceeds the limit raises the abort word (all workgroups leave) and reports through
// *info = SGP_DF_TIMEOUT, which the host turns into an error.
//
// Round 6 -- two generalisations of the same task loop:
//  * a PANEL launch of the sharded factorisation (csrc/multi.hip) may carry EXTERNAL SOURCES -- factored column panels left
//    of this matrix, received from another GPU: every task contracts their k blocks first (k ascending, exactly the order in
//    which the separate look-ahead update launch applied them) -- and may FACTOR only its first T_f tile columns: the tasks
//    of the columns from T_f on are update-only (contraction over the sources and the T_f factored columns, stored, not
//    solved, no progress published).  So "look-ahead update with the last received sub-panel + factorisation of the next
//    sub-panel + update of the panel's remaining columns" is ONE launch with tile-level dependencies: the diagonal chain
//    starts as soon as the first tile column has seen the sources and runs under the rest of the update.
//  * a BATCH of nb equally shaped, independent matrices as one task pool (sgp_logpdf_batch): ids are dealt round robin, every
//    matrix has its own progress counters; at sizes where one factorisation is bound by its diagonal chain (N <= 8192: 0.13
//    of the MFMA peak at N = 4096) the nb chains sit on different workgroups and hide each other.
#include "common.h"
#include "potrf_diag.h"
#include "panel_solve.h"
#include "df_tasks.h"
#include "kstep.h"
#include <vector>

namespace sgp {

constexpr int DF_KB = 16;                       // K chunk per LDS stage (as gemm_nt.hip)
constexpr int DF_STAGE = 2 * DF_KB * LDS_LD;    // doubles per stage: A chunk + B chunk
constexpr int DF_PROG = (int)SGP_DF_STATE_WORDS; // offset of the tile-row progress counters in the state words

struct DfArgs {
  DfProb p[DF_MAX_BATCH];   // the matrices of the launch: A (m_tot x n_pad, column-major, lower tiles + bordered rows), invall (T_c x
                            // INVD = 2048 doubles: inverse 16x16 diagonal blocks of every 128-block), slots (T_c logdet
                            // contributions), info
  int nb;           // how many (1: the plain factorisation)
  long ld;
  int T_r, T_c;     // tile rows (m_tot / 128), tile columns (n_pad / 128)
  int T_f;          // tile columns this launch FACTORS (<= T_c); the tasks of columns >= T_f are update-only
  int* state;       // [0] next task id, [1] abort, [DF_PROG + b * T_r + i] prog[i] of matrix b; zeroed before every launch
  long long spin_ticks;   // wall_clock64 ticks (100 MHz) a single wait may last
  long ntasks;      // nb * df_ntasks(T_r, T_c)
  const sz_word* nz;  // structural zeros (common.h): bit k of row i = tile (i, k) of the factor may be non-zero; nullptr = dense
  int nzw;            // words per row
  long gcol_base;     // global column of tile column 0 (the hybrid schedule factors PANELS with this kernel: PosDef info)
  int nz_t0;          // ... and its tile index: where the panel sits in the pattern
  DfExt ext[DF_MAX_EXT];   // external sources (factored panels left of this matrix), contracted first, in this order
  int n_ext;
  long long* stats;   // optional (SGP_DF_STATS): 8 tick counters per workgroup, see launch_chol_dataflow
  long long* cols;    // optional: 8 wall-clock stamps per tile column (the chain: diagonal task + the task below it)
};

#define DF_RLX_LOAD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)

// acc += A_panel[:, 16 c0 .. 16 c1) B_panel[:, 16 c0 .. 16 c1)' for one 128 x 128 tile; the chunk pipeline of
// gemm_nt_dma_tile (global -> LDS DMA one chunk ahead, two stages, raw barriers)
__device__ __forceinline__ void df_contract(const double* Ag, const double* Bg, long ld, long c0, long c1,
                                            double (&acc)[8][4], double* smem, int wu, int lane, unsigned a_off,
                                            unsigned b_off) {
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  const unsigned lds_base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) double*)smem;
  auto dma = [&](long k0, int stage) {
    double* sa = smem + stage * DF_STAGE;
    double* sb = sa + DF_KB * LDS_LD;
#pragma unroll
    for (int i = 0; i < DF_KB / 8; ++i) {  // wave w moves columns w and w + 8 of both operands
      const int col = wu + 8 * i;
      __builtin_amdgcn_global_load_lds((gptr_t)(Ag + 2 * lane + (k0 + col) * ld), (lptr_t)(sa + col * LDS_LD), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)(Bg + 2 * lane + (k0 + col) * ld), (lptr_t)(sb + col * LDS_LD), 16, 0, 0);
    }
  };
  dma(c0 * DF_KB, (int)(c0 & 1));
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  for (long c = c0; c < c1; ++c) {
    const int stage = (int)(c & 1);
    if (c + 1 < c1) dma((c + 1) * DF_KB, stage ^ 1);
    const unsigned a_addr = lds_base + (unsigned)(stage * DF_STAGE * 8) + a_off;
    const unsigned b_addr = lds_base + (unsigned)(stage * DF_STAGE * 8) + b_off;
    tile_kstep<0>(acc, a_addr, b_addr);
    tile_kstep<1>(acc, a_addr, b_addr);
    tile_kstep<2>(acc, a_addr, b_addr);
    tile_kstep<3>(acc, a_addr, b_addr);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
}


// The contraction of a DIAGONAL tile (round 5): only its lower 16 x 16 blocks are ever read (the diagonal-block routine takes
// them from LDS), 144 of the 256 MFMA fragments of the tile program.  The last k block of the diagonal tile is on the
// factorisation's critical chain -- it waits for the tile below the previous diagonal tile, and one CU needs 16 - 18 us for a
// full 128 x 128 x 128 product -- so the diagonal task runs a lower-only form: the eight 64 x 32 shares are dealt to the
// waves so that the two waves of every SIMD (wave w runs on SIMD w % 4) carry 32 / 32 / 40 / 40 fragments instead of
// 60 / 44 / 28 / 12, and a wave executes only the fragments with row block >= column block (kstep.h: tile_kstep_lower).
// (the deal and the three fragment shapes: kstep.h, diag_share / tile_kstep_lower)
__device__ __forceinline__ void df_contract_diag(const double* Ag, long ld, long c0, long c1, double (&acc)[8][4],
                                                 double* smem, int wu, int lane, unsigned a_off, unsigned b_off, int shape) {
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  const unsigned lds_base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) double*)smem;
  auto dma = [&](long k0, int stage) {
    double* sa = smem + stage * DF_STAGE;
    double* sb = sa + DF_KB * LDS_LD;
#pragma unroll
    for (int i = 0; i < DF_KB / 8; ++i) {  // (both operands are the tile row's own panel: the two stages hold the same rows)
      const int col = wu + 8 * i;
      __builtin_amdgcn_global_load_lds((gptr_t)(Ag + 2 * lane + (k0 + col) * ld), (lptr_t)(sa + col * LDS_LD), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)(Ag + 2 * lane + (k0 + col) * ld), (lptr_t)(sb + col * LDS_LD), 16, 0, 0);
    }
  };
  dma(c0 * DF_KB, (int)(c0 & 1));
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  for (long c = c0; c < c1; ++c) {
    const int stage = (int)(c & 1);
    if (c + 1 < c1) dma((c + 1) * DF_KB, stage ^ 1);
    const unsigned a_addr = lds_base + (unsigned)(stage * DF_STAGE * 8) + a_off;
    const unsigned b_addr = lds_base + (unsigned)(stage * DF_STAGE * 8) + b_off;
    if (shape == 1) {
      tile_kstep<0>(acc, a_addr, b_addr);
      tile_kstep<1>(acc, a_addr, b_addr);
      tile_kstep<2>(acc, a_addr, b_addr);
      tile_kstep<3>(acc, a_addr, b_addr);
    } else if (shape == 0) {
      tile_kstep_lower<0, 0>(acc, a_addr, b_addr);
      tile_kstep_lower<1, 0>(acc, a_addr, b_addr);
      tile_kstep_lower<2, 0>(acc, a_addr, b_addr);
      tile_kstep_lower<3, 0>(acc, a_addr, b_addr);
    } else if (shape == -2) {
      tile_kstep_lower<0, -2>(acc, a_addr, b_addr);
      tile_kstep_lower<1, -2>(acc, a_addr, b_addr);
      tile_kstep_lower<2, -2>(acc, a_addr, b_addr);
      tile_kstep_lower<3, -2>(acc, a_addr, b_addr);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
}

// thread 0: wait until min(prog[i], prog[j]) > have (returns min(.., cap)), or -1 on abort / timeout
__device__ __forceinline__ int df_wait(const DfArgs& a, int* prog, int* info, int i, int j, int have, int cap) {
  const long long t0 = wall_clock64();
  int avail;
  for (unsigned spins = 0;; ++spins) {
    const int pi = DF_RLX_LOAD(prog + i);
    const int pj = (i == j) ? pi : DF_RLX_LOAD(prog + j);
    avail = min(min(pi, pj), cap);
    if (avail > have) break;
    if (DF_RLX_LOAD(a.state + 1) != 0) return -1;
    __builtin_amdgcn_s_sleep(4);
    if ((spins & 63) == 63 && wall_clock64() - t0 > a.spin_ticks) {
      __hip_atomic_store(a.state + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      atomicCAS(info, 0, SGP_DF_TIMEOUT);
      return -1;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  return avail;
}

// thread 0: wait until *word >= target; 0, or -1 on abort / timeout
__device__ __forceinline__ int df_wait_word(const DfArgs& a, int* info, int* word, int target) {
  const long long t0 = wall_clock64();
  for (unsigned spins = 0;; ++spins) {
    if (DF_RLX_LOAD(word) >= target) break;
    if (DF_RLX_LOAD(a.state + 1) != 0) return -1;
    __builtin_amdgcn_s_sleep(2);
    if ((spins & 63) == 63 && wall_clock64() - t0 > a.spin_ticks) {
      __hip_atomic_store(a.state + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      atomicCAS(info, 0, SGP_DF_TIMEOUT);
      return -1;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  return 0;
}

// structural zeros: is tile (i, j) of the factor structurally non-zero?
// (nz_t0: the pattern is the whole matrix's; a panel factored by this kernel -- the hybrid schedule, the sharded sweep --
// sits nz_t0 tiles down and to the right in it, and its k blocks are the panel's own columns)
__device__ __forceinline__ bool df_nz(const DfArgs& a, int i, int j) {
  const int jj = j + a.nz_t0;
  return !a.nz || ((a.nz[(long)(i + a.nz_t0) * a.nzw + (jj >> 6)] >> (jj & 63)) & 1) != 0;
}
// the next run of k tiles in [k0, kend) (GLOBAL tile columns, k0 < kend) for which both pattern rows are non-zero:
// [ka, kb); ka == kend: none left
__device__ __forceinline__ void df_run(const sz_word* ri, const sz_word* rj, int k0, int kend, int& ka, int& kb) {
  ka = kend;
  kb = kend;
  const int qlast = (kend - 1) >> 6;
  int q = k0 >> 6;
  sz_word m = (ri[q] & rj[q]) & (~(sz_word)0 << (k0 & 63));
  while (m == 0 && q < qlast) {
    ++q;
    m = ri[q] & rj[q];
  }
  if (m == 0) return;
  const int first = q * 64 + __builtin_ctzll(m);
  if (first >= kend) return;
  ka = first;
  // the end of the run: the first k > ka that is not needed
  sz_word z = ~(ri[q] & rj[q]) & (~(sz_word)0 << (first & 63));
  while (z == 0 && q < qlast) {
    ++q;
    z = ~(ri[q] & rj[q]);
  }
  if (z != 0) kb = min(kend, q * 64 + (int)__builtin_ctzll(z));
}
// thread 0: the next run of the matrix's OWN k blocks >= k0 (and < jend) that task (i, j) has to contract -- both L_ik and
// L_jk structurally non-zero: [ka, kb); ka == jend: none left
__device__ __forceinline__ void df_next_run(const DfArgs& a, int i, int j, int k0, int jend, int& ka, int& kb) {
  const int t0 = a.nz_t0;
  df_run(a.nz + (long)(i + t0) * a.nzw, a.nz + (long)(j + t0) * a.nzw, k0 + t0, jend + t0, ka, kb);
  ka -= t0;
  kb -= t0;
}

// lane 0, after every wave has drained its stores and met at a barrier: release fence, drain, progress counter
__device__ __forceinline__ void df_release_store(int* word, int value) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (ROCm 7.2 may drop the wait after buffer_wbl2: restate it)
  __hip_atomic_store(word, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// lane 0: the next task -- tile (i << 16 | j), its matrix through *b -- or -1: none left / abort raised.  One counter:
// column-major order per matrix, the matrices of a batch round robin (df_tasks.h).
__device__ __forceinline__ int df_dequeue(const DfArgs& a, int* b) {
  if (DF_RLX_LOAD(a.state + 1) != 0) return -1;
  const int q = atomicAdd(a.state, 1);
  if ((long)q >= a.ntasks) return -1;
  long ql = q;
  *b = 0;
  if (a.nb > 1) df_batch_task((long)q, a.nb, *b, ql);
  int i, j;
  df_task_tile(ql, a.T_r, a.T_c, i, j);
  return (int)df_pack(i, j);
}

// The three phases of a task are separate (non-inlined) functions: each gets its own register allocation -- inlined
// into one loop body the diagonal-block routine (128 VGPRs), the substitution (wants > 200) and the contraction's 64
// accumulators + LDS pipeline spilled into each other (265 VGPR / 220 SGPR spills); nothing but (i, j) crosses a phase
// boundary: the accumulators leave through LDS (diagonal tile) or through the tile's own memory (off-diagonal).

// Phase 1: acc = -A_ij + sum_k L_ik L_jk' -- the external sources first, then the matrix's own columns as the operand rows
// become final.  Diagonal tile of a factored column: the result goes into potrf_diag_body's packed LDS layout; every other
// tile: T = -acc is stored in place.  Returns false on abort.
template <bool LOWER>   // LOWER: the diagonal tiles' lower-only contraction (the one-workgroup-per-CU instantiation: the sizes
                        // where the chain is the step; the lean kernel's register budget stays what it was)
__device__ __forceinline__ bool df_accumulate_body(const DfArgs& a, double* A, int* prog, int* info, int i, int j, double* smem,
                                                   int* s_word) {
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int w = t >> 6;
  const int wu = __builtin_amdgcn_readfirstlane(w);
  int wr = w >> 2, wc = w & 3, shape = 1;
  const bool fin = j < a.T_f;   // this launch factors column j (else: an update-only tile)
  const bool lower_only = LOWER && (i == j) && fin;   // the diagonal tile: lower blocks only, shares dealt for balance
  if (lower_only) diag_share(wu, wr, wc, shape);
  const int l15 = lane & 15, lq = lane >> 4, l3 = lane & 3;
  const unsigned a_off = (unsigned)((lq * LDS_LD + wr * 64 + l15) * 8);
  const unsigned b_off = (unsigned)((DF_KB * LDS_LD + lq * LDS_LD + wc * 32 + l3) * 8);
  const long ld = a.ld;
  const double* Ag = A + (long)i * TILE;
  const double* Bg = A + (long)j * TILE;
  double* Cg = A + ((long)i * TILE + wr * 64 + l15) + ((long)j * TILE + wc * 32 + lq) * ld;
  // acc[jj][ii] = -A[row 64 wr + 16 ii + l15][col 32 wc + 4 jj + lq]: the seed of C_new = -(acc + P P')
  double acc[8][4];
#pragma unroll
  for (int jj = 0; jj < 8; ++jj)
#pragma unroll
    for (int ii = 0; ii < 4; ++ii) acc[jj][ii] = Cg[ii * 16 + (long)(jj * 4) * ld];
#pragma unroll
  for (int jj = 0; jj < 8; ++jj)
#pragma unroll
    for (int ii = 0; ii < 4; ++ii) acc[jj][ii] *= -1.0;
  // ---- external sources: final before the launch (the host ordered it behind their arrival), no waits.  Every lane
  // derives the same runs from the pattern (wave-uniform values).
  for (int e = 0; e < a.n_ext; ++e) {
    const DfExt& x = a.ext[e];
    const double* Ae = x.base + (long)i * TILE;
    const double* Be = x.base + (long)j * TILE;
    int ka = x.kt0;
    const int kend = x.kt0 + x.kt;
    while (ka < kend) {
      int kb = kend;
      if (a.nz) {
        const int t0 = a.nz_t0;
        int ra, rb;
        df_run(a.nz + (long)(i + t0) * a.nzw, a.nz + (long)(j + t0) * a.nzw, ka, kend, ra, rb);
        ka = __builtin_amdgcn_readfirstlane(ra);
        kb = __builtin_amdgcn_readfirstlane(rb);
        if (ka >= kend) break;
      }
      const long c0 = (long)(ka - x.kt0) * (TILE / DF_KB), c1 = (long)(kb - x.kt0) * (TILE / DF_KB);
      if (lower_only)
        df_contract_diag(Ae, x.ld, c0, c1, acc, smem, wu, lane, a_off, b_off, shape);
      else
        df_contract(Ae, Be, x.ld, c0, c1, acc, smem, wu, lane, a_off, b_off);
      ka = kb;
    }
  }
  // ---- the matrix's own columns k < min(j, T_f)
  const int jend = min(j, a.T_f);
  int kdone = 0;
  while (kdone < jend) {
    if (t == 0) {
      const long long w0 = a.stats ? wall_clock64() : 0;
      int ka = kdone, kb = jend;
      if (a.nz) df_next_run(a, i, j, kdone, jend, ka, kb);   // skip the k blocks with a structurally zero operand tile
      s_word[6] = ka;
      s_word[1] = ka < jend ? df_wait(a, prog, info, i, j, ka, kb) : jend;
      if (a.stats) {
        const long long tot = ((long long)(unsigned)s_word[2] | ((long long)s_word[3] << 32)) + (wall_clock64() - w0);
        s_word[2] = (int)(unsigned)tot;
        s_word[3] = (int)(tot >> 32);
      }
    }
    __syncthreads();
    const int ka = __builtin_amdgcn_readfirstlane(s_word[6]);
    const int avail = __builtin_amdgcn_readfirstlane(s_word[1]);   // wave-uniform for the compiler too
    if (avail < 0) return false;
    if (ka >= jend) {
      __syncthreads();   // (s_word is rewritten by the caller's lane 0)
      break;
    }
    if (a.cols && t == 0 && i == j && avail == j) a.cols[(long)j * 8 + 1] = wall_clock64();
    if (lower_only)
      df_contract_diag(Ag, ld, (long)ka * (TILE / DF_KB), (long)avail * (TILE / DF_KB), acc, smem, wu, lane, a_off, b_off, shape);
    else
      df_contract(Ag, Bg, ld, (long)ka * (TILE / DF_KB), (long)avail * (TILE / DF_KB), acc, smem, wu, lane, a_off, b_off);
    kdone = avail;
  }
  if (a.cols && t == 0 && i == j) a.cols[(long)j * 8 + 2] = wall_clock64();
  if (i == j && fin) {
    // accumulators -> potrf_diag_body's packed LDS layout (as gemm_nt_dma_tile<HANDOFF>)
#pragma unroll
    for (int jj = 0; jj < 8; ++jj)
#pragma unroll
      for (int ii = 0; ii < 4; ++ii) {
        const int rb = 4 * wr + ii, cb = 2 * wc + (jj >> 2);
        const int k = 4 * (jj & 3) + lq;
        if (rb >= cb) smem[boff(rb, cb) + k * 16 + l15] = (rb > cb || l15 >= k) ? -acc[jj][ii] : 0.0;
      }
  } else {
#pragma unroll
    for (int jj = 0; jj < 8; ++jj)
#pragma unroll
      for (int ii = 0; ii < 4; ++ii) Cg[ii * 16 + (long)(jj * 4) * ld] = -acc[jj][ii];
  }
  return true;
}

// Phase 2a: Cholesky of the diagonal tile sitting in LDS
__device__ __forceinline__ void df_diag_body(const DfArgs& a, const DfProb& p, int j) {
  potrf_diag_body<false, double, true>(p.A + (long)j * TILE + (long)j * TILE * a.ld, a.ld, p.invall + (long)j * 2048,
                                       p.slots + j, p.info, a.gcol_base + (long)j * TILE, 0, nullptr);
}

// Phase 2b: L_ij = T inv(L_jj)' for the tile's 128 rows, eight waves x 16 rows
// (A, ld, invall by VALUE: through the `const DfArgs&` of a noinline phase function they are loaded from memory with vector
// loads at the top of the phase -- one more global round trip on the chain before the first useful load; round 5)
__device__ __forceinline__ void df_solve_body(double* A, long ld, const double* invall, int i, int j, double* smem) {
  const int t = threadIdx.x;
  const int lane = t & 63, w = t >> 6;
  const int l15 = lane & 15, lq = lane >> 4;
  const double* Ljj = A + (long)j * TILE + (long)j * TILE * ld;
  panel_solve_fill<double, 8>(smem, (const __attribute__((address_space(1))) double*)Ljj, ld, __builtin_amdgcn_readfirstlane(w), lane);
  __syncthreads();
  double* X = A + (long)i * TILE + (long)j * TILE * ld;
  const int loff = (int)(w * 16 + l15 + lq * ld);
  panel_solve_strip<double>(X, ld, loff, smem, invall + (long)j * 2048, 256, 16, lane);
}

// LEAN (two workgroups per CU, 128 VGPRs -- the sizes where the trailing contractions are the work): the phases are
// separate functions.  FAT (one workgroup per CU, 256 VGPRs -- the sizes where the diagonal chain is the critical path and
// occupancy buys nothing): everything inline, no spills in the substitution (228 VGPRs) or the diagonal-block routine.
__device__ __attribute__((noinline)) bool df_accumulate_lean(const DfArgs& a, double* A, int* prog, int* info, int i, int j,
                                                             double* smem, int* s_word) {
  return df_accumulate_body<false>(a, A, prog, info, i, j, smem, s_word);
}
__device__ __attribute__((noinline)) void df_diag_lean(const DfArgs& a, const DfProb& p, int j) { df_diag_body(a, p, j); }
__device__ __attribute__((noinline)) void df_solve_lean(double* A, long ld, const double* invall, int i, int j, double* smem) {
  df_solve_body(A, ld, invall, i, j, smem);
}
#define DF_FAT_FN __device__ __attribute__((noinline))
DF_FAT_FN bool df_accumulate_fat(const DfArgs& a, double* A, int* prog, int* info, int i, int j, double* smem, int* s_word) {
  return df_accumulate_body<true>(a, A, prog, info, i, j, smem, s_word);
}
DF_FAT_FN void df_diag_fat(const DfArgs& a, const DfProb& p, int j) { df_diag_body(a, p, j); }
DF_FAT_FN void df_solve_fat(double* A, long ld, const double* invall, int i, int j, double* smem) {
  df_solve_body(A, ld, invall, i, j, smem);
}
template <bool FAT>
__device__ __forceinline__ bool df_accumulate(const DfArgs& a, double* A, int* prog, int* info, int i, int j, double* smem,
                                              int* s_word) {
  if (FAT) return df_accumulate_fat(a, A, prog, info, i, j, smem, s_word);
  return df_accumulate_lean(a, A, prog, info, i, j, smem, s_word);
}
template <bool FAT>
__device__ __forceinline__ void df_diag(const DfArgs& a, const DfProb& p, int j) {
  if (FAT) df_diag_fat(a, p, j);
  else df_diag_lean(a, p, j);
}
template <bool FAT>
__device__ __forceinline__ void df_solve(const DfArgs& a, const DfProb& p, int i, int j, double* smem) {
  if (FAT) df_solve_fat(p.A, a.ld, p.invall, i, j, smem);
  else df_solve_lean(p.A, a.ld, p.invall, i, j, smem);
}

template <bool FAT>
__device__ __forceinline__ void chol_dataflow_body(const DfArgs& a) {
  extern __shared__ __attribute__((aligned(16))) double dyn_smem[];
  __shared__ int s_word[8];   // [0] task, [1] available k blocks / abort, [2..3] wait ticks (statistics), [4] the task's
                              // matrix (batch), [6] first k block of the run being contracted (structural zeros)
  const int t = threadIdx.x;
  // optional per-workgroup time accounting (lane 0, 100 MHz wall clock): [0] tasks [1] kernel [2] contraction incl. its
  // waits [3] wait for the diagonal tile [4] potrf [5] solve [6] publish + dequeue [7] of [2]: waiting
  long long tk0 = 0, tk = 0, acc_c = 0, acc_w = 0, acc_p = 0, acc_s = 0, acc_q = 0, ntask = 0;
  const bool st = a.stats != nullptr && t == 0;
  if (st) tk0 = wall_clock64();
  if (t == 0) {
    s_word[0] = df_dequeue(a, s_word + 4);
    s_word[2] = s_word[3] = 0;
  }
  // One lane-0 section per iteration (publish the finished tile AND take the next task), followed by the barrier at the
  // top: with a separate lane-0 `dequeue` at the top and lane-0 `publish` at the bottom the compiler threaded lane 0
  // from one straight into the other across the back edge, and the structurizer then ran lanes 1..63 of wave 0 into
  // the next iteration's barrier BEFORE lane 0's publish -- a deadlock.  Every value that steers control flow is read
  // through readfirstlane, so the task loop is scalar control flow for the compiler.
  for (;;) {
    __syncthreads();   // s_word[0] is set; the previous task's LDS phases are over for every wave
    const int q = __builtin_amdgcn_readfirstlane(s_word[0]);
    if (q < 0) break;
    const int b = __builtin_amdgcn_readfirstlane(s_word[4]);
    int j, i;
    df_unpack((uint32_t)q, i, j);
    const DfProb& p = a.p[b];
    int* prog = a.state + DF_PROG + (long)b * a.T_r;
    const bool fin = j < a.T_f;
    // the chain tasks run at raised wave priority: beside a contraction's back-to-back MFMAs the pivot chain of the
    // diagonal block and the substitution otherwise wait for issue slots (potrf 35 -> 100 us at N = 16384)
    if (a.nz && !df_nz(a, i, j)) {
      // a structurally zero tile: nothing to compute (its entries are the zeros the assembly wrote); the progress counter
      // of its row still moves in column order (an update-only column publishes nothing)
      __syncthreads();   // every wave has read s_word[0]
      if (t == 0) {
        int r = fin ? df_wait_word(a, p.info, prog + i, j) : 0;
        if (r == 0) {
          if (fin) df_release_store(prog + i, j + 1);
          r = df_dequeue(a, s_word + 4);
        }
        s_word[0] = r;
      }
      continue;
    }
    const bool chain = fin && (i == j || i == j + 1);
    if (chain) __builtin_amdgcn_s_setprio(3);
    else __builtin_amdgcn_s_setprio(0);
    if (st) tk = wall_clock64();
    if (st && a.cols && i == j) a.cols[(long)j * 8 + 0] = tk;
    if (!__builtin_amdgcn_readfirstlane((int)df_accumulate<FAT>(a, p.A, prog, p.info, i, j, dyn_smem, s_word))) break;
    if (st) {
      const long long n = wall_clock64();
      acc_c += n - tk;
      tk = n;
      ++ntask;
    }
    if (!fin) {
      // update-only: the tile is stored; the NEXT launch on the stream reads it (kernel boundary), nothing to publish
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (t == 0) s_word[0] = df_dequeue(a, s_word + 4);
      continue;
    }
    if (i == j) {
      __syncthreads();
      df_diag<FAT>(a, p, j);
      if (st) {
        const long long n = wall_clock64();
        acc_p += n - tk;
        tk = n;
        if (a.cols) a.cols[(long)j * 8 + 3] = n;
      }
    } else {
      if (t == 0) s_word[1] = df_wait(a, prog, p.info, j, j, j, j + 1);   // prog[j] == j + 1: the diagonal tile of column j is final
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (__builtin_amdgcn_readfirstlane(s_word[1]) < 0) break;
      if (st) {
        const long long n = wall_clock64();
        acc_w += n - tk;
        tk = n;
        if (a.cols && i == j + 1) a.cols[(long)j * 8 + 5] = n;
      }
      df_solve<FAT>(a, p, i, j, dyn_smem);
      if (st) {
        const long long n = wall_clock64();
        acc_s += n - tk;
        tk = n;
        if (a.cols && i == j + 1) a.cols[(long)j * 8 + 6] = n;
      }
    }
    // the tile is final: every wave drains its stores, then lane 0 publishes row i's progress and takes the next task
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t == 0) {
      // (with skipped k blocks this task may not have waited for every earlier tile of its row: prog[i] counts the final
      // tiles of row i in column order, so the count only moves past j once they all are)
      int r = a.nz ? df_wait_word(a, p.info, prog + i, j) : 0;
      if (r == 0) {
        df_release_store(prog + i, j + 1);
        r = df_dequeue(a, s_word + 4);
      }
      s_word[0] = r;
    }
    if (st) {
      const long long n = wall_clock64();
      acc_q += n - tk;
      if (a.cols && i == j) a.cols[(long)j * 8 + 4] = n;
      if (a.cols && i == j + 1) a.cols[(long)j * 8 + 7] = n;
    }
  }
  if (st) {
    long long* d = a.stats + (long)blockIdx.x * 8;
    // (bits 40..43: the XCD this workgroup ran on, XCC_ID)
    d[0] = ntask | ((long long)(__builtin_amdgcn_s_getreg(6164) & 15) << 40);
    d[1] = wall_clock64() - tk0;
    d[2] = acc_c;
    d[3] = acc_w;
    d[4] = acc_p;
    d[5] = acc_s;
    d[6] = acc_q;
    d[7] = (long long)(unsigned)s_word[2] | ((long long)s_word[3] << 32);   // ticks spent polling inside the contraction
  }
}

__global__ __launch_bounds__(512, 4) void chol_dataflow_kernel(DfArgs a) { chol_dataflow_body<false>(a); }
__global__ __launch_bounds__(512, 2) void chol_dataflow_fat_kernel(DfArgs a) { chol_dataflow_body<true>(a); }

// state words a launch of nb matrices with m_tot rows needs (the callers size d_state with it)
long df_state_words(long m_tot, int nb) { return SGP_DF_STATE_WORDS + (long)nb * (m_tot / TILE); }

static int df_launch(DfArgs& a, long n_pad, long m_tot, long ld, int n_wg, double timeout_s, int fat, hipStream_t s) {
  if (n_pad % TILE || m_tot % TILE || n_pad <= 0 || m_tot < n_pad) {
    set_error("chol_dataflow: sizes must be multiples of 128");
    return -1;
  }
  if (m_tot / TILE >= 32768) {   // a task is (row << 16 | column) in a non-negative int
    set_error("chol_dataflow: more than 32767 tile rows");
    return -1;
  }
  if ((long)16 * ld + m_tot >= (1L << 31)) {   // panel_solve_strip's 32-bit lane offset
    set_error("chol_dataflow: leading dimension too large");
    return -1;
  }
  SGP_LDS_ATTR_ONCE(chol_dataflow_kernel, PD_LDS);
  SGP_LDS_ATTR_ONCE(chol_dataflow_fat_kernel, PD_LDS);
  a.ld = ld;
  a.T_r = (int)(m_tot / TILE);
  a.T_c = (int)(n_pad / TILE);
  a.spin_ticks = (long long)(timeout_s * 1e8);
  a.ntasks = (long)a.nb * df_ntasks(a.T_r, a.T_c);
  if (a.ntasks >= (1L << 31)) {   // (the task counter is an int)
    set_error("chol_dataflow: too many tasks for one launch");
    return -1;
  }
  SGP_HIP(hipMemsetAsync(a.state, 0, sizeof(int) * (size_t)df_state_words(m_tot, a.nb), s));
  const long grid = std::min<long>(a.ntasks, n_wg);
  if (fat)
    hipLaunchKernelGGL(chol_dataflow_fat_kernel, dim3((unsigned)grid), dim3(512), PD_LDS, s, a);
  else
    hipLaunchKernelGGL(chol_dataflow_kernel, dim3((unsigned)grid), dim3(512), PD_LDS, s, a);
  SGP_HIP(hipGetLastError());
  return 0;
}

int launch_chol_dataflow(double* A, long ld, long n_pad, long m_tot, int* d_state, double* d_invall, double* d_slots,
                         int* d_info, int n_wg, double timeout_s, hipStream_t s, long long* d_stats, long long* d_cols,
                         int fat, const sz_word* d_nz, int nz_words, long gcol_base, const DfPanel* px) {
  DfArgs a;
  a.p[0] = DfProb{A, d_invall, d_slots, d_info};
  a.nb = 1;
  a.state = d_state;
  a.stats = d_stats;
  a.cols = d_stats ? d_cols : nullptr;
  a.nz = d_nz;
  a.nzw = nz_words;
  a.gcol_base = gcol_base;
  a.nz_t0 = (int)(gcol_base / TILE);
  a.T_f = (int)(n_pad / TILE);
  a.n_ext = 0;
  if (px) {
    if (px->n_fact % TILE || px->n_fact <= 0 || px->n_fact > n_pad || px->n_ext < 0 || px->n_ext > DF_MAX_EXT) {
      set_error("chol_dataflow: bad panel extension");
      return -1;
    }
    a.T_f = (int)(px->n_fact / TILE);
    a.n_ext = px->n_ext;
    for (int e = 0; e < px->n_ext; ++e) {
      a.ext[e] = px->ext[e];
      if (!a.ext[e].base || a.ext[e].kt <= 0 || a.ext[e].kt0 < 0 || a.ext[e].kt0 + a.ext[e].kt > a.nz_t0) {
        set_error("chol_dataflow: an external source must lie left of the panel");
        return -1;
      }
    }
  }
  return df_launch(a, n_pad, m_tot, ld, n_wg, timeout_s, fat, s);
}

// nb equally shaped independent matrices as one task pool (dense patterns: a batch shares no structure)
int launch_chol_dataflow_batch(const DfProb* probs, int nb, long ld, long n_pad, long m_tot, int* d_state, int n_wg,
                               double timeout_s, int fat, hipStream_t s) {
  if (nb < 1 || nb > DF_MAX_BATCH) {
    set_error("chol_dataflow: batch size out of range");
    return -1;
  }
  DfArgs a;
  for (int b = 0; b < nb; ++b) a.p[b] = probs[b];
  a.nb = nb;
  a.state = d_state;
  a.stats = nullptr;
  a.cols = nullptr;
  a.nz = nullptr;
  a.nzw = 0;
  a.gcol_base = 0;
  a.nz_t0 = 0;
  a.T_f = (int)(n_pad / TILE);
  a.n_ext = 0;
  return df_launch(a, n_pad, m_tot, ld, n_wg, timeout_s, fat, s);
}

}  // namespace sgp
