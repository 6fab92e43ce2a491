// fp64 MFMA GEMM kernels for gfx950.
//
// Instruction choice (measured on MI355X, profiles/archive/r01_microbench.md): v_mfma_f64_16x16x4_f64
// issues only every ~96 cycles per SIMD (49 TF/s chip-wide, 62 % of the 78.6 TF/s datasheet
// rate) while the 4-block v_mfma_f64_4x4x4_4b_f64 issues every 16 cycles (72-75 TF/s).
// The GEMMs therefore run on the 4x4x4 form.  Its lane maps (probed on hardware,
// tools/gpu_probe44.py; b = (l >> 2) & 3 is the block):
//   A-operand lane l holds A_b[i = l & 3][k = l >> 4]
//   B-operand lane l holds B_b[k = l >> 4][j = l & 3]
//   result    lane l holds D_b[i = l >> 4][j = l & 3]
// Feeding every block the same A (4 columns of C) and block b rows 4b..4b+3 of a 16-row
// fragment as B makes one instruction a 16(row) x 4(col) x 4(k) update whose result lanes run
// along the contiguous row dimension of C (row = l & 15, col = l >> 4).
//
// gemm_nt:  C = beta*C + alpha * A * B'   A: M x K, B: Nc x K, C: M x Nc, column-major.
//   One routine is the SYRK/GEMM trailing update of the blocked Cholesky (alpha=-1, beta=1,
//   lower tiles only), the panel TRSM (B = inv(L11), beta=0), the row-bordered forward
//   substitutions of logpdf / posterior / elbo, and L*Z for rand.  It replaces LAPACK dpotrf's
//   dsyrk/dgemm/dtrsm calls under LinearAlgebra.cholesky on the reference path
//   (SURVEY.md section 2 #10, section 8a A2-A5).
//
// Tiling: one 512-thread workgroup (8 waves as 2 x 4) per 128x128 tile of C, 64x32 per wave =
// 4 row fragments x 8 column fragments = 32 MFMA results (64 accumulator VGPRs), so two
// workgroups = 16 waves = 4 per SIMD are resident per CU.  K is consumed in chunks of 16 through
// two LDS stages.  Both operands are "row index contiguous", so one wave-instruction moves one
// 1 KiB column of a chunk; the LDS leading dimension 144 (== 16 mod 32) makes every ds_read_b64
// operand fetch conflict free (SQ_LDS_BANK_CONFLICT = 0).  The old C tile is read in the
// prologue straight into the accumulators (scaled by beta/alpha): the epilogue is store-only,
// each store instruction writing 4 columns x 16 consecutive rows (4 x 128-byte runs).
//
// Tile -> workgroup map: hardware places workgroup id on XCD id % 8.  XCD x owns one tile row
// out of every 8 (row 8j + x for even j, 8j + 7 - x for odd j, so that the live-tile counts under
// the lower-triangular mask are equal): every A row-panel is read through exactly one XCD's L2,
// and at any position in the grid all 8 XCDs see (almost) the same mask.  Inside an XCD the order is 8 owned rows x 8 tile columns, row fastest: 64
// consecutive workgroups form a patch sharing 8 A panels and 8 B panels in that XCD's L2.
#include "common.h"
#include "potrf_diag.h"
#include "tilemap.h"
#include "kstep.h"

namespace sgp {

constexpr int KB = 16;  // K chunk per LDS stage

// Workgroup -> tile enumeration: tilemap.h (host-compilable; tests/tilemap_host.cpp checks it exhaustively).
__device__ __forceinline__ bool tile_of_block(long n_tr, long n_tc, long mask_off, long& tr, long& tc) {
  return tile_of_id((long)blockIdx.x, n_tr, n_tc, mask_off, tr, tc);
}

// ---------------------------------------------------------------------------------------
// Production kernel: operands go global -> LDS directly (global_load_lds_dwordx4, no staging
// VGPRs, no ds_write), one chunk ahead of the MFMAs; raw s_barrier + counted waits; operand
// fetches are explicit ds_read_b64 (256 B/clk -- hipcc fuses plain loads into ds_read2_b64 at
// half that rate).  Measured 57 TF/s on a 32768^2 x 1024 lower update (profiles/).
// ---------------------------------------------------------------------------------------
// TAG only changes the symbol: <1> = the outer trailing updates of the blocked Cholesky (the launches
// bench.py times and the roofline line is about), <0> = every other use (inner K = 128 updates,
// solves' updates, Gram / inverse / L Z products), so that rocprofv3 --stats lists them apart.
// The tile program itself (one workgroup = one 128 x 128 tile of C), shared by the plain kernel and the fused
// update + next-diagonal-block kernel below.  smem: 2 stages x (A chunk + B chunk) = 73 728 bytes of LDS.
// Returns false for a workgroup without a tile; tr / tc = the tile it computed.
// HANDOFF: the workgroup of tile (0, 0) does not store its result but scatters it into LDS in potrf_diag_body's
// packed-block layout (lower 16x16 blocks, strictly upper entries of the diagonal blocks zeroed -- exactly what
// that routine's own load phase would have produced from the stored tile).
template <bool HANDOFF, bool STAMP = false>
__device__ __forceinline__ bool gemm_nt_dma_tile(const double* A, long lda, const double* B, long ldb, double* C,
                                                 long ldc, long K, double alpha, double beta, long mask_off,
                                                 long n_tr, long n_tc, long c_slice_stride, const double* Cin,
                                                 long ldcin, int klo, double* smem, long& tr, long& tc,
                                                 long long* dbg = nullptr, const TileSkip* sk = nullptr,
                                                 bool keep_first = false) {
  // STAMP (bench only, sgp_bench_gemm_stamps): s_memtime of thread 0 at the phase boundaries of the tile program
  long long st0 = 0, st1 = 0, st2 = 0, stA = 0;
  if (STAMP) st0 = (long long)__builtin_amdgcn_s_memtime();
  constexpr int NJ = 8, WCOLS = 32;
  if (klo == 3) {
    // Gram product split over K by XCD: workgroup id % 8 is the XCD the hardware puts it on, and that is
    // the K slice it contracts, so each XCD's L2 only ever holds its own slice of the operands (all
    // slices on all XCDs made the per-chunk working set 8 x 512 KB = the whole 4 MiB L2).  id / 8
    // enumerates the lower tiles linearly.
    // Round 4 -- whole rounds: an XCD has 64 workgroup slots and T (T + 1) / 2 tiles (528 at M = 4096: 8.25 rounds, the
    // last one a quarter full, ~8 % of the product's time).  The tiles beyond the last multiple of 64 ("leftover": 16) are
    // therefore split `sub` = 64 / leftover = 4 ways further over k, each piece into a slab of its own: 512 tiles x K and
    // 64 pieces x K / 4 = exactly 8.25 rounds of work on 8.25 rounds of slots.  mask_off carries `sub` (>= 1); slab of
    // (slice sl, piece q) = sl * sub + q, whole tiles use piece 0's slab; splitk_reduce_kernel sums them in that order.
    const long sub = mask_off > 1 ? mask_off : 1;
    const long tiles = n_tr * (n_tr + 1) / 2, full = sub > 1 ? tiles / 64 * 64 : tiles;
    const long sl = (long)blockIdx.x & 7;
    long k = (long)blockIdx.x >> 3, q = 0, kk = K;
    if (k >= full) {
      const long e = k - full;
      k = full + e / sub;
      q = e % sub;
      kk = K / sub;
    }
    tr = (long)((sqrt(8.0 * (double)k + 1.0) - 1.0) * 0.5);
    while (tr * (tr + 1) / 2 > k) --tr;
    while ((tr + 1) * (tr + 2) / 2 <= k) ++tr;
    tc = k - tr * (tr + 1) / 2;
    if (tr >= n_tr) return false;
    A += (sl * K + q * kk) * lda;
    B += (sl * K + q * kk) * ldb;
    C += (sl * sub + q) * c_slice_stride;
    K = kk;
  } else {
    if (sk && sk->cmap) {
      const int x = (int)(blockIdx.x & 7), pos = (int)(blockIdx.x >> 3);
      if (pos >= sk->cmap[x]) return false;
      if (!tile_of_id((long)sk->cmap[16 + (long)x * sk->cstride + pos] * 8 + x, n_tr, n_tc, mask_off, tr, tc)) return false;
    } else if (!tile_of_block(n_tr, n_tc, mask_off, tr, tc)) return false;
    // split-K (blockIdx.y > 0 only in launch_gemm_nt_splitk): slice s contracts columns
    // [s K, (s + 1) K) of A and B into its own slab of C
    A += (long)blockIdx.y * K * lda;
    B += (long)blockIdx.y * K * ldb;
    C += (long)blockIdx.y * c_slice_stride;
  }
  // structural zeros: the tile's update P[tr] P[tc]' is dead when, for every k tile of the panel, one of the two operand
  // tiles is structurally zero (keep_first: tile (0, 0) of a fused launch goes on to the diagonal-block routine anyway)
  // It contracts the k tiles from the first to the last needed one only (a panel that straddles a block boundary).
  if (sk && sk->nz) {
    if (sk->need0 >= 0) {   // a result tile nobody reads (launch_gemm_nt_uut: block pairs without terms)
      const sz_word* rn = sk->nz + (long)(sk->need0 + tr) * sk->words;
      if (!((rn[tc >> 6] >> (tc & 63)) & 1)) return false;
    }
    const sz_word* ra = sk->nz + (long)(sk->tr0 + tr) * sk->words;
    const sz_word* rb = sk->nz + (long)(sk->tc0 + tc) * sk->words;
    int kmin = -1, kmax = -1;
    for (int q = sk->kt0 >> 6; q <= (sk->kt1 - 1) >> 6; ++q) {
      sz_word m = ra[q] & rb[q];
      if (q == (sk->kt0 >> 6)) m &= ~(sz_word)0 << (sk->kt0 & 63);
      if (q == ((sk->kt1 - 1) >> 6) && (sk->kt1 & 63)) m &= ~(~(sz_word)0 << (sk->kt1 & 63));
      if (m != 0) {
        if (kmin < 0) kmin = q * 64 + (int)__builtin_ctzll(m);
        kmax = q * 64 + 63 - (int)__builtin_clzll(m);
      }
    }
    if (kmin < 0) {
      if (sk->need0 >= 0)
        K = 0;   // a needed result tile without a live product (cannot happen for an inverse; kept exact): zeros are stored
      else if (!(keep_first && tr == 0 && tc == 0))
        return false;
    } else {
      A += (long)(kmin - sk->kt0) * TILE * lda;
      B += (long)(kmin - sk->kt0) * TILE * ldb;
      K = (long)(kmax - kmin + 1) * TILE;
    }
  }
  if (STAMP) stA = (long long)__builtin_amdgcn_s_memtime();   // the tile is known
  // stage s: A chunk at smem + s*2*KB*LDS_LD, B chunk right after it
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int w = t >> 6;
  const int wr = w >> 2, wc = w & 3;
  const int l15 = lane & 15, lq = lane >> 4, l3 = lane & 3;
  const double* Ag = A + tr * TILE;
  const double* Bg = B + tc * TILE;
  // acc[j][i] = C[row = r0 + 16 i + l15][col = c0 + 4 j + lq], seeded with (beta / alpha) * C:
  // C_new = alpha * (A B' + (beta / alpha) C)
  double* Cg = C + (tr * TILE + wr * 64 + l15) + (tc * TILE + wc * WCOLS + lq) * ldc;
  // the seed may come from a different matrix than the one written (C_out = alpha A B' + beta C_in)
  const double* Cs = Cin + (tr * TILE + wr * 64 + l15) + (tc * TILE + wc * WCOLS + lq) * ldcin;
  double acc[NJ][4];
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  const int wu = __builtin_amdgcn_readfirstlane(w);
  const unsigned lds_base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) double*)smem;
  auto dma = [&](long k0, int stage) {
    double* sa = smem + stage * (2 * KB * LDS_LD);
    double* sb = sa + KB * LDS_LD;
#pragma unroll
    for (int i = 0; i < KB / 8; ++i) {  // wave w moves columns w and w + 8 of both operands
      const int col = wu + 8 * i;
      __builtin_amdgcn_global_load_lds((gptr_t)(Ag + 2 * lane + (k0 + col) * lda), (lptr_t)(sa + col * LDS_LD), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)(Bg + 2 * lane + (k0 + col) * ldb), (lptr_t)(sb + col * LDS_LD), 16, 0, 0);
    }
  };
  long nchunks = K / KB;
  // klo == 1: both operands are upper-triangular-by-tile (rows of inv(L)'), so tile row tr (>= tc)
  // contracts only columns k >= tr * 128  (C^-1 = inv(L)' inv(L) at a third of the dense flops).
  // klo == 2: A is lower-triangular-by-tile (the Cholesky factor in L Z): tile row tr stops at
  // k < (tr + 1) * 128.
  const long cbeg = klo == 1 ? tr * (TILE / KB) : 0;
  if (klo == 2 && (tr + 1) * (TILE / KB) < nchunks) nchunks = (tr + 1) * (TILE / KB);
  // prologue: the first operand chunk and the old C tile are requested together, so their
  // latencies overlap (one wait for both)
  auto kof = [&](long c) { return c * KB; };
  if (cbeg < nchunks) dma(kof(cbeg), 0);
  if (beta != 0.0) {
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[j][i] = Cs[i * 16 + (long)(j * 4) * ldcin];
    const double seed = beta / alpha;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[j][i] *= seed;
  } else {
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[j][i] = 0.0;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (STAMP) st1 = (long long)__builtin_amdgcn_s_memtime();
  for (long c = cbeg; c < nchunks; ++c) {
    const int stage = (int)(c & 1);
    // the other stage was last read in iteration c-1, which every wave left through the barrier
    if (c + 1 < nchunks) dma(kof(c + 1), stage ^ 1);
    const unsigned a_addr = lds_base + (unsigned)((stage * (2 * KB * LDS_LD) + lq * LDS_LD + wr * 64 + l15) * 8);
    const unsigned b_addr = lds_base + (unsigned)((stage * (2 * KB * LDS_LD) + KB * LDS_LD + lq * LDS_LD + wc * WCOLS + l3) * 8);
    {
      double a_r[4], b_c[NJ];
      asm volatile("ds_read_b64 %0, %1 offset:0" : "=v"(a_r[0]) : "v"(a_addr));
      asm volatile("ds_read_b64 %0, %1 offset:128" : "=v"(a_r[1]) : "v"(a_addr));
      asm volatile("ds_read_b64 %0, %1 offset:256" : "=v"(a_r[2]) : "v"(a_addr));
      asm volatile("ds_read_b64 %0, %1 offset:384" : "=v"(a_r[3]) : "v"(a_addr));
      asm volatile("ds_read_b64 %0, %1 offset:0" : "=v"(b_c[0]) : "v"(b_addr));
      asm volatile("ds_read_b64 %0, %1 offset:32" : "=v"(b_c[1]) : "v"(b_addr));
      asm volatile("ds_read_b64 %0, %1 offset:64" : "=v"(b_c[2]) : "v"(b_addr));
      asm volatile("ds_read_b64 %0, %1 offset:96" : "=v"(b_c[3]) : "v"(b_addr));
      asm volatile("ds_read_b64 %0, %1 offset:128" : "=v"(b_c[4]) : "v"(b_addr));
      asm volatile("ds_read_b64 %0, %1 offset:160" : "=v"(b_c[5]) : "v"(b_addr));
      asm volatile("ds_read_b64 %0, %1 offset:192" : "=v"(b_c[6]) : "v"(b_addr));
      asm volatile("ds_read_b64 %0, %1 offset:224" : "=v"(b_c[7]) : "v"(b_addr));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);  // keep the MFMAs below the wait (asm is opaque to hipcc)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[j][i] = mfma44_f64(b_c[j], a_r[i], acc[j][i]);
    }
    {
      double a_r[4], b_c[NJ];
      asm volatile("ds_read_b64 %0, %1 offset:4608" : "=v"(a_r[0]) : "v"(a_addr));
      asm volatile("ds_read_b64 %0, %1 offset:4736" : "=v"(a_r[1]) : "v"(a_addr));
      asm volatile("ds_read_b64 %0, %1 offset:4864" : "=v"(a_r[2]) : "v"(a_addr));
      asm volatile("ds_read_b64 %0, %1 offset:4992" : "=v"(a_r[3]) : "v"(a_addr));
      asm volatile("ds_read_b64 %0, %1 offset:4608" : "=v"(b_c[0]) : "v"(b_addr));
      asm volatile("ds_read_b64 %0, %1 offset:4640" : "=v"(b_c[1]) : "v"(b_addr));
      asm volatile("ds_read_b64 %0, %1 offset:4672" : "=v"(b_c[2]) : "v"(b_addr));
      asm volatile("ds_read_b64 %0, %1 offset:4704" : "=v"(b_c[3]) : "v"(b_addr));
      asm volatile("ds_read_b64 %0, %1 offset:4736" : "=v"(b_c[4]) : "v"(b_addr));
      asm volatile("ds_read_b64 %0, %1 offset:4768" : "=v"(b_c[5]) : "v"(b_addr));
      asm volatile("ds_read_b64 %0, %1 offset:4800" : "=v"(b_c[6]) : "v"(b_addr));
      asm volatile("ds_read_b64 %0, %1 offset:4832" : "=v"(b_c[7]) : "v"(b_addr));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);  // keep the MFMAs below the wait (asm is opaque to hipcc)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[j][i] = mfma44_f64(b_c[j], a_r[i], acc[j][i]);
    }
    {
      double a_r[4], b_c[NJ];
      asm volatile("ds_read_b64 %0, %1 offset:9216" : "=v"(a_r[0]) : "v"(a_addr));
      asm volatile("ds_read_b64 %0, %1 offset:9344" : "=v"(a_r[1]) : "v"(a_addr));
      asm volatile("ds_read_b64 %0, %1 offset:9472" : "=v"(a_r[2]) : "v"(a_addr));
      asm volatile("ds_read_b64 %0, %1 offset:9600" : "=v"(a_r[3]) : "v"(a_addr));
      asm volatile("ds_read_b64 %0, %1 offset:9216" : "=v"(b_c[0]) : "v"(b_addr));
      asm volatile("ds_read_b64 %0, %1 offset:9248" : "=v"(b_c[1]) : "v"(b_addr));
      asm volatile("ds_read_b64 %0, %1 offset:9280" : "=v"(b_c[2]) : "v"(b_addr));
      asm volatile("ds_read_b64 %0, %1 offset:9312" : "=v"(b_c[3]) : "v"(b_addr));
      asm volatile("ds_read_b64 %0, %1 offset:9344" : "=v"(b_c[4]) : "v"(b_addr));
      asm volatile("ds_read_b64 %0, %1 offset:9376" : "=v"(b_c[5]) : "v"(b_addr));
      asm volatile("ds_read_b64 %0, %1 offset:9408" : "=v"(b_c[6]) : "v"(b_addr));
      asm volatile("ds_read_b64 %0, %1 offset:9440" : "=v"(b_c[7]) : "v"(b_addr));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);  // keep the MFMAs below the wait (asm is opaque to hipcc)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[j][i] = mfma44_f64(b_c[j], a_r[i], acc[j][i]);
    }
    {
      double a_r[4], b_c[NJ];
      asm volatile("ds_read_b64 %0, %1 offset:13824" : "=v"(a_r[0]) : "v"(a_addr));
      asm volatile("ds_read_b64 %0, %1 offset:13952" : "=v"(a_r[1]) : "v"(a_addr));
      asm volatile("ds_read_b64 %0, %1 offset:14080" : "=v"(a_r[2]) : "v"(a_addr));
      asm volatile("ds_read_b64 %0, %1 offset:14208" : "=v"(a_r[3]) : "v"(a_addr));
      asm volatile("ds_read_b64 %0, %1 offset:13824" : "=v"(b_c[0]) : "v"(b_addr));
      asm volatile("ds_read_b64 %0, %1 offset:13856" : "=v"(b_c[1]) : "v"(b_addr));
      asm volatile("ds_read_b64 %0, %1 offset:13888" : "=v"(b_c[2]) : "v"(b_addr));
      asm volatile("ds_read_b64 %0, %1 offset:13920" : "=v"(b_c[3]) : "v"(b_addr));
      asm volatile("ds_read_b64 %0, %1 offset:13952" : "=v"(b_c[4]) : "v"(b_addr));
      asm volatile("ds_read_b64 %0, %1 offset:13984" : "=v"(b_c[5]) : "v"(b_addr));
      asm volatile("ds_read_b64 %0, %1 offset:14016" : "=v"(b_c[6]) : "v"(b_addr));
      asm volatile("ds_read_b64 %0, %1 offset:14048" : "=v"(b_c[7]) : "v"(b_addr));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);  // keep the MFMAs below the wait (asm is opaque to hipcc)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[j][i] = mfma44_f64(b_c[j], a_r[i], acc[j][i]);
    }
    // chunk c+1 has landed for every wave, and every wave is done reading this stage
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  if (STAMP) st2 = (long long)__builtin_amdgcn_s_memtime();
  if (HANDOFF && tr == 0 && tc == 0) {
    // the loop's closing barrier: every wave is done reading the operand stages this overwrites.
    // acc[j][i] is C[row = 64 wr + 16 i + l15][col = 32 wc + 4 j + lq]: block row 4 wr + i, block column
    // 2 wc + (j >> 2), element [m = l15][k = 4 (j & 3) + lq] at [k * 16 + m] -- 64 consecutive doubles per store
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int rb = 4 * wr + i, cb = 2 * wc + (j >> 2);
        const int k = 4 * (j & 3) + lq;
        if (rb >= cb) smem[boff(rb, cb) + k * 16 + l15] = (rb > cb || l15 >= k) ? alpha * acc[j][i] : 0.0;
      }
    return true;
  }
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) Cg[i * 16 + (long)(j * 4) * ldc] = alpha * acc[j][i];
  if (STAMP) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the stores have left: what the slot waits for before it is reused
    if (threadIdx.x == 0) {
      long long* d = dbg + (long)blockIdx.x * 8;
      d[0] = st0;
      d[1] = st1;
      d[2] = st2;
      d[3] = (long long)__builtin_amdgcn_s_memtime();
      d[4] = (long long)((__builtin_amdgcn_s_getreg(6164) & 15) << 16 | (__builtin_amdgcn_s_getreg(63492) & 0xffff));   // XCC_ID, HW_ID
      d[5] = tr;
      d[6] = tc;
      d[7] = stA;
    }
  }
  return true;
}

// bench only: the production tile program with phase stamps (own symbol: the production kernels' code is untouched)
__global__ __launch_bounds__(512, 4) void gemm_nt_dma_stamp_kernel(const double* A, long lda, double* C, long ldc, long K,
                                                                   long n_tr, long n_tc, long long* dbg) {
  __shared__ __attribute__((aligned(16))) double smem[2 * 2 * KB * LDS_LD];
  long tr, tc;
  gemm_nt_dma_tile<false, true>(A, lda, A, lda, C, ldc, K, -1.0, 1.0, 0L, n_tr, n_tc, 0L, C, ldc, 0, smem, tr, tc, dbg);
}

int launch_gemm_nt_stamps(const double* P, long ldp, double* C, long ldc, long M, long Nc, long K, long long* dbg,
                          long* n_ids, hipStream_t s) {
  long n_tr = M / TILE, n_tc = Nc / TILE;
  long per_xcd = tri_ids_per_xcd(tri_shape(n_tr, n_tc, -1));
  *n_ids = per_xcd * 8;
  if (!dbg) return 0;
  hipLaunchKernelGGL(gemm_nt_dma_stamp_kernel, dim3((unsigned)(per_xcd * 8)), dim3(512), 0, s, P, ldp, C, ldc, K, n_tr, n_tc, dbg);
  SGP_HIP(hipGetLastError());
  return 0;
}

template <int TAG>
__global__ __launch_bounds__(512, 4) void gemm_nt_dma_kernel(const double* A, long lda,
                                                             const double* B, long ldb, double* C,
                                                             long ldc, long K, double alpha,
                                                             double beta, long mask_off, long n_tr,
                                                             long n_tc, long c_slice_stride,
                                                             const double* Cin, long ldcin, int klo, TileSkip sk) {
  __shared__ __attribute__((aligned(16))) double smem[2 * 2 * KB * LDS_LD];
  long tr, tc;
  gemm_nt_dma_tile<false>(A, lda, B, ldb, C, ldc, K, alpha, beta, mask_off, n_tr, n_tc, c_slice_stride, Cin, ldcin,
                          klo, smem, tr, tc, nullptr, &sk);
}

// Fused lower update + Cholesky of the NEXT diagonal block (SGP_FUSE_POTRF, capi.hip: panel_factor /
// chol_bordered): tile (0, 0) of every trailing update C[lower] -= P P' is the diagonal block the factorisation
// needs next.  Its workgroup -- the first one dispatched -- goes straight from the store of the updated tile
// into potrf_diag_body on it, while the other workgroups of the launch are still updating their tiles: the
// 128-pivot chain of block column j + 1 runs under the update with block column j instead of after it, with no
// extra launch, stream or event (the cross-stream version of this overlap, SGP_INNER_LA, lost more on its two
// event hand-offs per block than it hid).  Dynamic LDS: PD_LDS bytes (>= the tile program's 73 728), so two
// workgroups still fit a CU.
// HANDOFF = false: the tile goes through global memory (stored by the tile program, re-read by potrf_diag_body's
// own load phase); true: straight from the accumulators into potrf_diag_body's LDS layout (no store, no fence, no
// reload: ~4 us less on the serial chain).
template <int TAG, bool HANDOFF>
__global__ __launch_bounds__(512, 4) void gemm_nt_dma_potrf_kernel(const double* A, long lda, const double* B,
                                                                   long ldb, double* C, long ldc, long K,
                                                                   long n_tr, long n_tc, double* invd,
                                                                   double* logdet_slot, int* info, long gcol0,
                                                                   int prio, TileSkip sk) {
  extern __shared__ __attribute__((aligned(16))) double dyn_smem[];
  long tr, tc;
  const bool live = gemm_nt_dma_tile<HANDOFF>(A, lda, B, ldb, C, ldc, K, -1.0, 1.0, 0L, n_tr, n_tc, 0L, C, ldc, 0,
                                              dyn_smem, tr, tc, nullptr, &sk, true);
  if (live && tr == 0 && tc == 0) {   // workgroup-uniform
    if (!HANDOFF) {
      // the tile was written by all eight waves: stores complete + visible to the workgroup before it is re-read
      __threadfence_block();
    }
    __syncthreads();
    potrf_diag_body<false, double, HANDOFF>(C, ldc, invd, logdet_slot, info, gcol0, prio, nullptr);
  }
}

// ---------------------------------------------------------------------------------------
// Batched, segmented trailing update (round 4; the multi-GPU drivers: csrc/multi.hip, stheno.jl_amd/dist.py).
// A rank of the sharded factorisation owns several PACKED column panels (each its own matrix with its own leading
// dimension) and receives the factored panels one by one, each again a matrix of its own.  One launch per (source panel,
// destination panel) pair made 8 small launches per rank and step at K = one panel width -- 62 TFLOP/s against the 67
// the single-GPU schedule reaches with its K = 4096 launches, and a launch rate one host thread had to keep up for eight
// GPUs.  This kernel takes a LIST of destination panels, each with a RANGE of source panels, in ONE launch:
//   C_d[lower trapezoid] -= sum_{s in range(d)} P_s[rows of C_d] P_s[rows of C_d's diagonal block]'
// contracted source after source in ascending order with the accumulators kept in registers (a tile sees k ascending
// over the whole range: bit-identical to applying the panels one launch at a time) and the LDS-DMA pipeline running
// across the segment boundaries (the next chunk's address simply comes from the next source).  So the owner of far-away
// panels can apply G received panels at once (K = G x 1024: the C tiles travel once instead of G times, the per-tile
// prologue is paid once) and all destinations of a step share a launch.
// The tile program is gemm_nt_dma_tile's: 128 x 128 tile, 8 waves as 2 x 4, chunks of 16 through two LDS stages.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(512, 4) void gemm_nt_seg_kernel(SegBatch b) {
  __shared__ __attribute__((aligned(16))) double smem[2 * 2 * KB * LDS_LD];
  // destination of this workgroup: ids are dealt destination after destination (id0 ascending, multiples of 8 so that
  // id % 8 -- the XCD -- keeps its meaning inside every destination)
  unsigned id = blockIdx.x;
  if (b.cmap) {
    // structured launches (round 6): the live tiles of every XCD compacted to the front of its id sequence
    // (seg_compact_kernel) -- the dead ids leave at the END of the grid, and the live tiles keep the order, and with it the
    // lock-step operand sharing, of the dense enumeration
    const int x = (int)(id & 7), pos = (int)(id >> 3);
    if (pos >= b.cmap[x]) return;
    id = (unsigned)b.cmap[16 + (long)x * b.cstride + pos] * 8u + (unsigned)x;
  }
  int d = 0;
  for (int q = 1; q < b.n_dst; ++q)
    if (id >= b.dst[q].id0) d = q;
  d = __builtin_amdgcn_readfirstlane(d);
  const long c0 = b.dst[d].c0;
  const long n_tr = (b.m_tot - c0) / TILE, n_tc = b.dst[d].w / TILE;
  long tr, tc;
  if (!tile_of_id((long)(id - b.dst[d].id0), n_tr, n_tc, 0L, tr, tc)) return;
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int w = t >> 6;
  const int wu = __builtin_amdgcn_readfirstlane(w);
  const int wr = w >> 2, wc = w & 3;
  const int l15 = lane & 15, lq = lane >> 4, l3 = lane & 3;
  const long ldc = b.dst[d].ldc;
  double* Cg = b.dst[d].C + (tr * TILE + wr * 64 + l15) + (tc * TILE + wc * 32 + lq) * ldc;
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  const unsigned lds_base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) double*)smem;
  // source cursor: the NEXT chunk to request
  int s = b.dst[d].s_first;
  const int s_end = s + b.dst[d].s_count;
  // structural zeros: a source all of whose k tiles have a structurally zero operand tile for this tile is skipped (bit q
  // of `livemask`); a tile without a live source is left alone
  // Round 5: and inside a live source the tile contracts the k tiles from its first to its last needed one only (a source
  // panel that straddles a block boundary of the programme) -- the single-GPU kernel's trim (gemm_nt_dma_tile); the tiles left
  // out are exact zeros, the bits stay.  range_of(q): first needed k tile and number of k tiles of source q relative to the
  // source's first column (sources whose columns start and end on tile boundaries; else the whole source), -1 = dead.
  const sz_word* ra = b.nz ? b.nz + (c0 / TILE + tr) * b.nz_words : nullptr;
  const sz_word* rb = b.nz ? b.nz + (c0 / TILE + tc) * b.nz_words : nullptr;
  auto range_of = [&](int q, int& first, int& count) {
    first = 0;
    count = b.src[q].w / KB;   // in chunks of KB columns
    if (!ra) return;
    const long kc = b.src[q].k0 >= 0 ? b.src[q].k0 : b.src[q].row0;
    const int kt0 = (int)(kc / TILE), kt1 = (int)((kc + b.src[q].w + TILE - 1) / TILE);
    int kmin = -1, kmax = -1;
    for (int k = kt0; k < kt1; ++k)
      if ((((ra[k >> 6] & rb[k >> 6]) >> (k & 63)) & 1) != 0) {
        if (kmin < 0) kmin = k;
        kmax = k;
      }
    if (kmin < 0) {
      count = -1;
    } else if (kc % TILE == 0 && b.src[q].w % TILE == 0) {
      first = (kmin - kt0) * (TILE / KB);
      count = (kmax - kmin + 1) * (TILE / KB);
    }
  };
  unsigned livemask = ~0u;
  long total = 0;
  {
    livemask = 0;
    for (int q = s; q < s_end; ++q) {
      int f, cnt;
      range_of(q, f, cnt);
      if (cnt > 0) {
        livemask |= 1u << q;
        total += cnt;
      }
    }
    if (livemask == 0) return;
    while (!((livemask >> s) & 1)) ++s;
  }
  const double *pa = nullptr, *pb = nullptr;
  long ld = 0;
  int left = 0;
  auto open = [&](int q) {
    const long r0 = c0 - b.src[q].row0;   // first stored row of the source is its own column offset
    ld = b.src[q].ld;
    int f, cnt;
    range_of(q, f, cnt);
    const long skip = (long)f * KB * ld;   // the columns before the first needed k tile
    pa = b.src[q].base + r0 + tr * TILE + skip;
    pb = b.src[q].base + r0 + tc * TILE + skip;
    left = cnt;
  };
  auto request = [&](int stage) {
    double* sa = smem + stage * (2 * KB * LDS_LD);
    double* sb = sa + KB * LDS_LD;
#pragma unroll
    for (int i = 0; i < KB / 8; ++i) {  // wave w moves columns w and w + 8 of both operands
      const int col = wu + 8 * i;
      __builtin_amdgcn_global_load_lds((gptr_t)(pa + 2 * lane + col * ld), (lptr_t)(sa + col * LDS_LD), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)(pb + 2 * lane + col * ld), (lptr_t)(sb + col * LDS_LD), 16, 0, 0);
    }
    pa += KB * ld;
    pb += KB * ld;
    if (--left == 0) {
      ++s;
      while (s < s_end && !((livemask >> s) & 1)) ++s;
      if (s < s_end) open(s);
    }
  };
  double acc[8][4];
  if (total > 0) {
    open(s);
    request(0);
  }
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[j][i] = Cg[i * 16 + (long)(j * 4) * ldc];
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[j][i] *= -1.0;   // C_new = -(-C + P P')
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  const unsigned a_off = (unsigned)((lq * LDS_LD + wr * 64 + l15) * 8);
  const unsigned b_off = (unsigned)((KB * LDS_LD + lq * LDS_LD + wc * 32 + l3) * 8);
  for (long c = 0; c < total; ++c) {
    const int stage = (int)(c & 1);
    if (c + 1 < total) request(stage ^ 1);
    const unsigned a_addr = lds_base + (unsigned)(stage * (2 * KB * LDS_LD) * 8) + a_off;
    const unsigned b_addr = lds_base + (unsigned)(stage * (2 * KB * LDS_LD) * 8) + b_off;
    tile_kstep<0>(acc, a_addr, b_addr);
    tile_kstep<1>(acc, a_addr, b_addr);
    tile_kstep<2>(acc, a_addr, b_addr);
    tile_kstep<3>(acc, a_addr, b_addr);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) Cg[i * 16 + (long)(j * 4) * ldc] = -acc[j][i];
}

// One workgroup per XCD walks that XCD's ids of a segmented update in order and writes the live ones (some source has a k tile
// with both operand tiles structurally non-zero) to the front: out[x] = live ids of XCD x, out[16 + x * per_xcd + pos] = the
// pos-th live id / 8 (the single-GPU launches' tile_compact_kernel, for a list of destinations).
__global__ __launch_bounds__(1024) void seg_compact_kernel(SegBatch b, int per_xcd, int* out) {
  __shared__ int s_wave[16];
  __shared__ int s_base;
  const int x = blockIdx.x, t = threadIdx.x, lane = t & 63, w = t >> 6;
  if (t == 0) s_base = 0;
  __syncthreads();
  for (int k0 = 0; k0 < per_xcd; k0 += 1024) {
    const int k = k0 + t;
    bool live = false;
    if (k < per_xcd) {
      const unsigned id = (unsigned)k * 8u + (unsigned)x;
      int d = 0;
      for (int q = 1; q < b.n_dst; ++q)
        if (id >= b.dst[q].id0) d = q;
      const long c0 = b.dst[d].c0;
      long tr, tc;
      if (tile_of_id((long)(id - b.dst[d].id0), (b.m_tot - c0) / TILE, b.dst[d].w / TILE, 0L, tr, tc)) {
        const sz_word* ra = b.nz + (c0 / TILE + tr) * b.nz_words;
        const sz_word* rb = b.nz + (c0 / TILE + tc) * b.nz_words;
        for (int q = b.dst[d].s_first; q < b.dst[d].s_first + b.dst[d].s_count && !live; ++q) {
          const long kc = b.src[q].k0 >= 0 ? b.src[q].k0 : b.src[q].row0;
          const int kt0 = (int)(kc / TILE), kt1 = (int)((kc + b.src[q].w + TILE - 1) / TILE);
          for (int kk = kt0; kk < kt1 && !live; ++kk) live = (((ra[kk >> 6] & rb[kk >> 6]) >> (kk & 63)) & 1) != 0;
        }
      }
    }
    const unsigned long long bal = __ballot(live);
    const int pre = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) s_wave[w] = __popcll(bal);
    __syncthreads();
    int woff = 0, tot = 0;
    for (int q = 0; q < 16; ++q) {
      if (q < w) woff += s_wave[q];
      tot += s_wave[q];
    }
    const int base = s_base;
    if (live) out[16 + (long)x * per_xcd + base + woff + pre] = k;
    __syncthreads();
    if (t == 0) s_base = base + tot;
    __syncthreads();
  }
  if (t == 0) out[x] = s_base;
}

// dsts[i].id0 is filled in here.  Every width a multiple of 128, every c0 a multiple of 128 and >= the row0 of its
// sources.  Returns the number of workgroups launched through *n_ids (may be NULL).
int launch_gemm_nt_seg(SegBatch& b, hipStream_t s, long* n_ids) {
  if (b.n_dst <= 0) return 0;
  if (b.n_dst > SEG_MAX_DST || b.m_tot % TILE) {
    set_error("gemm_nt_seg: bad batch");
    return -1;
  }
  long ids = 0;
  for (int d = 0; d < b.n_dst; ++d) {
    const SegDst& D = b.dst[d];
    if (D.c0 % TILE || D.w % TILE || D.w <= 0 || D.c0 + D.w > b.m_tot || D.s_first < 0 || D.s_count < 0 ||
        D.s_first + D.s_count > SEG_MAX_SRC) {
      set_error("gemm_nt_seg: bad destination panel");
      return -1;
    }
    for (int q = D.s_first; q < D.s_first + D.s_count; ++q)
      if (b.src[q].w % KB || b.src[q].w <= 0 || b.src[q].row0 > D.c0) {
        set_error("gemm_nt_seg: bad source panel");
        return -1;
      }
    b.dst[d].id0 = (unsigned)ids;
    ids += tile_ids((b.m_tot - D.c0) / TILE, D.w / TILE, 0);
  }
  if (ids >= (1L << 31)) {
    set_error("gemm_nt_seg: too many tiles for one launch");
    return -1;
  }
  if (n_ids) *n_ids = ids;
  // big structured launches (the caller lends a scratch map that only launches of THIS stream use): compact the live ids first
  const int* map = nullptr;
  if (b.nz && b.map_scratch && ids >= b.map_min_ids && 16 + ids <= b.map_ints) {
    b.cmap = nullptr;
    hipLaunchKernelGGL(seg_compact_kernel, dim3(8), dim3(1024), 0, s, b, (int)(ids / 8), b.map_scratch);
    SGP_HIP(hipGetLastError());
    map = b.map_scratch;
  }
  b.cmap = map;
  b.cstride = (int)(ids / 8);
  hipLaunchKernelGGL(gemm_nt_seg_kernel, dim3((unsigned)ids), dim3(512), 0, s, b);
  SGP_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------------------
// Register-staged kernel (global_load_dwordx4 -> VGPR -> ds_write_b128, __syncthreads).  Kept for
// the K-capped product (rand: only the lower triangle of L is valid, so tile column tc stops
// at k = (tc + 1) * 128) and as the A/B baseline of the bench (50 vs 57 TF/s).
// ---------------------------------------------------------------------------------------
// Shared inner product of one LDS stage: NJ column fragments (4 cols each) x 4 row fragments
// (16 rows each) per wave, K = 16 in four k-steps of 4.
#define SGP_COMPUTE(buf_, NJ_, COL0_)                                            \
  {                                                                              \
    const double* pa = &sA[buf_][wr * 64 + l15];                                 \
    const double* pb = &sB[buf_][(COL0_) + l3];                                  \
    _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                           \
      const int kk = ks * 4 + lq;                                                \
      double a_r[4], b_c[NJ_];                                                   \
      _Pragma("unroll") for (int i = 0; i < 4; ++i) a_r[i] = pa[kk * LDS_LD + i * 16]; \
      _Pragma("unroll") for (int j = 0; j < NJ_; ++j) b_c[j] = pb[kk * LDS_LD + j * 4]; \
      _Pragma("unroll") for (int j = 0; j < NJ_; ++j)                            \
          _Pragma("unroll") for (int i = 0; i < 4; ++i)                          \
              acc[j][i] = mfma44_f64(b_c[j], a_r[i], acc[j][i]);                 \
    }                                                                            \
  }

template <bool KCAP>
__global__ __launch_bounds__(512, 4) void gemm_nt_reg_kernel(const double* A, long lda,
                                                             const double* B, long ldb, double* C,
                                                             long ldc, long K, double alpha,
                                                             double beta, long mask_off,
                                                             long kcap_off, long n_tr, long n_tc) {
  constexpr int NJ = 8, WCOLS = 32, NU = 2;
  long tr, tc;
  if (!tile_of_block(n_tr, n_tc, mask_off, tr, tc)) return;
  long Keff = K;
  if (KCAP) {
    long cap = (tc + 1) * TILE + kcap_off;
    if (cap < Keff) Keff = cap;
    if (Keff <= 0) Keff = 0;
  }
  __shared__ __attribute__((aligned(16))) double sA[2][KB * LDS_LD];
  __shared__ __attribute__((aligned(16))) double sB[2][KB * LDS_LD];
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int w = t >> 6;
  const int wr = w >> 2, wc = w & 3;
  const int l15 = lane & 15, lq = lane >> 4, l3 = lane & 3;
  const double* Ag = A + tr * TILE;
  const double* Bg = B + tc * TILE;
  const int scol = t >> 6;          // staging: unit i of thread t -> column scol + 8 i, rows srow, srow + 1
  const int srow = 2 * (t & 63);
  double2 ra[NU], rb[NU];
  double* Cg = C + (tr * TILE + wr * 64 + l15) + (tc * TILE + wc * WCOLS + lq) * ldc;
  double acc[NJ][4];
  if (beta != 0.0) {
    const double seed = beta / alpha;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[j][i] = Cg[i * 16 + (long)(j * 4) * ldc];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[j][i] *= seed;
  } else {
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[j][i] = 0.0;
  }
#define SGP_GLOAD(k0_)                                                           \
  _Pragma("unroll") for (int i = 0; i < NU; ++i) {                               \
    long col = (k0_) + scol + 8 * i;                                             \
    ra[i] = *reinterpret_cast<const double2*>(Ag + srow + col * lda);            \
    rb[i] = *reinterpret_cast<const double2*>(Bg + srow + col * ldb);            \
  }
#define SGP_SSTORE(buf_)                                                         \
  _Pragma("unroll") for (int i = 0; i < NU; ++i) {                               \
    int col = scol + 8 * i;                                                      \
    *reinterpret_cast<double2*>(&sA[buf_][col * LDS_LD + srow]) = ra[i];         \
    *reinterpret_cast<double2*>(&sB[buf_][col * LDS_LD + srow]) = rb[i];         \
  }
  if (Keff > 0) {
    SGP_GLOAD(0);
    SGP_SSTORE(0);
    __syncthreads();
    int buf = 0;
    for (long k0 = KB; k0 < Keff; k0 += KB) {
      SGP_GLOAD(k0);
      SGP_COMPUTE(buf, NJ, wc * WCOLS);
      SGP_SSTORE(buf ^ 1);
      __syncthreads();
      buf ^= 1;
    }
    SGP_COMPUTE(buf, NJ, wc * WCOLS);
  }
#undef SGP_GLOAD
#undef SGP_SSTORE
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) Cg[i * 16 + (long)(j * 4) * ldc] = alpha * acc[j][i];
}

// Structure of the matrix being factored (structural zeros, common.h): set by chol_bordered for its scope.
namespace {
struct GemmStructure {
  const char* base = nullptr;   // (bytes: the fp32 instantiation registers a float matrix, f32.hip)
  long elem = 8;                // bytes per element of the registered matrix
  long ld = 0;
  const sz_word* nz = nullptr;
  int words = 0;
  int* scratch = nullptr;
  long scratch_ints = 0;
  long tile0 = 0;   // global tile coordinate of base[0] (a packed panel: its first column / 128)
  long ncols = 0;   // columns of the registered matrix: a pointer beyond base + ld * ncols is NOT part of it (advisor, round 5: a
                    // scratch buffer with the same leading dimension allocated above the matrix must not inherit its pattern)
  // a second map for the launches of ONE other stream (the look-ahead schedules issue the far updates there: launches of a
  // stream are ordered, so one map per stream is enough)
  hipStream_t stream2 = nullptr;
  int* scratch2 = nullptr;
};
thread_local GemmStructure g_st;
// the skip record of the diagonal-aligned lower update C[lower] -= P P' (P's rows = C's rows = C's columns), or an empty one
TileSkip skip_for_bytes(const void* P, long ldp, const void* C, long ldc, long K, long elem) {
  TileSkip sk;
  if (!g_st.nz || ldp != g_st.ld || ldc != g_st.ld || elem != g_st.elem) return sk;
  const long oc = ((const char*)C - g_st.base) / elem, op = ((const char*)P - g_st.base) / elem;
  if ((const char*)C < g_st.base || (const char*)P < g_st.base || oc >= g_st.ld * g_st.ncols || op >= g_st.ld * g_st.ncols) return sk;
  const long cr = oc % g_st.ld, cc = oc / g_st.ld, pr = op % g_st.ld, pc = op / g_st.ld;
  if (cr != cc || pr != cr || cr % TILE || pc % TILE || K % TILE || pc + K > cc) return sk;
  sk.nz = g_st.nz;
  sk.words = g_st.words;
  sk.tr0 = sk.tc0 = (int)(cr / TILE + g_st.tile0);
  sk.kt0 = (int)(pc / TILE + g_st.tile0);
  sk.kt1 = sk.kt0 + (int)(K / TILE);
  return sk;
}
TileSkip skip_for(const double* P, long ldp, const double* C, long ldc, long K) { return skip_for_bytes(P, ldp, C, ldc, K, 8); }
StripSkip strip_skip_bytes(const void* X, long ldx, long elem) {
  StripSkip sk;
  if (!g_st.nz || ldx != g_st.ld || elem != g_st.elem || (const char*)X < g_st.base) return sk;
  const long ox = ((const char*)X - g_st.base) / elem;
  if (ox >= g_st.ld * g_st.ncols) return sk;
  const long xr = ox % g_st.ld, xc = ox / g_st.ld;
  if (xr % TILE || xc % TILE || xr <= xc) return sk;   // rows below the diagonal block of a block column
  sk.nz = g_st.nz;
  sk.words = g_st.words;
  sk.tr0 = (int)(xr / TILE + g_st.tile0);
  sk.kt = (int)(xc / TILE + g_st.tile0);
  return sk;
}
}  // namespace
void gemm_set_structure(const double* base, long ld, const sz_word* d_nz, int words, long ncols, int* scratch,
                        long scratch_ints, long tile0, hipStream_t stream2, int* scratch2) {
  g_st.base = (const char*)base;
  g_st.elem = 8;
  g_st.ld = ld;
  g_st.nz = d_nz;
  g_st.words = words;
  g_st.scratch = scratch;
  g_st.scratch_ints = scratch_ints;
  g_st.tile0 = tile0;
  g_st.ncols = ncols;
  g_st.stream2 = stream2;
  g_st.scratch2 = scratch2;
}
// the fp32 instantiation's matrix (f32.hip: no compacted id maps -- its launches keep the dense enumeration)
void gemm_set_structure_f32(const float* base, long ld, const sz_word* d_nz, int words, long ncols) {
  gemm_set_structure(nullptr, ld, d_nz, words, ncols);
  g_st.base = (const char*)base;
  g_st.elem = 4;
}
TileSkip gemm_skip_for_f32(const float* P, long ldp, const float* C, long ldc, long K) { return skip_for_bytes(P, ldp, C, ldc, K, 4); }
StripSkip strip_skip_for_f32(const float* X, long ldx) { return strip_skip_bytes(X, ldx, 4); }
StripSkip strip_skip_for(const double* X, long ldx) { return strip_skip_bytes(X, ldx, 8); }

// One workgroup per XCD walks that XCD's ids of a lower update in order and writes the live ones (some k tile of the panel
// has both operand tiles structurally non-zero; `keep_first`: tile (0, 0), which a fused launch factors) to the front.
__global__ __launch_bounds__(1024) void tile_compact_kernel(TileSkip sk, long n_tr, long n_tc, int per_xcd, int* out,
                                                            int keep_first) {
  __shared__ int s_wave[16];
  __shared__ int s_base;
  const int x = blockIdx.x, t = threadIdx.x, lane = t & 63, w = t >> 6;
  if (t == 0) s_base = 0;
  __syncthreads();
  for (int k0 = 0; k0 < per_xcd; k0 += 1024) {
    const int k = k0 + t;
    bool live = false;
    long tr, tc;
    if (k < per_xcd && tile_of_id((long)k * 8 + x, n_tr, n_tc, 0L, tr, tc)) {
      const sz_word* ra = sk.nz + (long)(sk.tr0 + tr) * sk.words;
      const sz_word* rb = sk.nz + (long)(sk.tc0 + tc) * sk.words;
      for (int q = sk.kt0 >> 6; q <= (sk.kt1 - 1) >> 6; ++q) {
        sz_word m = ra[q] & rb[q];
        if (q == (sk.kt0 >> 6)) m &= ~(sz_word)0 << (sk.kt0 & 63);
        if (q == ((sk.kt1 - 1) >> 6) && (sk.kt1 & 63)) m &= ~(~(sz_word)0 << (sk.kt1 & 63));
        live = live || m != 0;
      }
      live = live || (keep_first && tr == 0 && tc == 0);
    }
    const unsigned long long bal = __ballot(live);
    const int pre = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) s_wave[w] = __popcll(bal);
    __syncthreads();
    int woff = 0, tot = 0;
    for (int q = 0; q < 16; ++q) {
      if (q < w) woff += s_wave[q];
      tot += s_wave[q];
    }
    const int base = s_base;
    if (live) out[16 + (long)x * per_xcd + base + woff + pre] = k;
    __syncthreads();
    if (t == 0) s_base = base + tot;
    __syncthreads();
  }
  if (t == 0) out[x] = s_base;
}
// big structured launches on an ordered stream: build the compacted id map first (the update kernel reads it)
static int maybe_compact(TileSkip& sk, long n_tr, long n_tc, long per_xcd, int keep_first, hipStream_t s) {
  int* map = (g_st.scratch2 && s == g_st.stream2) ? g_st.scratch2 : g_st.scratch;
  if (!sk.nz || !map || per_xcd * 8 < 4096 || 16 + per_xcd * 8 > g_st.scratch_ints) return 0;
  hipLaunchKernelGGL(tile_compact_kernel, dim3(8), dim3(1024), 0, s, sk, n_tr, n_tc, (int)per_xcd, map, keep_first);
  SGP_HIP(hipGetLastError());
  sk.cmap = map;
  sk.cstride = (int)per_xcd;
  return 0;
}

int launch_gemm_nt(const double* A, long lda, const double* B, long ldb, double* C, long ldc,
                   long M, long Nc, long K, double alpha, double beta, long mask_off,
                   int kcap_mode, long kcap_off, hipStream_t s) {
  constexpr long REG_BASELINE = -(1L << 50);  // kcap_mode == 0 and this kcap_off: see below
  if (M <= 0 || Nc <= 0) return 0;
  if (M % TILE || Nc % TILE || K % KB) {
    set_error("gemm_nt: M, Nc must be multiples of 128 and K of 16");
    return -1;
  }
  long n_tr = M / TILE, n_tc = Nc / TILE;
  long groups = ((n_tr + 7) / 8 + 7) / 8;  // groups of 8 owned rows per XCD
  long per_xcd = (mask_off == 0) ? tri_ids_per_xcd(tri_shape(n_tr, n_tc, -1)) : groups * 8 * n_tc;
  dim3 grid((unsigned)(per_xcd * 8));
  if (kcap_mode)
    hipLaunchKernelGGL((gemm_nt_reg_kernel<true>), grid, dim3(512), 0, s, A, lda, B, ldb, C, ldc, K, alpha,
                       beta, mask_off, kcap_off, n_tr, n_tc);
  else if (kcap_off == REG_BASELINE)  // bench A/B only (sgp_bench_gemm): the register-staged kernel
    hipLaunchKernelGGL((gemm_nt_reg_kernel<false>), grid, dim3(512), 0, s, A, lda, B, ldb, C, ldc, K, alpha,
                       beta, mask_off, 0L, n_tr, n_tc);
  else
    hipLaunchKernelGGL(gemm_nt_dma_kernel<0>, grid, dim3(512), 0, s, A, lda, B, ldb, C, ldc, K, alpha, beta,
                       mask_off, n_tr, n_tc, 0L, (const double*)C, ldc, 0,
                       (A == B && mask_off == 0 && alpha == -1.0 && beta == 1.0) ? skip_for(A, lda, C, ldc, K) : TileSkip());
  SGP_HIP(hipGetLastError());
  return 0;
}

// The outer trailing update C[lower] -= P P' (K = outer panel width): same kernel under its own symbol.
int launch_gemm_nt_update(const double* P, long ldp, double* C, long ldc, long M, long Nc, long K, hipStream_t s) {
  if (M <= 0 || Nc <= 0) return 0;
  if (M % TILE || Nc % TILE || K % KB) {
    set_error("gemm_nt_update: M, Nc must be multiples of 128 and K of 16");
    return -1;
  }
  long n_tr = M / TILE, n_tc = Nc / TILE;
  long per_xcd = tri_ids_per_xcd(tri_shape(n_tr, n_tc, -1));
  TileSkip sk = skip_for(P, ldp, C, ldc, K);
  if (int rc = maybe_compact(sk, n_tr, n_tc, per_xcd, 0, s)) return rc;
  hipLaunchKernelGGL(gemm_nt_dma_kernel<1>, dim3((unsigned)(per_xcd * 8)), dim3(512), 0, s, P, ldp, P, ldp, C, ldc, K,
                     -1.0, 1.0, 0L, n_tr, n_tc, 0L, (const double*)C, ldc, 0, sk);
  SGP_HIP(hipGetLastError());
  return 0;
}

// C[lower] -= P P' (as launch_gemm_nt_update / the inner K = 128 updates) with the Cholesky of tile (0, 0) -- the
// next diagonal block -- fused into the workgroup that updates it.  outer != 0: the <1> symbol (the launches
// bench.py times as trailing updates), else <0>.
template <int TAG, bool HANDOFF>
static int launch_potrf_variant(unsigned grid, const double* P, long ldp, double* C, long ldc, long K, long n_tr,
                                long n_tc, double* d_invd, double* d_logdet_slot, int* d_info, long gcol0,
                                hipStream_t s) {
  SGP_LDS_ATTR_ONCE((gemm_nt_dma_potrf_kernel<TAG, HANDOFF>), PD_LDS);
  TileSkip sk = skip_for(P, ldp, C, ldc, K);
  if (int rc = maybe_compact(sk, n_tr, n_tc, (long)grid / 8, 1, s)) return rc;
  hipLaunchKernelGGL((gemm_nt_dma_potrf_kernel<TAG, HANDOFF>), dim3(grid), dim3(512), PD_LDS, s, P, ldp, P, ldp, C, ldc,
                     K, n_tr, n_tc, d_invd, d_logdet_slot, d_info, gcol0, panel_prio(), sk);
  SGP_HIP(hipGetLastError());
  return 0;
}

int launch_gemm_nt_potrf(const double* P, long ldp, double* C, long ldc, long M, long Nc, long K, int outer,
                         int handoff, double* d_invd, double* d_logdet_slot, int* d_info, long gcol0,
                         hipStream_t s) {
  if (M <= 0 || Nc <= 0) return 0;
  if (M % TILE || Nc % TILE || K % KB) {
    set_error("gemm_nt_potrf: M, Nc must be multiples of 128 and K of 16");
    return -1;
  }
  long n_tr = M / TILE, n_tc = Nc / TILE;
  long per_xcd = tri_ids_per_xcd(tri_shape(n_tr, n_tc, -1));
  const unsigned grid = (unsigned)(per_xcd * 8);
  if (outer)
    return handoff ? launch_potrf_variant<1, true>(grid, P, ldp, C, ldc, K, n_tr, n_tc, d_invd, d_logdet_slot, d_info, gcol0, s)
                   : launch_potrf_variant<1, false>(grid, P, ldp, C, ldc, K, n_tr, n_tc, d_invd, d_logdet_slot, d_info, gcol0, s);
  return handoff ? launch_potrf_variant<0, true>(grid, P, ldp, C, ldc, K, n_tr, n_tc, d_invd, d_logdet_slot, d_info, gcol0, s)
                 : launch_potrf_variant<0, false>(grid, P, ldp, C, ldc, K, n_tr, n_tc, d_invd, d_logdet_slot, d_info, gcol0, s);
}

// C_out = alpha A B' + beta C_in with C_in a different matrix than C_out (full rectangle).  C_out may
// alias A when Nc == K == 128 (every workgroup then owns complete rows of A and has read them all
// before its store-only epilogue) -- the refinement step of the panel solve uses exactly that.
int launch_gemm_nt_cin(const double* A, long lda, const double* B, long ldb, const double* Cin, long ldcin,
                       double* C, long ldc, long M, long Nc, long K, double alpha, double beta,
                       hipStream_t s) {
  if (M <= 0 || Nc <= 0) return 0;
  if (M % TILE || Nc % TILE || K % KB) {
    set_error("gemm_nt_cin: M, Nc must be multiples of 128 and K of 16");
    return -1;
  }
  long n_tr = M / TILE, n_tc = Nc / TILE;
  long groups = ((n_tr + 7) / 8 + 7) / 8;
  dim3 grid((unsigned)(groups * 8 * n_tc * 8));
  hipLaunchKernelGGL(gemm_nt_dma_kernel<0>, grid, dim3(512), 0, s, A, lda, B, ldb, C, ldc, K, alpha, beta,
                     -(1L << 40), n_tr, n_tc, 0L, Cin, ldcin, 0, TileSkip());
  SGP_HIP(hipGetLastError());
  return 0;
}

// C[lower tiles] = X X' for X upper-triangular by 128-tile (n x n, X[i][k] = 0 for k < 128 floor(i/128)):
// tile (tr, tc), tr >= tc, contracts k >= tr * 128 only.
// sk (round 5, structured models): X's tile pattern (rows tr0 + t of sk->nz = tile row t of X) and the result tiles that are
// needed at all (need0); a tile then contracts its first .. last k tile with both operand tiles non-zero -- which already
// starts at k >= tr, X being upper triangular by tile IN the pattern -- and the tiles nobody reads are not computed.
int launch_gemm_nt_uut(const double* X, long ldx, double* C, long ldc, long n, hipStream_t s, const TileSkip* sk) {
  if (n <= 0) return 0;
  if (n % TILE) {
    set_error("gemm_nt_uut: n must be a multiple of 128");
    return -1;
  }
  long n_t = n / TILE;
  long per_xcd = tri_ids_per_xcd(tri_shape(n_t, n_t, -1));
  if (sk && sk->nz) {
    TileSkip k2 = *sk;
    k2.kt0 = 0;
    k2.kt1 = (int)n_t;
    k2.cmap = nullptr;
    hipLaunchKernelGGL(gemm_nt_dma_kernel<0>, dim3((unsigned)(per_xcd * 8)), dim3(512), 0, s, X, ldx, X, ldx, C, ldc, n,
                       1.0, 0.0, 0L, n_t, n_t, 0L, (const double*)C, ldc, 0, k2);
  } else
    hipLaunchKernelGGL(gemm_nt_dma_kernel<0>, dim3((unsigned)(per_xcd * 8)), dim3(512), 0, s, X, ldx, X, ldx, C, ldc, n,
                       1.0, 0.0, 0L, n_t, n_t, 0L, (const double*)C, ldc, 1, TileSkip());
  SGP_HIP(hipGetLastError());
  return 0;
}

// C (n x ns) = beta C + L Zt' for the lower-triangular-by-tile factor L (n x n, ld ldl) and Zt (ns x n,
// ld ldz): rand's m + L Z, with every tile row contracting only the columns left of its diagonal
// tile's right edge.  Tile rows are spread over all XCDs (n / 128 of them) -- the transposed
// arrangement (ns / 128 tile rows) would leave most of the chip idle for a few hundred samples.
// _k: L is an n x K column panel of the factor (K <= n, its top K x K block the triangle): the multi-GPU rand.
int launch_gemm_nt_lz_k(const double* L, long ldl, const double* Zt, long ldz, double* C, long ldc, long n,
                        long ns, long K, double beta, hipStream_t s) {
  if (n <= 0 || ns <= 0 || K <= 0) return 0;
  if (n % TILE || ns % TILE || K % TILE) {
    set_error("gemm_nt_lz: n, ns, K must be multiples of 128");
    return -1;
  }
  long n_tr = n / TILE, n_tc = ns / TILE;
  long groups = ((n_tr + 7) / 8 + 7) / 8;
  hipLaunchKernelGGL(gemm_nt_dma_kernel<0>, dim3((unsigned)(groups * 8 * n_tc * 8)), dim3(512), 0, s, L, ldl, Zt, ldz, C,
                     ldc, K, 1.0, beta, -(1L << 40), n_tr, n_tc, 0L, (const double*)C, ldc, 2, TileSkip());
  SGP_HIP(hipGetLastError());
  return 0;
}
int launch_gemm_nt_lz(const double* L, long ldl, const double* Zt, long ldz, double* C, long ldc, long n,
                      long ns, double beta, hipStream_t s) {
  return launch_gemm_nt_lz_k(L, ldl, Zt, ldz, C, ldc, n, ns, n, beta, s);
}

// Cpart[s] = A[:, sK' : (s+1)K'] B[:, sK' : (s+1)K']'  for s < nsplit (K' = K / nsplit, a multiple of
// 16), slabs `part_stride` doubles apart: one launch, nsplit x the tiles -- for Gram matrices
// with a huge contraction dimension and few output tiles (VFE: A A', M = 4096, K = 262144).
// Further k split of the leftover tiles of the one-slice-per-XCD Gram product (see the kernel, klo == 3): 64 / leftover when
// that is a whole number > 1 and the pieces stay multiples of the 16-column chunk; 1 = none.  The slab buffer then holds
// 8 * sub slabs (launch_gemm_nt_splitk / launch_splitk_reduce callers size it with splitk_slabs).
long splitk_sub(long M, long K) {
  const long n_t = M / TILE, tiles = n_t * (n_t + 1) / 2, left = tiles % 64;
  if (left == 0 || tiles < 64 || 64 % left) return 1;
  const long sub = 64 / left;
  if (sub < 2 || sub > 8 || K % (8 * sub * KB)) return 1;
  return sub;
}
long splitk_slabs(long M, long K, int nsplit) { return nsplit == 8 ? 8 * splitk_sub(M, K) : nsplit; }

int launch_gemm_nt_splitk(const double* A, long lda, const double* B, long ldb, double* Cpart, long ldc,
                          long M, long Nc, long K, int nsplit, long part_stride, int lower_only,
                          hipStream_t s) {
  if (M % TILE || Nc % TILE || nsplit < 1 || K % (KB * (long)nsplit)) {
    set_error("gemm_nt_splitk: bad sizes");
    return -1;
  }
  long n_tr = M / TILE, n_tc = Nc / TILE;
  if (lower_only && nsplit == 8 && M == Nc) {  // one K slice per XCD (see the kernel, klo == 3)
    const long tiles = n_tr * (n_tr + 1) / 2;
    const long sub = splitk_sub(M, K);
    const long full = sub > 1 ? tiles / 64 * 64 : tiles;
    const long ids = full + (tiles - full) * sub;
    hipLaunchKernelGGL(gemm_nt_dma_kernel<0>, dim3((unsigned)(ids * 8)), dim3(512), 0, s, A, lda, B, ldb, Cpart, ldc,
                       K / nsplit, 1.0, 0.0, sub, n_tr, n_tc, part_stride, (const double*)Cpart, ldc, 3, TileSkip());
    SGP_HIP(hipGetLastError());
    return 0;
  }
  long groups = ((n_tr + 7) / 8 + 7) / 8;
  long mask_off = lower_only ? 0 : -(1L << 40);
  long per_xcd = lower_only ? tri_ids_per_xcd(tri_shape(n_tr, n_tc, -1)) : groups * 8 * n_tc;
  dim3 grid((unsigned)(per_xcd * 8), (unsigned)nsplit);
  hipLaunchKernelGGL(gemm_nt_dma_kernel<0>, grid, dim3(512), 0, s, A, lda, B, ldb, Cpart, ldc, K / nsplit, 1.0,
                     0.0, mask_off, n_tr, n_tc, part_stride, (const double*)Cpart, ldc, 0, TileSkip());
  SGP_HIP(hipGetLastError());
  return 0;
}

// C = beta C + alpha sum_s part[s]   (fixed order: deterministic).  sub > 1 (the XCD-sliced Gram product with its leftover
// tiles split further): whole tiles sum slabs 0, sub, 2 sub, ..., leftover tiles all 8 * sub slabs -- ascending k either way.
__global__ void splitk_reduce_kernel(const double* part, long part_stride, int nsplit, double* C,
                                     long ldc, long M, long Nc, double alpha, double beta,
                                     int lower_only, int sub, long full) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= M * Nc) return;
  long r = idx % M, c = idx / M;
  const long tr = r / TILE, tc = c / TILE;
  if (lower_only && tr < tc) return;
  double s = 0.0;
  if (sub > 1) {
    const bool whole = tr * (tr + 1) / 2 + tc < full;
    const int step = whole ? sub : 1;
    for (int k = 0; k < nsplit * sub; k += step) s += part[k * part_stride + r + c * ldc];
  } else {
    for (int k = 0; k < nsplit; ++k) s += part[k * part_stride + r + c * ldc];
  }
  double* p = C + r + c * ldc;
  double old = (beta == 0.0) ? 0.0 : beta * (*p);
  *p = old + alpha * s;
}

// K: the contraction length the slabs came from (launch_gemm_nt_splitk's K; 0 = plain nsplit slabs)
int launch_splitk_reduce(const double* part, long part_stride, int nsplit, double* C, long ldc, long M,
                         long Nc, double alpha, double beta, int lower_only, hipStream_t s, long K) {
  long tot = M * Nc;
  const long sub = (K > 0 && lower_only && nsplit == 8 && M == Nc) ? splitk_sub(M, K) : 1;
  const long n_t = M / TILE, tiles = n_t * (n_t + 1) / 2;
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, part,
                     part_stride, nsplit, C, ldc, M, Nc, alpha, beta, lower_only, (int)sub, tiles / 64 * 64);
  SGP_HIP(hipGetLastError());
  return 0;
}

}  // namespace sgp
