// fp32 instantiation of the hot path: covariance assembly + blocked Cholesky logpdf in single precision
// (the reference is type-stable in Float32: /root/reference/test/gp/util.jl:76-88).
//
// A second, smaller set of kernels -- the fp64 kernels are designs around the f64 MFMA forms and their
// lane maps, not templates -- sharing the data layout (bordered m_tot x n_pad matrix, 128-tile grid,
// (y - m)' as a bordered row so that the forward substitution rides along) and the flattened spec:
//   assemble_f32_kernel     fused distance + kappa + scales + sum of terms + noise, fp32 arithmetic
//                           (inputs stay fp64 in HBM and are rounded once while staged into LDS)
//   gemm_nt_f32_kernel      C -= A B' on v_mfma_f32_32x32x2_f32 (exact f32, 157 TFLOP/s peak): 128 x 128 tile,
//                           4 waves x (2 x 2) MFMA tiles, K chunks of 16 through two LDS stages
//   diagonal block + row solve: the fp64 path's potrf_diag / panel_solve kernels instantiated on fp32 storage
//                           (potrf.hip: values converted while they are read and written, arithmetic in fp64) --
//                           fp32 kernels of their own took 75 + 60 us per 128 columns against 33 + 17 us, and
//                           made N <= 16384 slower than fp64 (profiles/archive/r02_microbench.md)
// Driver: two-level right-looking Cholesky (outer panels, 128-column steps inside) with the fp64 driver's
// one-panel look-ahead on two streams.
// Entry points: sgp_logpdf_f32, sgp_kernelmatrix_f32, sgp_rand_f32, sgp_posterior_mean_var_f32 (include/sthenomi.h).  Accuracy is fp32's: the
// tests hold 1e-4 relative on logpdf against the fp64 oracle at N <= 3000.
#include "driver.h"
#include "tilemap.h"

#include <algorithm>
#include <cmath>
#include <vector>

using namespace sgp;

namespace {

typedef float f16v __attribute__((ext_vector_type(16)));
constexpr int LDF = 128 + 4;     // LDS leading dimension (floats) of a 128-row operand chunk

enum { K_SE = 0, K_M12 = 1, K_M32 = 2, K_M52 = 3, K_WHITE = 4, K_CONST = 5 };

__device__ __forceinline__ float kern_f32(int kind, float d2, float param) {
  switch (kind) {
    case K_SE: return __expf(-0.5f * d2);
    case K_M12: return __expf(-sqrtf(d2));
    case K_M32: {
      float l = 1.7320508f * sqrtf(d2);
      return (1.0f + l) * __expf(-l);
    }
    case K_M52: {
      float l = 2.236068f * sqrtf(d2);
      return fmaf(l, fmaf(l, 0.33333334f, 1.0f), 1.0f) * __expf(-l);
    }
    case K_WHITE: return d2 == 0.0f ? 1.0f : 0.0f;
    default: return param;
  }
}

// one 256-thread workgroup per 128 x 128 tile; thread t owns row (t & 127) and 64 columns of half t >> 7
template <int DMAX>
__global__ __launch_bounds__(256) void assemble_f32_kernel(float* K, long ld, long r0, long nr, long c0, long nc,
                                                           const DevTerm* terms, int nterms, int lower_only,
                                                           int noise_kind, float sigma2, const double* noise_diag,
                                                           long tile_r_first, long tile_c_first) {
  const long gtr = tile_r_first + blockIdx.x, gtc = tile_c_first + blockIdx.y;
  if (lower_only && gtr < gtc) return;
  extern __shared__ __attribute__((aligned(16))) float smf[];  // [nterms][128][DMAX] column points
  const int t = threadIdx.x, trow = t & 127, th = t >> 7;
  long cbeg = std::max(gtc * TILE, c0), cend = std::min(gtc * TILE + TILE, c0 + nc);
  long rbeg = std::max(gtr * TILE, r0), rend = std::min(gtr * TILE + TILE, r0 + nr);
  if (cbeg >= cend || rbeg >= rend) return;
  for (int tm = 0; tm < nterms; ++tm) {
    const DevTerm T = terms[tm];
    for (int idx = t; idx < TILE * DMAX; idx += 256) {
      const int p = idx / DMAX, d = idx % DMAX;
      const long gc = gtc * TILE + p;
      smf[(tm * TILE + p) * DMAX + d] = (d < T.dim && gc >= cbeg && gc < cend) ? (float)T.xc[(gc - c0) * T.ldc + d] : 0.0f;
    }
  }
  __syncthreads();
  const long grow = gtr * TILE + trow;
  if (grow < rbeg || grow >= rend) return;
  const long lrow = grow - r0;
  for (int jc = 0; jc < 64; jc += 8) {
    const int pbase = th * 64 + jc;
    if (gtc * TILE + pbase >= cend) break;
    float acc[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[q] = 0.0f;
    for (int tm = 0; tm < nterms; ++tm) {
      const DevTerm T = terms[tm];
      float xi[DMAX];
#pragma unroll
      for (int d = 0; d < DMAX; ++d) xi[d] = (d < T.dim) ? (float)T.xr[lrow * T.ldr + d] : 0.0f;
      const float rsv = (float)(T.coef * (T.rs ? T.rs[lrow] : 1.0));
      const float* sp = &smf[(tm * TILE + pbase) * DMAX];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const long gc = gtc * TILE + pbase + q;
        float d2 = 0.0f;
#pragma unroll
        for (int d = 0; d < DMAX; ++d) {
          const float df = xi[d] - sp[q * DMAX + d];
          d2 = fmaf(df, df, d2);
        }
        float cw = rsv;
        if (T.cs) cw = (gc >= cbeg && gc < cend) ? rsv * (float)T.cs[gc - c0] : 0.0f;
        acc[q] = fmaf(kern_f32(T.kind, d2, (float)T.param), cw, acc[q]);
      }
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const long gc = gtc * TILE + pbase + q;
      if (gc >= cbeg && gc < cend) {
        float v = acc[q];
        if (noise_kind >= 0 && gc == grow) v += (noise_kind == 0) ? sigma2 : (float)noise_diag[grow];
        K[grow + gc * ld] = v;
      }
    }
  }
}

// identity padding of rows / columns N .. n_pad, zeros below the square part of padded columns
__global__ void fill_pad_f32_kernel(float* K, long ld, long N, long n_pad, long m_tot) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long h = n_pad - N;
  if (h <= 0) return;
  if (idx < h * n_pad) {                       // rows [N, n_pad) of every column
    const long r = N + idx % h, c = idx / h;
    K[r + c * ld] = (r == c) ? 1.0f : 0.0f;
  }
  if (idx < h * m_tot) {                       // columns [N, n_pad), all rows
    const long c = N + idx % h, r = idx / h;
    K[r + c * ld] = (r == c) ? 1.0f : 0.0f;
  }
}

// bordered rows: A[n_pad + s, c] = y[c] - mean[c] (s == 0, c < N), 0 otherwise (128 rows)
__global__ void border_f32_kernel(float* A, long ld, long n_pad, long N, const double* y, const double* mean) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)TILE * n_pad) return;
  const long s = idx % TILE, c = idx / TILE;
  A[n_pad + s + c * ld] = (s == 0 && c < N) ? (float)(y[c] - (mean ? mean[c] : 0.0)) : 0.0f;
}

// ---------------------------------------------------------------------------------------------------------
// C (M x Nc, lower 128-tiles when lower != 0) -= A (M x K) B (Nc x K)', all column-major fp32.
// MFMA 32x32x2 f32 (guide section 3): a-operand lane l = Aop[i = l & 31][k = l >> 5], b-operand lane l =
// Bop[k = l >> 5][j = l & 31], result reg r = D[i = (r & 3) + 8 (r >> 2) + 4 (l >> 5)][j = l & 31].  The tile is
// computed TRANSPOSED (Aop = rows of B, Bop = rows of A), so j = l & 31 runs along the contiguous row index
// of C: one store instruction writes 32 consecutive rows of two columns.
// ---------------------------------------------------------------------------------------------------------
template <int KBF, int OCC>
__global__ __launch_bounds__(256, OCC) void gemm_nt_f32_kernel(const float* A, long lda, const float* B, long ldb, float* C,
                                                            long ldc, long K, int lower, long n_tr, long n_tc) {
  long tr, tc;
  if (lower) {   // 1-D grid over the live tiles only, XCD-aware (tilemap.h: the fp64 update's enumeration -- every A row
                 // panel goes through one XCD's L2, 64 consecutive workgroups of an XCD share 8 + 8 operand panels)
    if (!tile_of_id((long)blockIdx.x, n_tr, n_tc, 0L, tr, tc)) return;
  } else {
    tr = blockIdx.x;
    tc = blockIdx.y;
  }
  if (tr >= n_tr || tc >= n_tc) return;
  __shared__ __attribute__((aligned(16))) float sA[2][KBF * LDF];
  __shared__ __attribute__((aligned(16))) float sB[2][KBF * LDF];
  const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);   // (wave-uniform for the compiler too)
  const int wr = w >> 1, wc = w & 1;       // wave tile: rows wr * 64 .., cols wc * 64 ..
  const int l31 = lane & 31, lh = lane >> 5;
  const float* Ag = A + tr * TILE;
  const float* Bg = B + tc * TILE;
  // staging: thread t moves rows 4 (t & 31) .. +3 of columns (t >> 5) and (t >> 5) + 8 of both operands
  const int srow = 4 * (t & 31), scol = t >> 5;
  // (seeding the accumulators with the old C tile in the prologue, as the fp64 kernel does, was tried in round 4: the 32
  // column addresses then stay live across the main loop -- 118 -> 176 .. 238 VGPRs, half the occupancy, 1057 -> 1266 ms at
  // N = 65536 -- so the tile is read-modified-written in the epilogue)
  f16v acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
  constexpr int NU = KBF / 8;   // staging units per thread and operand
  float4 ra[NU], rb[NU];
  auto gload = [&](long k0) {
#pragma unroll
    for (int i = 0; i < NU; ++i) {
      ra[i] = *reinterpret_cast<const float4*>(Ag + srow + (k0 + scol + 8 * i) * lda);
      rb[i] = *reinterpret_cast<const float4*>(Bg + srow + (k0 + scol + 8 * i) * ldb);
    }
  };
  auto sstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NU; ++i) {
      *reinterpret_cast<float4*>(&sA[buf][(scol + 8 * i) * LDF + srow]) = ra[i];
      *reinterpret_cast<float4*>(&sB[buf][(scol + 8 * i) * LDF + srow]) = rb[i];
    }
  };
  auto compute = [&](int buf) {
#pragma unroll
    for (int ks = 0; ks < KBF / 2; ++ks) {
      const int kk = 2 * ks + lh;
      float av[2], bv[2];
#pragma unroll
      for (int a = 0; a < 2; ++a) av[a] = sA[buf][kk * LDF + wr * 64 + a * 32 + l31];   // rows of the C tile
#pragma unroll
      for (int b = 0; b < 2; ++b) bv[b] = sB[buf][kk * LDF + wc * 64 + b * 32 + l31];   // columns of the C tile
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[b], av[a], acc[a][b], 0, 0, 0);
    }
  };
  gload(0);
  sstore(0);
  __syncthreads();
  int buf = 0;
  for (long k0 = KBF; k0 < K; k0 += KBF) {
    gload(k0);
    compute(buf);
    sstore(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }
  compute(buf);
  // acc[a][b][r]: row = wr * 64 + a * 32 + l31, column = wc * 64 + b * 32 + (r & 3) + 8 (r >> 2) + 4 lh
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const long row = tr * TILE + wr * 64 + a * 32 + l31;
        const long col = tc * TILE + wc * 64 + b * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        float* p = C + row + col * ldc;
        *p -= acc[a][b][r];
      }
}

// Round 4: the same tile program with the fp64 kernel's operand path -- global -> LDS directly (global_load_lds_dwordx4:
// no staging VGPRs, no ds_write), one chunk ahead of the MFMAs, raw s_barrier + one counted wait per chunk.  One wave
// instruction moves 1 KiB = TWO 128-row fp32 columns, so the LDS chunk is unpadded (leading dimension 128 floats: the second
// column must follow the first); the ds_read_b32 operand fetches (lanes 0 .. 31 consecutive rows, lanes 32 .. 63 the next
// k) are conflict free on it.  256 threads = 4 waves x (2 x 2) MFMA 32x32x2 tiles as above; 2 stages x 2 operands x KD x 512
// B = 32 KB of LDS at KD = 16.
template <int KD>
__global__ __launch_bounds__(256, 4) void gemm_nt_f32_dma_kernel(const float* A, long lda, const float* B, long ldb, float* C,
                                                                 long ldc, long K, int lower, long n_tr, long n_tc, TileSkip sk) {
  long tr, tc;
  if (lower) {
    if (!tile_of_id((long)blockIdx.x, n_tr, n_tc, 0L, tr, tc)) return;
  } else {
    tr = blockIdx.x;
    tc = blockIdx.y;
  }
  if (tr >= n_tr || tc >= n_tc) return;
  // structural zeros (round 6; the fp64 tile program's rule, gemm_nt.hip): the update P[tr] P[tc]' is dead when every k tile
  // of the panel has a structurally zero operand tile -- C keeps its bits (C - 0); a live tile contracts from its first to
  // its last live k tile only (the products outside are exact zeros).  Tile boundaries are multiples of KD, so the chunks,
  // and with them the bits, are those of the dense run.
  if (sk.nz) {
    const sz_word* ra = sk.nz + (long)(sk.tr0 + tr) * sk.words;
    const sz_word* rb = sk.nz + (long)(sk.tc0 + tc) * sk.words;
    int kmin = -1, kmax = -1;
    for (int q = sk.kt0 >> 6; q <= (sk.kt1 - 1) >> 6; ++q) {
      sz_word m = ra[q] & rb[q];
      if (q == (sk.kt0 >> 6)) m &= ~(sz_word)0 << (sk.kt0 & 63);
      if (q == ((sk.kt1 - 1) >> 6) && (sk.kt1 & 63)) m &= ~(~(sz_word)0 << (sk.kt1 & 63));
      if (m != 0) {
        if (kmin < 0) kmin = q * 64 + (int)__builtin_ctzll(m);
        kmax = q * 64 + 63 - (int)__builtin_clzll(m);
      }
    }
    if (kmin < 0) return;
    A += (long)(kmin - sk.kt0) * TILE * lda;
    B += (long)(kmin - sk.kt0) * TILE * ldb;
    K = (long)(kmax - kmin + 1) * TILE;
  }
  constexpr int STAGE = 2 * KD * TILE;   // floats per stage: A chunk, then B chunk
  __shared__ __attribute__((aligned(16))) float smem[2 * STAGE];
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wr = w >> 1, wc = w & 1;
  const int l31 = lane & 31, lh = lane >> 5;
  const float* Ag = A + tr * TILE + 4 * l31;   // this lane's 4 rows of the column pair it moves
  const float* Bg = B + tc * TILE + 4 * l31;
  auto dma = [&](long k0, int stage) {
    float* sa = smem + stage * STAGE;
    float* sb = sa + KD * TILE;
#pragma unroll
    for (int i = 0; i < KD / 8; ++i) {   // wave w moves column pairs w, w + 4, ...: columns 2 p and 2 p + 1
      const int p = w + 4 * i;
      __builtin_amdgcn_global_load_lds((gptr_t)(Ag + (k0 + 2 * p + lh) * lda), (lptr_t)(sa + 2 * p * TILE), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)(Bg + (k0 + 2 * p + lh) * ldb), (lptr_t)(sb + 2 * p * TILE), 16, 0, 0);
    }
  };
  f16v acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
  const long nch = K / KD;
  dma(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  for (long c = 0; c < nch; ++c) {
    const int stage = (int)(c & 1);
    if (c + 1 < nch) dma((c + 1) * KD, stage ^ 1);
    const float* sa = smem + stage * STAGE + lh * TILE + wr * 64 + l31;
    const float* sb = smem + stage * STAGE + KD * TILE + lh * TILE + wc * 64 + l31;
#pragma unroll
    for (int ks = 0; ks < KD / 2; ++ks) {
      float av[2], bv[2];
#pragma unroll
      for (int a = 0; a < 2; ++a) av[a] = sa[2 * ks * TILE + a * 32];
#pragma unroll
      for (int b = 0; b < 2; ++b) bv[b] = sb[2 * ks * TILE + b * 32];
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[b], av[a], acc[a][b], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  // C -= acc: the 16 loads of one MFMA tile in flight together, then its 16 stores (a load-wait-store chain per element
  // would serialise 64 global round trips per thread)
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      float* p = C + tr * TILE + wr * 64 + a * 32 + l31 + (tc * TILE + wc * 64 + b * 32 + 4 * lh) * ldc;
      float cv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) cv[r] = __builtin_nontemporal_load(p + ((r & 3) + 8 * (r >> 2)) * ldc);
#pragma unroll
      for (int r = 0; r < 16; ++r) p[((r & 3) + 8 * (r >> 2)) * ldc] = cv[r] - acc[a][b][r];
    }
}

int launch_gemm_f32(const float* A, long lda, const float* B, long ldb, float* C, long ldc, long M, long Nc, long K,
                    int lower, hipStream_t s) {
  if (M <= 0 || Nc <= 0 || K <= 0) return 0;
  const long n_tr = M / TILE, n_tc = Nc / TILE;
  dim3 grid((unsigned)n_tr, (unsigned)n_tc);
  if (lower) grid = dim3((unsigned)tile_ids(n_tr, n_tc, 0));   // (M >= Nc for every lower update)
  // (KBF = 32 and launch bounds asking for 3 - 4 workgroups per CU measured no better: 1244 / 1068 / 1070 ms vs
  // 1069 ms at N = 65536 -- gpurun_out/f32_gemm_ab.txt; round 4, with the deep serial schedule: <16, 4> (128 VGPRs, 4
  // spills) 1036 ms vs <16, 2> (130 VGPRs) 987 ms)
  // SGP_F32_DMA (read per launch, so that one process can compare the two): unset / 1 the LDS-DMA kernel, 0 the
  // register-staged one of round 3.  The 16-byte lane loads want 16-byte aligned operands and leading dimensions that are
  // multiples of 4 floats (every caller's are multiples of 128).  Measured at N = 65536 (logpdf_f32, whole call): 767.6 ms
  // against 992.3 ms; a 32-deep chunk (2 workgroups per CU instead of 4) 796.8 ms.
  const char* e = getenv("SGP_F32_DMA");
  if ((!e || atoi(e) != 0) && K % 16 == 0 && lda % 4 == 0 && ldb % 4 == 0 && (((uintptr_t)A | (uintptr_t)B) & 15) == 0) {
    const TileSkip sk = (lower && A == B) ? gemm_skip_for_f32(A, lda, C, ldc, K) : TileSkip();
    hipLaunchKernelGGL((gemm_nt_f32_dma_kernel<16>), grid, dim3(256), 0, s, A, lda, B, ldb, C, ldc, K, lower, n_tr, n_tc, sk);
    SGP_HIP(hipGetLastError());
    return 0;
  }
  hipLaunchKernelGGL((gemm_nt_f32_kernel<16, 2>), grid, dim3(256), 0, s, A, lda, B, ldb, C, ldc, K, lower, n_tr, n_tc);
  SGP_HIP(hipGetLastError());
  return 0;
}

__global__ void rowsumsq_f32_kernel(const float* row, long ld, long nc, double* out) {
  __shared__ double sh[4];
  double acc = 0.0;
  for (long c = threadIdx.x; c < nc; c += blockDim.x) {
    const double v = (double)row[c * ld];
    acc = fma(v, v, acc);
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

__global__ void mirror_cast_f32_kernel(const float* K, long ld, long N, long M, int symmetric, float* out, long ldo) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= N * M) return;
  const long r = idx % N, c = idx / N;
  out[r + c * ldo] = (symmetric && r < c) ? K[c + r * ld] : K[r + c * ld];
}

template <int DMAX>
int launch_assemble_f32_t(float* K, long ld, long r0, long nr, long c0, long nc, const DevTerm* d_terms, int nterms,
                          int lower_only, int noise_kind, float sigma2, const double* d_noise, long trf, long tcf,
                          long trc, long tcc, hipStream_t s) {
  size_t lds = std::max<size_t>(16, (size_t)nterms * TILE * DMAX * sizeof(float));
  hipLaunchKernelGGL(assemble_f32_kernel<DMAX>, dim3((unsigned)trc, (unsigned)tcc), dim3(256), lds, s, K, ld, r0, nr, c0,
                     nc, d_terms, nterms, lower_only, noise_kind, sigma2, d_noise, trf, tcf);
  SGP_HIP(hipGetLastError());
  return 0;
}

int assemble_f32(const sgp_dspec* ds, float* K, long ld, int lower_only, int noise_kind, float sigma2,
                 const double* d_noise, hipStream_t s) {
  for (int I = 0; I < ds->nrb; ++I)
    for (int J = 0; J < ds->ncb; ++J) {
      if (ds->row_len[I] == 0 || ds->col_len[J] == 0 || (lower_only && I < J)) continue;
      const long r0 = ds->row_off[I], nr = ds->row_len[I], c0 = ds->col_off[J], nc = ds->col_len[J];
      const long trf = r0 / TILE, trl = (r0 + nr - 1) / TILE + 1, tcf = c0 / TILE, tcl = (c0 + nc - 1) / TILE + 1;
      const int p = I * ds->ncb + J;
      const int t0 = ds->term_ptr[p], t1 = ds->term_ptr[p + 1];
      const int dmax = ds->pair_dmax[p];
      if (dmax > 16 || (t1 - t0) * dmax > 64) {
        set_error("fp32 path: input dimension <= 16 and (terms per block pair) x dimension <= 64");
        return -1;
      }
      const int nk = (ds->symmetric && I == J) ? noise_kind : -1;
#define SGP_A32(DM) \
  return_code = launch_assemble_f32_t<DM>(K, ld, r0, nr, c0, nc, ds->d_terms + t0, t1 - t0, lower_only, nk, sigma2, d_noise, trf, tcf, trl - trf, tcl - tcf, s)
      int return_code = 0;
      if (dmax <= 1) SGP_A32(1);
      else if (dmax <= 2) SGP_A32(2);
      else if (dmax <= 4) SGP_A32(4);
      else if (dmax <= 8) SGP_A32(8);
      else SGP_A32(16);
#undef SGP_A32
      if (return_code) return return_code;
    }
  return 0;
}

// factor one column panel in place (128-column steps: diagonal block, row solve, K = 128 update of the rest
// of the panel)
int panel_factor_f32(sgp_ctx* ctx, float* P, long ld, long m, long w, long g0, hipStream_t s) {
  for (long j = 0; j < w; j += TILE) {
    float* D = P + j + j * ld;
    // diagonal block and row solve in fp64 arithmetic on the fp32 storage (potrf.hip)
    if (int rc = launch_potrf_diag_f32(D, ld, ctx->d_invd, ctx->d_slots + (g0 + j) / TILE, ctx->d_info, g0 + j, s)) return rc;
    const long mrest = m - j - TILE;
    if (mrest > 0) {
      float* A21 = P + (j + TILE) + j * ld;
      const StripSkip sk = strip_skip_for_f32(A21, ld);   // (empty unless sgp_logpdf_f32 registered a pattern)
      if (int rc = launch_panel_solve_f32(A21, ld, mrest, D, ld, ctx->d_invd, 256, 16, s, &sk)) return rc;
      const long wrest = w - j - TILE;
      if (wrest > 0)
        if (int rc = launch_gemm_f32(A21, ld, A21, ld, P + (j + TILE) + (j + TILE) * ld, ld, mrest, wrest, TILE, 1, s)) return rc;
    }
  }
  return 0;
}

// two-level right-looking Cholesky of the bordered fp32 matrix (m_tot x n_pad, ld) with the fp64 driver's
// one-panel look-ahead (capi.hip: chol_bordered): the next panel is updated and factored on the panel stream
// while the rest of the trailing matrix is updated with the current panel on the update stream
// an outer panel of w columns by recursive halving down to wmid (the fp64 driver's panel_factor_mid): the left half, ONE
// update of the right half with it (K = the half's width), the right half
int panel_factor_mid_f32(sgp_ctx* ctx, float* P, long ld, long m, long w, long g0, hipStream_t s, long wmid) {
  if (wmid <= 0 || w <= wmid) return panel_factor_f32(ctx, P, ld, m, w, g0, s);
  const long wl = std::max(wmid, (w / 2 + wmid - 1) / wmid * wmid);
  if (wl >= w) return panel_factor_f32(ctx, P, ld, m, w, g0, s);
  if (int rc = panel_factor_mid_f32(ctx, P, ld, m, wl, g0, s, wmid)) return rc;
  if (int rc = launch_gemm_f32(P + wl, ld, P + wl, ld, P + wl + wl * ld, ld, m - wl, w - wl, wl, 1, s)) return rc;
  return panel_factor_mid_f32(ctx, P + wl + wl * ld, ld, m - wl, w - wl, g0 + wl, s, wmid);
}

int chol_f32(sgp_ctx* ctx, float* A, long ld, long n_pad, long m_tot, hipStream_t s) {
  // Round 4: from 32768 columns on the fp64 driver's serial deep schedule -- outer panels of
  // 4096 columns factored by recursive halving down to 512, one K = 4096 trailing update per panel, no look-ahead: an fp32
  // tile runs twice as fast as an fp64 one, so the per-tile prologue and the C-tile traffic of shallow (K = 512) updates
  // weigh twice as much.
  const bool deep = n_pad >= 32768;
  const long W = deep ? 4096 : (n_pad <= 2048 ? n_pad : (n_pad <= 8192 ? 1024 : 512));
  const long WMID = deep ? 512 : 0;
  const bool la = ctx->lookahead && s == ctx->stream && !deep;
  hipStream_t sB = la ? ctx->stream2 : s;
  bool rest_pending = false;
  if (la) {
    SGP_HIP(hipEventRecord(ctx->ev_panel, s));
    SGP_HIP(hipStreamWaitEvent(sB, ctx->ev_panel, 0));
  }
  for (long J0 = 0; J0 < n_pad; J0 += W) {
    const long wj = std::min(W, n_pad - J0);
    if (int rc = panel_factor_mid_f32(ctx, A + J0 + J0 * ld, ld, m_tot - J0, wj, J0, s, WMID)) return rc;
    const long c0 = J0 + wj;
    if (c0 >= n_pad) break;
    const long w1 = std::min(W, n_pad - c0), c1 = c0 + w1;
    const float* Pj = A + J0 * ld;
    if (la) {
      SGP_HIP(hipEventRecord(ctx->ev_panel, s));
      if (rest_pending) SGP_HIP(hipStreamWaitEvent(s, ctx->ev_rest, 0));
      if (int rc = launch_gemm_f32(Pj + c0, ld, Pj + c0, ld, A + c0 + c0 * ld, ld, m_tot - c0, w1, wj, 1, s)) return rc;
      if (c1 < n_pad) {
        SGP_HIP(hipStreamWaitEvent(sB, ctx->ev_panel, 0));
        if (int rc = launch_gemm_f32(Pj + c1, ld, Pj + c1, ld, A + c1 + c1 * ld, ld, m_tot - c1, n_pad - c1, wj, 1, sB)) return rc;
        SGP_HIP(hipEventRecord(ctx->ev_rest, sB));
        rest_pending = true;
      }
    } else {
      if (int rc = launch_gemm_f32(Pj + c0, ld, Pj + c0, ld, A + c0 + c0 * ld, ld, m_tot - c0, n_pad - c0, wj, 1, s)) return rc;
    }
  }
  if (la && rest_pending) SGP_HIP(hipStreamWaitEvent(s, ctx->ev_rest, 0));
  return 0;
}

}  // namespace

#define F_CHECK_ARG(cond, msg) \
  do {                         \
    if (!(cond)) {             \
      sgp::set_error(msg);     \
      return -1;               \
    }                          \
  } while (0)

int sgp_dspec_create_nolock(sgp_ctx* ctx, const sgp_cov_spec* sp, sgp_dspec** out);
void sgp_dspec_free_nolock(sgp_dspec* ds);

extern "C" int sgp_logpdf_f32(sgp_ctx* ctx, const sgp_cov_spec* spec, const double* mean, int noise_kind,
                              const double* noise, const double* y, double* out) {
  F_CHECK_ARG(ctx && spec && noise && y && out, "sgp_logpdf_f32: NULL argument");
  F_CHECK_ARG(spec->symmetric, "sgp_logpdf_f32: spec must be symmetric");
  F_CHECK_ARG(noise_kind == SGP_NOISE_SCALAR || noise_kind == SGP_NOISE_DIAG,
              "sgp_logpdf_f32: noise kind must be SCALAR or DIAG");
  CtxScope scope(ctx);
  sgp_dspec* ds = nullptr;
  if (int rc = sgp_dspec_create_nolock(ctx, spec, &ds)) return rc;
  struct G {
    sgp_dspec* d;
    ~G() { sgp_dspec_free_nolock(d); }
  } g{ds};
  const long N = ds->N;
  F_CHECK_ARG(N >= 1, "sgp_logpdf_f32: empty data");
  int64_t n_pad, m_tot;
  sgp_geometry(N, 1, &n_pad, &m_tot);
  F_CHECK_ARG(n_pad / TILE <= ctx->n_slots, "matrix too large for the logdet slot buffer");
  hipStream_t s = ctx->stream;
  DevBuf dA, dy, dm, dn;
  if (int rc = dA.alloc((size_t)(m_tot * n_pad + 1) / 2 + 64)) return rc;   // floats in a double-typed buffer
  if (int rc = dy.upload(y, N)) return rc;
  if (mean)
    if (int rc = dm.upload(mean, N)) return rc;
  if (noise_kind == SGP_NOISE_DIAG)
    if (int rc = dn.upload(noise, N)) return rc;
  float* A = reinterpret_cast<float*>(dA.p);
  SGP_HIP(hipMemsetAsync(ctx->d_info, 0, sizeof(int), s));
  if (int rc = assemble_f32(ds, A, m_tot, 1, noise_kind, noise_kind == SGP_NOISE_SCALAR ? (float)noise[0] : 0.0f,
                            dn.p, s))
    return rc;
  {
    const long h = n_pad - N;
    if (h > 0) {
      const long tot = h * m_tot;
      hipLaunchKernelGGL(fill_pad_f32_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, A, (long)m_tot, N,
                         (long)n_pad, (long)m_tot);
    }
    const long totb = (long)TILE * n_pad;
    hipLaunchKernelGGL(border_f32_kernel, dim3((unsigned)((totb + 255) / 256)), dim3(256), 0, s, A, (long)m_tot, (long)n_pad, N,
                       dy.p, mean ? dm.p : nullptr);
    SGP_HIP(hipGetLastError());
  }
  // structural zeros (round 6): the fp64 driver's tile pattern of the factor (capi.hip: sz_pattern, host; one upload) --
  // dead tile products are skipped, a live tile contracts its live k range, the row solve leaves zero tiles alone.  The
  // bits are those of the dense run (SGP_STRUCT_ZEROS=0).
  {
    int words = 0;
    const sz_word* d_nz = nullptr;
    if (int rc = drv_sz_pattern(ctx, ds, noise_kind, n_pad, m_tot, &words)) return rc;
    if (words > 0)
      if (int rc = drv_sz_upload(ctx, ctx, words, s, &d_nz)) return rc;
    struct Scope {
      bool on;
      ~Scope() { if (on) gemm_set_structure(nullptr, 0, nullptr, 0); }
    } scope_sz{d_nz != nullptr};
    if (d_nz) gemm_set_structure_f32(A, m_tot, d_nz, words, n_pad);
    if (int rc = chol_f32(ctx, A, m_tot, n_pad, m_tot, s)) return rc;
  }
  double* d_logdet = ctx->d_scal;
  double* d_sq = ctx->d_scal + 16;
  hipLaunchKernelGGL(rowsumsq_f32_kernel, dim3(1), dim3(256), 0, s, A + n_pad, (long)m_tot, N, d_sq);
  if (int rc = launch_sum_array(ctx->d_slots, n_pad / TILE, d_logdet, s)) return rc;
  double h2[17];
  SGP_HIP(hipMemcpyAsync(h2, ctx->d_scal, sizeof(double) * 17, hipMemcpyDeviceToHost, s));
  int info = 0;
  SGP_HIP(hipMemcpyAsync(&info, ctx->d_info, sizeof(int), hipMemcpyDeviceToHost, s));
  SGP_HIP(hipStreamSynchronize(s));
  if (info > 0) {
    set_error("matrix is not positive definite (fp32); Cholesky factorization failed at leading minor " +
              std::to_string(info));
    return info;
  }
  out[0] = -0.5 * ((double)N * 1.8378770664093453 + h2[0] + h2[16]);
  return 0;
}

extern "C" int sgp_kernelmatrix_f32(sgp_ctx* ctx, const sgp_cov_spec* spec, float* K, int64_t ldk) {
  F_CHECK_ARG(ctx && spec && K, "sgp_kernelmatrix_f32: NULL argument");
  CtxScope scope(ctx);
  sgp_dspec* ds = nullptr;
  if (int rc = sgp_dspec_create_nolock(ctx, spec, &ds)) return rc;
  struct G {
    sgp_dspec* d;
    ~G() { sgp_dspec_free_nolock(d); }
  } g{ds};
  const long N = ds->N, M = ds->M;
  F_CHECK_ARG(ldk >= N, "sgp_kernelmatrix_f32: ldk < N");
  if (N == 0 || M == 0) return 0;
  hipStream_t s = ctx->stream;
  DevBuf dK, dO;
  if (int rc = dK.alloc((size_t)(N * M + 1) / 2 + 64)) return rc;
  if (int rc = dO.alloc((size_t)(N * M + 1) / 2 + 64)) return rc;
  float* Kd = reinterpret_cast<float*>(dK.p);
  float* Od = reinterpret_cast<float*>(dO.p);
  SGP_HIP(hipMemsetAsync(Kd, 0, sizeof(float) * N * M, s));
  if (int rc = assemble_f32(ds, Kd, N, ds->symmetric, -1, 0.0f, nullptr, s)) return rc;
  hipLaunchKernelGGL(mirror_cast_f32_kernel, dim3((unsigned)((N * M + 255) / 256)), dim3(256), 0, s, Kd, N, N, M,
                     ds->symmetric, Od, N);
  SGP_HIP(hipGetLastError());
  SGP_HIP(hipStreamSynchronize(s));
  SGP_HIP(hipMemcpy2D(K, sizeof(float) * ldk, Od, sizeof(float) * N, sizeof(float) * N, (size_t)M, hipMemcpyDeviceToHost));
  return 0;
}

// ---------------------------------------------------------------------------------------------------------
// rand and posterior moments on the fp32 factor (round 3; the reference's type-stability test covers rand:
// /root/reference/test/gp/util.jl:76-88)
// ---------------------------------------------------------------------------------------------------------
namespace {

// per bordered row r < nrows:  dot[r] = sum_c R[r, c] z[c],  ssq[r] = sum_c R[r, c]^2  (fp32 storage, fp64 sums;
// R and the z row live in the same bordered matrix: z[c] = zrow[c * ld]).  32 rows x 8 column lanes per workgroup:
// every wave load is 32 consecutive rows of one column.
__global__ __launch_bounds__(256) void rows_stats_f32_kernel(const float* R, long ld, long nrows, long nc, const float* zrow,
                                                             double* dot, double* ssq) {
  __shared__ double sh[2][8][33];
  const int r = threadIdx.x & 31, kq = threadIdx.x >> 5;
  const long j = (long)blockIdx.x * 32 + r;
  double a = 0.0, b = 0.0;
  if (j < nrows)
    for (long c = kq; c < nc; c += 8) {
      const double v = (double)R[j + c * ld];
      a = fma(v, (double)zrow[c * ld], a);
      b = fma(v, v, b);
    }
  sh[0][kq][r] = a;
  sh[1][kq][r] = b;
  __syncthreads();
  if (kq == 0 && j < nrows) {
    double ta = 0.0, tb = 0.0;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      ta += sh[0][q][r];
      tb += sh[1][q][r];
    }
    dot[j] = ta;
    ssq[j] = tb;
  }
}

struct SpecG {
  sgp_dspec* d = nullptr;
  ~SpecG() {
    if (d) sgp_dspec_free_nolock(d);
  }
};

// K + Sigma_y (fp32, lower tiles, identity padding) into the zero-initialised m_tot x n_pad matrix A
int build_f32(sgp_ctx* ctx, const sgp_dspec* ds, float* A, long n_pad, long m_tot, int noise_kind, const double* noise,
              const double* d_noise, hipStream_t s) {
  const long N = ds->N;
  SGP_HIP(hipMemsetAsync(A, 0, sizeof(float) * m_tot * n_pad, s));
  if (int rc = assemble_f32(ds, A, m_tot, 1, noise_kind, noise_kind == SGP_NOISE_SCALAR ? (float)noise[0] : 0.0f, d_noise, s))
    return rc;
  const long h = n_pad - N;
  if (h > 0) {
    const long tot = h * m_tot;
    hipLaunchKernelGGL(fill_pad_f32_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, A, m_tot, N, n_pad, m_tot);
    SGP_HIP(hipGetLastError());
  }
  return 0;
}

int posdef_info(sgp_ctx* ctx, hipStream_t s) {
  int info = 0;
  SGP_HIP(hipMemcpyAsync(&info, ctx->d_info, sizeof(int), hipMemcpyDeviceToHost, s));
  SGP_HIP(hipStreamSynchronize(s));
  if (info > 0)
    set_error("matrix is not positive definite (fp32); Cholesky factorization failed at leading minor " +
              std::to_string(info));
  return info;
}

}  // namespace

// rand(rng, fx, S) in single precision: out = mean .+ L Z with L the fp32 factor (AbstractGPs rand [EXT], App. A.4).
// Z (N x S doubles, the caller's draw) is rounded to fp32.  The strictly upper 128-tiles of the factor buffer are zero
// (never written) and potrf_diag zeroes the upper part of the diagonal tiles, so L Z is one plain fp32 MFMA product.
extern "C" int sgp_rand_f32(sgp_ctx* ctx, const sgp_cov_spec* spec, const double* mean, int noise_kind, const double* noise,
                            const double* Z, int64_t ldz, int64_t S, float* out, int64_t ldo) {
  F_CHECK_ARG(ctx && spec && noise && Z && out, "sgp_rand_f32: NULL argument");
  F_CHECK_ARG(spec->symmetric, "sgp_rand_f32: spec must be symmetric");
  F_CHECK_ARG(noise_kind == SGP_NOISE_SCALAR || noise_kind == SGP_NOISE_DIAG, "sgp_rand_f32: noise kind must be SCALAR or DIAG");
  CtxScope scope(ctx);
  SpecG g;
  if (int rc = sgp_dspec_create_nolock(ctx, spec, &g.d)) return rc;
  const long N = g.d->N;
  F_CHECK_ARG(N >= 1 && S >= 1 && ldz >= N && ldo >= N, "sgp_rand_f32: bad sizes");
  int64_t n_pad, m_tot;
  sgp_geometry(N, 0, &n_pad, &m_tot);
  F_CHECK_ARG(n_pad / TILE <= ctx->n_slots, "matrix too large for the logdet slot buffer");
  const long s_pad = (S + TILE - 1) / TILE * TILE;
  hipStream_t s = ctx->stream;
  DevBuf dA, dn, dZt, dC;
  if (int rc = dA.alloc((size_t)(m_tot * n_pad + 1) / 2 + 64)) return rc;
  if (int rc = dZt.alloc((size_t)(s_pad * n_pad + 1) / 2 + 64)) return rc;
  if (int rc = dC.alloc((size_t)(n_pad * s_pad + 1) / 2 + 64)) return rc;
  if (noise_kind == SGP_NOISE_DIAG)
    if (int rc = dn.upload(noise, N)) return rc;
  float* A = reinterpret_cast<float*>(dA.p);
  float* Zt = reinterpret_cast<float*>(dZt.p);
  float* Cm = reinterpret_cast<float*>(dC.p);
  // host staging: Zt[s + k * s_pad] = -Z[k, s] (the product kernel computes C -= A B'), C[i + s * n_pad] = mean[i]
  std::vector<float> hz((size_t)s_pad * n_pad, 0.0f), hc((size_t)n_pad * s_pad, 0.0f);
  for (long sc = 0; sc < S; ++sc)
    for (long k = 0; k < N; ++k) {
      hz[(size_t)sc + (size_t)k * s_pad] = -(float)Z[k + sc * ldz];
      hc[(size_t)k + (size_t)sc * n_pad] = mean ? (float)mean[k] : 0.0f;
    }
  SGP_HIP(hipMemcpyAsync(Zt, hz.data(), sizeof(float) * hz.size(), hipMemcpyHostToDevice, s));
  SGP_HIP(hipMemcpyAsync(Cm, hc.data(), sizeof(float) * hc.size(), hipMemcpyHostToDevice, s));
  SGP_HIP(hipMemsetAsync(ctx->d_info, 0, sizeof(int), s));
  if (int rc = build_f32(ctx, g.d, A, n_pad, m_tot, noise_kind, noise, dn.p, s)) return rc;
  if (int rc = chol_f32(ctx, A, m_tot, n_pad, m_tot, s)) return rc;
  if (int rc = launch_gemm_f32(A, m_tot, Zt, s_pad, Cm, n_pad, n_pad, s_pad, n_pad, 0, s)) return rc;
  if (int info = posdef_info(ctx, s)) return info;
  SGP_HIP(hipMemcpy2D(out, sizeof(float) * ldo, Cm, sizeof(float) * n_pad, sizeof(float) * N, (size_t)S,
                      hipMemcpyDeviceToHost));
  return 0;
}

// posterior(fx, y) followed by mean_and_var at x*, in single precision and ONE factorisation: the test points ride
// through the fp32 Cholesky as bordered rows below the observation row (the trick the sharded fp64 posterior uses):
//   rows n_pad            : (y - m)'          ->  z' = (L^-1 (y - m))'
//   rows n_pad + 128 ...  : K(x*, x)          ->  V' = K(x*, x) L^-T
// mean* = m* + V' z, var* = diag K** - rowsumsq(V')  (AbstractGPs posterior + mean_and_var [EXT], App. A.5).
// cross: rows x*, columns x;  prior_ss: symmetric spec at x* (its diagonal is evaluated in fp64 and rounded).
extern "C" int sgp_posterior_mean_var_f32(sgp_ctx* ctx, const sgp_cov_spec* spec, const double* mean, int noise_kind,
                                          const double* noise, const double* y, const sgp_cov_spec* cross,
                                          const sgp_cov_spec* prior_ss, const double* mean_s, float* mean_out,
                                          float* var_out) {
  F_CHECK_ARG(ctx && spec && noise && y && cross, "sgp_posterior_mean_var_f32: NULL argument");
  F_CHECK_ARG(spec->symmetric && !cross->symmetric, "sgp_posterior_mean_var_f32: spec symmetric, cross not");
  F_CHECK_ARG(noise_kind == SGP_NOISE_SCALAR || noise_kind == SGP_NOISE_DIAG,
              "sgp_posterior_mean_var_f32: noise kind must be SCALAR or DIAG");
  F_CHECK_ARG(!var_out || prior_ss, "sgp_posterior_mean_var_f32: prior_ss spec required for var");
  CtxScope scope(ctx);
  SpecG g, gc, gp;
  if (int rc = sgp_dspec_create_nolock(ctx, spec, &g.d)) return rc;
  if (int rc = sgp_dspec_create_nolock(ctx, cross, &gc.d)) return rc;
  if (prior_ss)
    if (int rc = sgp_dspec_create_nolock(ctx, prior_ss, &gp.d)) return rc;
  const long N = g.d->N, Ns = gc.d->N;
  F_CHECK_ARG(N >= 1 && gc.d->M == N, "sgp_posterior_mean_var_f32: cross spec columns != training size");
  F_CHECK_ARG(!gp.d || gp.d->N == Ns, "sgp_posterior_mean_var_f32: prior_ss size != number of x*");
  if (Ns == 0) return 0;
  int64_t n_pad, mt1;
  sgp_geometry(N, 1, &n_pad, &mt1);
  F_CHECK_ARG(n_pad / TILE <= ctx->n_slots, "matrix too large for the logdet slot buffer");
  const long ns_pad = (Ns + TILE - 1) / TILE * TILE;
  const long m_tot = mt1 + ns_pad, row0 = mt1;
  hipStream_t s = ctx->stream;
  DevBuf dA, dy, dm, dn, dres;
  if (int rc = dA.alloc((size_t)(m_tot * n_pad + 1) / 2 + 64)) return rc;
  if (int rc = dy.upload(y, N)) return rc;
  if (mean)
    if (int rc = dm.upload(mean, N)) return rc;
  if (noise_kind == SGP_NOISE_DIAG)
    if (int rc = dn.upload(noise, N)) return rc;
  if (int rc = dres.alloc((size_t)3 * ns_pad)) return rc;
  float* A = reinterpret_cast<float*>(dA.p);
  SGP_HIP(hipMemsetAsync(ctx->d_info, 0, sizeof(int), s));
  if (int rc = build_f32(ctx, g.d, A, n_pad, m_tot, noise_kind, noise, dn.p, s)) return rc;
  {
    const long totb = (long)TILE * n_pad;
    hipLaunchKernelGGL(border_f32_kernel, dim3((unsigned)((totb + 255) / 256)), dim3(256), 0, s, A, m_tot, (long)n_pad, N,
                       dy.p, mean ? dm.p : nullptr);
    SGP_HIP(hipGetLastError());
  }
  // K(x*, x): element (r, c) at A[row0 + r + c * m_tot]
  if (int rc = assemble_f32(gc.d, A + row0, m_tot, 0, -1, 0.0f, nullptr, s)) return rc;
  if (int rc = chol_f32(ctx, A, m_tot, n_pad, m_tot, s)) return rc;
  double* d_dot = dres.p;
  double* d_ssq = dres.p + ns_pad;
  double* d_prior = dres.p + 2 * ns_pad;
  hipLaunchKernelGGL(rows_stats_f32_kernel, dim3((unsigned)((Ns + 31) / 32)), dim3(256), 0, s, A + row0, m_tot, Ns, N,
                     A + n_pad, d_dot, d_ssq);
  SGP_HIP(hipGetLastError());
  if (var_out)
    if (int rc = drv_diag_of_spec(ctx, gp.d, d_prior, s)) return rc;
  if (int info = posdef_info(ctx, s)) return info;
  std::vector<double> h((size_t)3 * ns_pad);
  SGP_HIP(hipMemcpy(h.data(), dres.p, sizeof(double) * 3 * ns_pad, hipMemcpyDeviceToHost));
  for (long r = 0; r < Ns; ++r) {
    if (mean_out) mean_out[r] = (float)((mean_s ? mean_s[r] : 0.0) + h[r]);
    if (var_out) var_out[r] = (float)(h[2 * ns_pad + r] - h[ns_pad + r]);
  }
  return 0;
}
