// Structural zeros (common.h, capi.hip: sz_pattern): the tile-level pattern of the Cholesky factor of K + Sigma_y, derived
// from the block-pair table of a symmetric spec.  Host-only, integer work that must be exact: compiled for the host with
// g++ and checked against brute force and against numerical factorisations in tests/sz_pattern_host.cpp.
//   block level : bnz[I * nb + J] != 0: block pair (I, J) has terms (the caller closes it symmetrically)
//   tile level  : tile (ti, tj) of the matrix is non-zero when some block pair it overlaps has terms (block boundaries
//                 need not sit on tile boundaries); diagonal tiles always (noise, identity padding); rows T_c .. T_r - 1
//                 (the bordered rows: y - m, extra right-hand sides) are dense
//   factor      : symbolic factorisation, column by column -- tile (i, j) fills in when rows i and j share a non-zero
//                 tile in an earlier column
// nz: T_r rows of `words` 64-bit words, bit k of row i = tile (i, k) of the factor may be non-zero.  executed / dense: the
// 128-column tile products of the contractions sum_{k < j} L_ik L_jk' over the non-zero tiles / over all tiles.
#pragma once
#include <algorithm>
#include <vector>

namespace sgp {

typedef unsigned long long sz_pattern_word;

struct SzPattern {
  std::vector<sz_pattern_word> nz;
  int words = 0;
  double executed = 0, dense = 0;
  bool zeros_left = false;   // false: the factor is structurally dense (nothing to skip)
  std::vector<double> col_work;   // executed tile products of tile column j (sum over the rows i >= j): sums to `executed`
};

// border_identity (round 5, the gradient path: capi.hip logpdf_grad_core): the bordered rows are [one dense tile row (the
// observation row) ; T_c tile rows that start as the IDENTITY] -- row T_c + 1 + q holds a single non-zero tile, at column
// q, and what the factorisation turns it into is inv(L)': tile (q, k) of it fills in by the same rule as any other row
// (the pattern of the inverse factor is the closure of the factor's under elimination).  T_r must be 2 T_c + 1.  `dense`
// then counts the same bordered matrix with every block pair coupled (what SGP_STRUCT_ZEROS=0 executes).
inline void sz_symbolic(const std::vector<char>& bnz, int nb, const std::vector<long>& off, const std::vector<long>& len,
                        long N, long tile, long T_c, long T_r, SzPattern& out, bool border_identity = false) {
  typedef sz_pattern_word word;
  const int W = (int)((T_c + 63) / 64);
  out.words = W;
  out.nz.assign((size_t)T_r * W, 0);
  out.dense = 0;
  for (long j = 0; j < T_c; ++j) out.dense += (double)j * (double)(T_r - j);
  // blocks a tile overlaps: [blo, bhi]
  std::vector<int> blo(T_c, 0), bhi(T_c, -1);
  for (long t = 0; t < T_c; ++t) {
    const long p0 = t * tile, p1 = std::min<long>(p0 + tile, N);
    int lo = nb, hi = -1;
    for (int I = 0; I < nb; ++I) {
      if (len[I] <= 0) continue;
      if (off[I] < p1 && off[I] + len[I] > p0) {
        lo = std::min(lo, I);
        hi = std::max(hi, I);
      }
    }
    blo[t] = lo;
    bhi[t] = hi;
  }
  std::vector<word>& nz = out.nz;
  auto setbit = [&](long i, long k) { nz[(size_t)i * W + (k >> 6)] |= (word)1 << (k & 63); };
  auto getbit = [&](long i, long k) { return (nz[(size_t)i * W + (k >> 6)] >> (k & 63)) & 1; };
  for (long i = 0; i < T_c; ++i) {
    setbit(i, i);
    for (long k = 0; k < i; ++k) {
      bool on = false;
      for (int I = blo[i]; I <= bhi[i] && !on; ++I) {
        if (len[I] <= 0) continue;   // (an empty block between two blocks the tile overlaps)
        for (int J = blo[k]; J <= bhi[k] && !on; ++J) on = len[J] > 0 && bnz[(size_t)I * nb + J] != 0;
      }
      if (on) setbit(i, k);
    }
  }
  for (long i = T_c; i < T_r; ++i) {
    if (border_identity && i > T_c) {
      if (i - T_c - 1 < T_c) setbit(i, i - T_c - 1);
      continue;
    }
    for (long k = 0; k < T_c; ++k) setbit(i, k);
  }
  // fill-in, and the work of the contractions
  out.executed = 0;
  out.zeros_left = false;
  out.col_work.assign((size_t)T_c, 0.0);
  for (long j = 0; j < T_c; ++j) {
    const word* rj = &nz[(size_t)j * W];
    for (long i = j; i < T_r; ++i) {
      word* ri = &nz[(size_t)i * W];
      long shared = 0;
      if (j > 0) {
        const int qlast = (int)((j - 1) >> 6);
        for (int q = 0; q <= qlast; ++q) {
          word m = ri[q] & rj[q];
          if (q == qlast && (j & 63)) m &= ~(~(word)0 << (j & 63));
          shared += __builtin_popcountll(m);
        }
      }
      if (shared > 0 && !getbit(i, j)) setbit(i, j);
      if (getbit(i, j)) {
        out.executed += (double)shared;
        out.col_work[(size_t)j] += (double)shared;
      } else if (!(border_identity && i > T_c && i - T_c - 1 > j))   // (left of an identity row's own tile: zero in ANY model)
        out.zeros_left = true;
    }
  }
  if (border_identity) {   // the dense count of this bordered shape: the same elimination with every block pair coupled
    bool all = true;
    for (char c : bnz) all = all && c != 0;
    if (all) {
      out.dense = out.executed;
    } else {
      SzPattern full;
      sz_symbolic(std::vector<char>(bnz.size(), 1), nb, off, len, N, tile, T_c, T_r, full, true);
      out.dense = full.executed;
    }
  }
}

}  // namespace sgp
