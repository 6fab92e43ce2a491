// Workgroup id -> tile (tr, tc) of the 128 x 128 tile grid of a GEMM launch (gemm_nt.hip), as plain integer
// functions that also compile for the host: tests/tilemap_host.cpp enumerates whole launches on the CPU and checks
// that every live tile is produced exactly once, that nothing else is, and that the XCDs get equal shares.
#pragma once
#ifndef __HIPCC__
#include <cmath>
#define __host__
#define __device__
#define __forceinline__ inline
#endif

namespace sgp {

// Workgroup -> tile enumeration.  Hardware places workgroup id on XCD id % 8; XCD x owns one tile row of every 8
// (row 8 j + x for even j, 8 j + 7 - x for odd j: boustrophedon, so that the live-tile counts under the triangular
// mask are equal), and k = id / 8 walks that XCD's tiles.  Rectangular launches enumerate (8 owned rows) x n_tc per
// group of 8 owned rows.  Lower-triangular launches (mask_off == 0) enumerate ONLY live tiles: dead workgroups above
// the diagonal cost dispatcher time (a 128^2-tile lower update enumerated as a rectangle lost 20 % to them).  Per XCD:
//   A  groups G < Ga (tile rows 64 G .. 64 G + 63, diagonal 64-block complete): 512 G tiles strictly left of the
//      block + exactly 260 live tiles inside it (independent of the XCD thanks to the boustrophedon ownership);
//   D  the group the matrix edge cuts through -- fewer than 64 tile columns left for its diagonal block (c), or fewer
//      than 8 owned rows (R): R x 64 G tiles left of the block, then row by row min(8 jj + off + 1, c) tiles inside
//      it.  That count depends on the XCD by a few tiles, so every XCD walks its OWN count (its later segments start
//      where its own D ends); the launch is sized for the largest XCD and the others' last few ids return dead.  (The
//      sizes of the trailing updates are multiples of 4 or 8 tiles, not of 64: as a rectangle this group made a third
//      of all ids of an N = 16384 factorisation dead.)
//   B  groups below the last tile column (every tile live): rectangles of 8 (last: r_last) owned rows x n_tc.
// (32-bit integers throughout: a launch has fewer than 2^31 workgroups, and 64-bit integer division -- which the
// enumeration needs in a few places -- is a ~100-instruction software routine on the GPU: the id -> tile mapping cost
// every workgroup ~4000 cycles of its prologue with `long` arithmetic, round 3 stamps)
typedef int tm_int;
struct TriShape {
  tm_int Ga, sA;            // A: number of groups, ids
  tm_int Rd, cd, sDl, sD;   // D: owned rows, columns of its diagonal block, ids left of the block, all ids (0: none)
  tm_int nB, sB, r_last;    // B: full groups, their ids, owned rows of the partial last group
  tm_int n_tc;
};
__host__ __device__ __forceinline__ tm_int tri_diag_row_count(tm_int jj, tm_int xcd, tm_int c) {
  const tm_int cnt = 8 * jj + ((jj & 1) ? 7 - xcd : xcd) + 1;
  return cnt < c ? cnt : c;
}
// xcd in 0..7: the shape as that XCD walks it; xcd < 0: the largest over the XCDs (grid sizing on the host).
__host__ __device__ __forceinline__ TriShape tri_shape(long n_tr_, long n_tc_, long xcd_) {
  TriShape t;
  const tm_int n_tr = (tm_int)n_tr_, n_tc = (tm_int)n_tc_, xcd = (tm_int)xcd_;
  const tm_int J = (n_tr + 7) / 8;  // owned rows per XCD
  const tm_int Gn = J / 8, r_last = J % 8, g_full = n_tc / 64;
  t.n_tc = n_tc;
  t.Ga = Gn < g_full ? Gn : g_full;
  t.sA = 256 * t.Ga * (t.Ga - 1) + 260 * t.Ga;
  // the next group: complete rows (if any full group is left) or the partial last one
  tm_int g = t.Ga;
  const tm_int Rg = g < Gn ? 8 : r_last;
  const tm_int cg = n_tc - 64 * g < 64 ? n_tc - 64 * g : 64;
  t.Rd = 0, t.cd = 0, t.sDl = 0, t.sD = 0;
  if (Rg > 0 && cg > 0) {
    t.Rd = Rg;
    t.cd = cg;
    t.sDl = Rg * 64 * g;
    tm_int tsum = 0;
    if (xcd >= 0) {
      for (tm_int jj = 0; jj < Rg; ++jj) tsum += tri_diag_row_count(jj, xcd, cg);
    } else {
      for (tm_int x = 0; x < 8; ++x) {
        tm_int tx = 0;
        for (tm_int jj = 0; jj < Rg; ++jj) tx += tri_diag_row_count(jj, x, cg);
        tsum = tx > tsum ? tx : tsum;
      }
    }
    t.sD = t.sDl + tsum;
    ++g;
  }
  // what is left lies entirely below the last tile column
  t.nB = g < Gn ? Gn - g : 0;
  t.sB = t.nB * 8 * n_tc;
  t.r_last = g <= Gn ? r_last : 0;   // g == Gn + 1: the partial group was D
  return t;
}
__host__ __device__ __forceinline__ long tri_ids_per_xcd(const TriShape& t) {
  return (long)t.sA + t.sD + t.sB + t.r_last * t.n_tc;
}

__host__ __device__ __forceinline__ bool tile_of_id(long id_, long n_tr_, long n_tc_, long mask_off, long& tr, long& tc) {
  const tm_int id = (tm_int)id_, n_tr = (tm_int)n_tr_, n_tc = (tm_int)n_tc_;
  const tm_int xcd = id & 7, k = id >> 3;
  tm_int j, c;
  if (mask_off == 0) {
    const TriShape t = tri_shape(n_tr, n_tc, xcd);
    if (k < t.sA) {
      // S(G) = 256 G (G - 1) + 260 G = 256 G^2 + 4 G
      tm_int G = (tm_int)((sqrtf(16.0f + 1024.0f * (float)k) - 4.0f) * (1.0f / 512.0f));
      while (256 * G * G + 4 * G > k) --G;
      while (256 * (G + 1) * (G + 1) + 4 * (G + 1) <= k) ++G;
      const tm_int within = k - (256 * G * G + 4 * G);
      if (within < 512 * G) {
        j = G * 8 + (within & 7);
        c = within >> 3;
      } else {
        tm_int d = within - 512 * G, jj = 0, cum = 0;
        for (; jj < 8; ++jj) {
          const tm_int cnt = 8 * jj + ((jj & 1) ? 7 - xcd : xcd) + 1;
          if (d < cum + cnt) break;
          cum += cnt;
        }
        j = G * 8 + jj;
        c = 64 * G + (d - cum);
      }
    } else if (k < t.sA + t.sD) {
      const tm_int kk = k - t.sA;
      if (kk < t.sDl) {
        const tm_int q = (tm_int)((unsigned)kk / (unsigned)t.Rd);
        j = t.Ga * 8 + (kk - q * t.Rd);
        c = q;
      } else {
        tm_int d = kk - t.sDl, jj = 0, cum = 0;
        for (; jj < t.Rd; ++jj) {
          const tm_int cnt = tri_diag_row_count(jj, xcd, t.cd);
          if (d < cum + cnt) break;
          cum += cnt;
        }
        if (jj == t.Rd) return false;   // this XCD's share of the block is smaller than the uniform grid's
        j = t.Ga * 8 + jj;
        c = 64 * t.Ga + (d - cum);
      }
    } else if (k < t.sA + t.sD + t.sB) {
      const tm_int kk = k - t.sA - t.sD;
      const tm_int q = (tm_int)((unsigned)kk / (unsigned)(8 * n_tc)), within = kk - q * 8 * n_tc;
      j = (t.Ga + (t.sD > 0 ? 1 : 0) + q) * 8 + (within & 7);
      c = within >> 3;
    } else {  // partial last group: r_last owned rows, every tile column
      if (t.r_last == 0) return false;  // past this XCD's last tile (the launch is sized for the largest XCD)
      const tm_int within = k - t.sA - t.sD - t.sB;
      const tm_int q = (tm_int)((unsigned)within / (unsigned)t.r_last);
      j = (t.Ga + (t.sD > 0 ? 1 : 0) + t.nB) * 8 + (within - q * t.r_last);
      c = q;
    }
  } else {
    const tm_int gs = 8 * n_tc;
    const tm_int q = (tm_int)((unsigned)k / (unsigned)gs), r = k - q * gs;
    j = q * 8 + (r & 7);
    c = r >> 3;
  }
  const tm_int row = 8 * j + ((j & 1) ? 7 - xcd : xcd);  // boustrophedon: equal live-tile counts per XCD
  tr = row;
  tc = c;
  return row < n_tr && c < n_tc && (long)row >= (long)c + mask_off;
}

// ids per launch (what the launchers put into the grid)
__host__ __device__ __forceinline__ long tile_ids(long n_tr, long n_tc, long mask_off) {
  const long groups = ((n_tr + 7) / 8 + 7) / 8;  // groups of 8 owned rows per XCD
  const long per_xcd = (mask_off == 0) ? tri_ids_per_xcd(tri_shape(n_tr, n_tc, -1)) : groups * 8 * n_tc;
  return per_xcd * 8;
}

}  // namespace sgp
