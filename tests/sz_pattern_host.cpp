// Host-side check of the structural-zero pattern (stheno.jl_amd/csrc/sz_pattern.h; compiled by g++ in
// tests/test_sz_pattern_host.py -- no GPU involved).  Random programmes: processes as random combinations of independent
// atoms, random ragged block sizes, a small tile so that a few hundred points span many tiles.  For each:
//   * an INDEPENDENT brute-force statement (boolean matrices, the textbook elimination rule) must give the same pattern
//     and the same executed / dense tile-product counts as sz_symbolic's bit-mask form,
//   * a NUMERICAL check: a dense SPD matrix with exactly that block structure (sum over atoms of c_pa c_qa Phi_a Phi_a' +
//     identity: blocks of processes that share no atom are exact zeros) is factored by a plain triple-loop Cholesky; every
//     tile of L with a non-zero entry must be inside the pattern (the pattern may be larger: numerical cancellation), and
//     the factor computed with the skipped tile products LEFT OUT must equal the full one bit for bit.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../stheno.jl_amd/csrc/sz_pattern.h"

static long failures = 0;
#define CHECK(c, ...)                      \
  do {                                     \
    if (!(c)) {                            \
      if (failures < 20) {                 \
        std::printf("FAIL: " __VA_ARGS__); \
        std::printf("\n");                 \
      }                                    \
      ++failures;                          \
    }                                      \
  } while (0)

static unsigned long long rng_state = 88172645463325252ULL;
static unsigned long long rnd() {
  rng_state ^= rng_state << 13;
  rng_state ^= rng_state >> 7;
  rng_state ^= rng_state << 17;
  return rng_state;
}
static double urand() { return (double)(rnd() >> 11) / 9007199254740992.0; }

int main() {
  long cases = 0, with_zeros = 0, tiles_skipped = 0;
  for (int trial = 0; trial < 400; ++trial) {
    const int nb = 2 + (int)(rnd() % 5), na = 1 + (int)(rnd() % 4);
    const long tile = 8;
    std::vector<long> len(nb), off(nb);
    long N = 0;
    for (int I = 0; I < nb; ++I) {
      len[I] = (rnd() % 7 == 0) ? 0 : 3 + (long)(rnd() % 40);
      off[I] = N;
      N += len[I];
    }
    if (N < 2 * tile) continue;
    // process I uses atom a with coefficient coef[I][a] (possibly 0); every process uses at least one atom
    std::vector<double> coef((size_t)nb * na, 0.0);
    for (int I = 0; I < nb; ++I) {
      bool any = false;
      for (int a = 0; a < na; ++a)
        if (rnd() % 3 == 0) {
          coef[(size_t)I * na + a] = 0.5 + urand();
          any = true;
        }
      if (!any) coef[(size_t)I * na + (int)(rnd() % na)] = 1.0;
    }
    std::vector<char> bnz((size_t)nb * nb, 0);
    for (int I = 0; I < nb; ++I)
      for (int J = 0; J < nb; ++J)
        for (int a = 0; a < na; ++a)
          if (coef[(size_t)I * na + a] != 0.0 && coef[(size_t)J * na + a] != 0.0) bnz[(size_t)I * nb + J] = 1;
    const long T_c = (N + tile - 1) / tile, n_pad = T_c * tile;
    const long T_r = T_c + (long)(rnd() % 3);   // 0 .. 2 bordered tile rows
    sgp::SzPattern pat;
    sgp::sz_symbolic(bnz, nb, off, len, N, tile, T_c, T_r, pat);
    ++cases;
    // ---- brute force: boolean matrices
    std::vector<int> blk(n_pad, -1);
    for (int I = 0; I < nb; ++I)
      for (long p = off[I]; p < off[I] + len[I]; ++p) blk[p] = I;
    std::vector<char> B((size_t)T_r * T_c, 0);
    for (long i = 0; i < T_c; ++i)
      for (long k = 0; k <= i; ++k) {
        bool on = i == k;
        for (long p = i * tile; p < std::min(N, (i + 1) * tile) && !on; ++p)
          for (long q = k * tile; q < std::min(N, (k + 1) * tile) && !on; ++q) on = bnz[(size_t)blk[p] * nb + blk[q]] != 0;
        B[(size_t)i * T_c + k] = on;
      }
    for (long i = T_c; i < T_r; ++i)
      for (long k = 0; k < T_c; ++k) B[(size_t)i * T_c + k] = 1;
    double ex = 0, de = 0;
    bool zeros = false;
    for (long j = 0; j < T_c; ++j)
      for (long i = j; i < T_r; ++i) {
        long shared = 0;
        for (long k = 0; k < j; ++k) shared += (B[(size_t)i * T_c + k] && B[(size_t)j * T_c + k]) ? 1 : 0;
        if (shared) B[(size_t)i * T_c + j] = 1;
        if (B[(size_t)i * T_c + j]) ex += (double)shared;
        else zeros = true;
        de += (double)j;
      }
    CHECK(ex == pat.executed && de == pat.dense && zeros == pat.zeros_left, "trial %d: counts %g %g %d vs %g %g %d", trial,
          ex, de, (int)zeros, pat.executed, pat.dense, (int)pat.zeros_left);
    for (long i = 0; i < T_r; ++i)
      for (long k = 0; k < T_c; ++k) {
        const bool bit = (pat.nz[(size_t)i * pat.words + (k >> 6)] >> (k & 63)) & 1;
        const bool want = k <= i ? B[(size_t)i * T_c + k] != 0 : false;
        CHECK(bit == want, "trial %d: pattern bit (%ld, %ld) %d vs %d", trial, i, k, (int)bit, (int)want);
      }
    if (zeros) ++with_zeros;
    // ---- numerical: K = sum_a (c Phi_a)(c Phi_a)' + I on the n_pad points (padding: identity), two factorisations
    const int r = 3;
    std::vector<double> K((size_t)n_pad * n_pad, 0.0);
    for (int a = 0; a < na; ++a) {
      std::vector<double> Phi((size_t)n_pad * r, 0.0);
      for (long p = 0; p < N; ++p)
        for (int c = 0; c < r; ++c) Phi[(size_t)p * r + c] = coef[(size_t)blk[p] * na + a] * (urand() - 0.5);
      for (long p = 0; p < N; ++p)
        for (long q = 0; q < N; ++q) {
          double s = 0;
          for (int c = 0; c < r; ++c) s += Phi[(size_t)p * r + c] * Phi[(size_t)q * r + c];
          if (coef[(size_t)blk[p] * na + a] != 0.0 && coef[(size_t)blk[q] * na + a] != 0.0) K[(size_t)p * n_pad + q] += s;
        }
    }
    for (long p = 0; p < n_pad; ++p) K[(size_t)p * n_pad + p] += 1.0;
    auto factor = [&](bool skip, std::vector<double>& L) {
      L = K;
      // left-looking by tile column, the contraction by k tile (as the device does), k tiles skipped under the pattern
      for (long jt = 0; jt < T_c; ++jt)
        for (long j = jt * tile; j < (jt + 1) * tile; ++j) {
          for (long i = j; i < n_pad; ++i) {
            const long it = i / tile;
            double s = L[(size_t)i * n_pad + j];
            for (long kt = 0; kt <= jt; ++kt) {
              if (skip && kt < jt && !(B[(size_t)it * T_c + kt] && B[(size_t)jt * T_c + kt])) {
                ++tiles_skipped;
                continue;
              }
              for (long k = kt * tile; k < std::min(j, (kt + 1) * tile); ++k) s -= L[(size_t)i * n_pad + k] * L[(size_t)j * n_pad + k];
            }
            L[(size_t)i * n_pad + j] = (i == j) ? std::sqrt(s) : s / L[(size_t)j * n_pad + j];
          }
        }
    };
    std::vector<double> L0, L1;
    factor(false, L0);
    factor(true, L1);
    bool same = true;
    for (long i = 0; i < n_pad && same; ++i)
      for (long j = 0; j <= i; ++j)
        if (L0[(size_t)i * n_pad + j] != L1[(size_t)i * n_pad + j]) {
          same = false;
          break;
        }
    CHECK(same, "trial %d: the factor changes when the structurally dead tile products are left out", trial);
    for (long i = 0; i < n_pad; ++i)
      for (long j = 0; j <= i; ++j)
        if (L0[(size_t)i * n_pad + j] != 0.0)
          CHECK(B[(size_t)(i / tile) * T_c + j / tile], "trial %d: L(%ld, %ld) = %g lies outside the pattern", trial, i, j,
                L0[(size_t)i * n_pad + j]);
    // ---- round 5, the gradient path's border [one dense tile row ; identity rows] (sz_symbolic: border_identity): the
    // pattern of row T_c + 1 + q is that of tile row q of inv(L)' -- against brute-force elimination and against the
    // numerical inverse of the factor computed above
    {
      const long T_g = 2 * T_c + 1;
      sgp::SzPattern pg;
      sgp::sz_symbolic(bnz, nb, off, len, N, tile, T_c, T_g, pg, true);
      std::vector<char> G((size_t)T_g * T_c, 0);
      for (long i = 0; i < T_c; ++i)
        for (long k = 0; k <= i; ++k) {
          bool on = i == k;
          for (long p = i * tile; p < std::min(N, (i + 1) * tile) && !on; ++p)
            for (long q = k * tile; q < std::min(N, (k + 1) * tile) && !on; ++q) on = bnz[(size_t)blk[p] * nb + blk[q]] != 0;
          G[(size_t)i * T_c + k] = on;
        }
      for (long k = 0; k < T_c; ++k) G[(size_t)T_c * T_c + k] = 1;
      for (long q = 0; q < T_c; ++q) G[(size_t)(T_c + 1 + q) * T_c + q] = 1;
      double exg = 0;
      for (long j = 0; j < T_c; ++j)
        for (long i = j; i < T_g; ++i) {
          long shared = 0;
          for (long k = 0; k < j; ++k) shared += (G[(size_t)i * T_c + k] && G[(size_t)j * T_c + k]) ? 1 : 0;
          if (shared) G[(size_t)i * T_c + j] = 1;
          if (G[(size_t)i * T_c + j]) exg += (double)shared;
        }
      CHECK(exg == pg.executed, "trial %d: gradient-border executed count %g vs %g", trial, exg, pg.executed);
      CHECK(pg.dense >= pg.executed, "trial %d: gradient-border dense count below the executed one", trial);
      for (long i = 0; i < T_g; ++i)
        for (long k = 0; k < T_c; ++k) {
          const bool bit = (pg.nz[(size_t)i * pg.words + (k >> 6)] >> (k & 63)) & 1;
          const bool want = (i < T_c && k > i) ? false : G[(size_t)i * T_c + k] != 0;
          CHECK(bit == want, "trial %d: gradient-border pattern bit (%ld, %ld) %d vs %d", trial, i, k, (int)bit, (int)want);
        }
      // numerical: W = inv(L0) by forward substitution; W(k-rows, q-cols) != 0 must lie inside row T_c + 1 + q, column k
      std::vector<double> Wn((size_t)n_pad * n_pad, 0.0);
      for (long c = 0; c < n_pad; ++c)
        for (long i = c; i < n_pad; ++i) {
          double sacc = (i == c) ? 1.0 : 0.0;
          for (long k = c; k < i; ++k) sacc -= L0[(size_t)i * n_pad + k] * Wn[(size_t)k * n_pad + c];
          Wn[(size_t)i * n_pad + c] = sacc / L0[(size_t)i * n_pad + i];
        }
      for (long i = 0; i < n_pad; ++i)
        for (long c = 0; c <= i; ++c)
          if (Wn[(size_t)i * n_pad + c] != 0.0)
            CHECK(G[(size_t)(T_c + 1 + c / tile) * T_c + i / tile], "trial %d: inv(L)(%ld, %ld) = %g lies outside the identity rows' pattern",
                  trial, i, c, Wn[(size_t)i * n_pad + c]);
    }
  }
  std::printf("cases %ld with_zeros %ld failures %ld\n", cases, with_zeros, failures);
  return failures ? 1 : 0;
}
