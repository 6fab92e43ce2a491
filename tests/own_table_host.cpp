// Host-side check of the panel ownership table of the sharded factorisation (stheno.jl_amd/csrc/own_table.h; compiled by
// g++ in tests/test_own_table_host.py -- no GPU involved).
//   * every round of P consecutive panels gives every rank at most one panel (the property the schedule relies on);
//   * equal costs give the cyclic deal; random costs never come out worse than the cyclic deal's largest load;
//   * the north-star model (f3 = f1 + f2, blocks 21846 / 21845 / 21845, N = 65536, W = 1024, 8 ranks): the loads the table
//     produces from the symbolic pattern are within +-3 % of their mean, the cyclic deal's are not (printed: the figures
//     DESIGN.md quotes).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../stheno.jl_amd/csrc/sz_pattern.h"
#include "../stheno.jl_amd/csrc/own_table.h"

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

static unsigned long long rng_state = 1234567891234567ULL;
static unsigned long long rnd() {
  rng_state ^= rng_state << 13;
  rng_state ^= rng_state >> 7;
  rng_state ^= rng_state << 17;
  return rng_state;
}
static double urand() { return (double)(rnd() >> 11) / 9007199254740992.0; }

static std::vector<double> loads(const std::vector<double>& cost, const std::vector<int>& own, int P) {
  std::vector<double> l((size_t)P, 0.0);
  for (size_t J = 0; J < own.size(); ++J) l[(size_t)own[J]] += cost[J];
  return l;
}

static void rounds_ok(const std::vector<int>& own, int P, const char* what) {
  const long npan = (long)own.size();
  for (long r0 = 0; r0 < npan; r0 += P) {
    std::vector<int> seen((size_t)P, 0);
    for (long J = r0; J < std::min<long>(npan, r0 + P); ++J) {
      CHECK(own[(size_t)J] >= 0 && own[(size_t)J] < P, "%s: owner out of range", what);
      if (own[(size_t)J] >= 0 && own[(size_t)J] < P) seen[(size_t)own[(size_t)J]] += 1;
    }
    for (int i = 0; i < P; ++i) CHECK(seen[(size_t)i] <= 1, "%s: rank %d twice in the round at panel %ld", what, i, r0);
  }
}

int main() {
  long cases = 0;
  // ---- properties on random cost vectors
  for (int trial = 0; trial < 2000; ++trial) {
    const int P = 1 + (int)(rnd() % 9);
    const long npan = 1 + (long)(rnd() % 80);
    std::vector<double> cost((size_t)npan);
    const int shape = (int)(rnd() % 3);
    for (long J = 0; J < npan; ++J)
      cost[(size_t)J] = shape == 0 ? 1.0 : shape == 1 ? 0.1 + urand() : (double)(J + 1) * (double)(npan - J) * (0.5 + urand());
    std::vector<int> own = sgp::balanced_owners(cost, P);
    CHECK((long)own.size() == npan, "size");
    rounds_ok(own, P, "random");
    std::vector<int> cyc((size_t)npan);
    for (long J = 0; J < npan; ++J) cyc[(size_t)J] = (int)(J % P);
    if (shape == 0) CHECK(own == cyc, "equal costs must give the cyclic deal (P %d, %ld panels)", P, npan);
    CHECK(sgp::max_load(cost, own, P) <= sgp::max_load(cost, cyc, P) * (1.0 + 1e-12), "worse than the cyclic deal (P %d, %ld panels, shape %d)", P,
          npan, shape);
    CHECK(sgp::owners_valid(own, P), "valid");
    ++cases;
  }
  // ---- the north-star model and the dense model at N = 65536, W = 1024, 8 ranks
  const long tile = 128, N = 65536, T_c = N / tile, T_r = T_c + 1, W = 1024;
  std::vector<long> c0s;
  for (long c = 0; c <= N; c += W) c0s.push_back(c);
  for (int model = 0; model < 2; ++model) {
    std::vector<double> col_work;
    if (model == 1) {
      const int nb = 3;
      std::vector<long> len = {21846, 21845, 21845}, off = {0, 21846, 43691};
      std::vector<char> bnz = {1, 0, 1, 0, 1, 1, 1, 1, 1};   // f1 | f2 independent, f3 = f1 + f2
      sgp::SzPattern pat;
      sgp::sz_symbolic(bnz, nb, off, len, N, tile, T_c, T_r, pat);
      CHECK(pat.zeros_left, "the north-star model has structural zeros");
      double s = 0;
      for (double v : pat.col_work) s += v;
      CHECK(std::fabs(s - pat.executed) <= 1e-9 * pat.executed, "col_work must sum to executed");
      col_work = pat.col_work;
      std::printf("north-star model: executed / dense tile products %.4f\n", pat.executed / pat.dense);
    }
    std::vector<double> cost = sgp::panel_costs(c0s, tile, T_r, col_work);
    for (int P : {2, 4, 8}) {
      std::vector<int> own = sgp::balanced_owners(cost, P), cyc(own.size());
      for (size_t J = 0; J < cyc.size(); ++J) cyc[J] = (int)(J % (size_t)P);
      rounds_ok(own, P, "model");
      std::vector<double> lb = loads(cost, own, P), lc = loads(cost, cyc, P);
      double mean = 0;
      for (double v : lb) mean += v / P;
      double bmin = 1e300, bmax = 0, cmin = 1e300, cmax = 0;
      for (int i = 0; i < P; ++i) {
        bmin = std::min(bmin, lb[(size_t)i]);
        bmax = std::max(bmax, lb[(size_t)i]);
        cmin = std::min(cmin, lc[(size_t)i]);
        cmax = std::max(cmax, lc[(size_t)i]);
      }
      std::printf("%s P=%d: model ms per rank: cyclic %.1f .. %.1f (%+.1f %% / %+.1f %% of the mean %.1f), table %.1f .. %.1f (%+.1f %% / %+.1f %%)\n",
                  model ? "north-star" : "dense", P, cmin, cmax, 100 * (cmin / mean - 1), 100 * (cmax / mean - 1), mean, bmin, bmax,
                  100 * (bmin / mean - 1), 100 * (bmax / mean - 1));
      CHECK(bmax <= cmax * (1 + 1e-12), "table worse than cyclic");
      CHECK(bmax <= 1.03 * mean && bmin >= 0.97 * mean, "table loads must be within 3 %% of the mean (P %d, model %d)", P, model);
      if (model == 1 && P == 8) CHECK(cmax > 1.05 * mean, "(the cyclic deal of the north-star model is expected to be off by more than 5 %%)");
      ++cases;
    }
  }
  std::printf("cases %ld failures %ld\n", cases, failures);
  return failures ? 1 : 0;
}
