// Who owns which column panel of the sharded factorisation (multi.hip).  Host-only integer / double work, compiled for the
// host with g++ and checked in tests/own_table_host.cpp.
//
// Round 4 dealt the panels out cyclically (owner = J % P).  That balances a DENSE factorisation (the work of panel J as a
// destination -- every tile product that lands in its columns -- varies smoothly with J), but not a programme with
// independent components: the structural zeros (sz_pattern.h) take whole ranges of k away from some panels and not from
// others, and with 8 ranks the cyclic deal left the per-rank update sums of the north-star model 96 .. 115 ms apart
// (profiles/r04_projection_target_P8_default.txt) -- the slowest rank sets the step.
//
// The table built here keeps what makes the cyclic deal a good SCHEDULE -- every round of P consecutive panels gives every
// rank exactly one panel, so all ranks have trailing work at every step and a panel's owner has had P - 1 steps to bring it
// up to date -- and chooses the permutation inside each round: rounds are taken in order, the round's panels sorted by
// cost (descending) are matched with the ranks sorted by accumulated load (ascending) -- the longest-processing-time rule
// applied round by round -- followed by pairwise swaps inside rounds while they lower the largest load.
// cost(J) = executed tile products of the panel's tile columns (SzPattern::col_work, or the dense count) priced at the
// update kernel's rate + the panel's own factorisation (launch-latency bound: a per-panel constant + a per-row term).
// Dense models come out within a fraction of a percent of the cyclic deal; with cost all equal the table IS the cyclic deal.
#pragma once
#include <algorithm>
#include <numeric>
#include <vector>

namespace sgp {

// per panel: milliseconds of work its owner spends on it over the whole factorisation (a model; only ratios matter).
// c0s: first column of every panel and n_pad (multiples of `tile`); col_work: per tile column, or empty = dense
inline std::vector<double> panel_costs(const std::vector<long>& c0s, long tile, long T_r, const std::vector<double>& col_work) {
  const long npan = (long)c0s.size() - 1;
  std::vector<double> cost((size_t)npan, 0.0);
  const double ms_per_product = 2.0 * (double)tile * (double)tile * (double)tile / 66.0e9;   // 66 TFLOP/s update launches
  for (long J = 0; J < npan; ++J) {
    double prod = 0;
    for (long j = c0s[J] / tile; j < c0s[J + 1] / tile; ++j)
      prod += col_work.empty() ? (double)j * (double)(T_r - j) : col_work[(size_t)j];
    const double rows = (double)(T_r * tile - c0s[J]), w = (double)(c0s[J + 1] - c0s[J]);
    // the panel's factorisation + its look-ahead update (measured, N = 65536, W = 1024: 0.25 ms + 2.9e-5 ms per row)
    const double fac = (0.25 + 2.9e-5 * rows) * (w / 1024.0);
    cost[(size_t)J] = prod * ms_per_product + fac;
  }
  return cost;
}

inline double max_load(const std::vector<double>& cost, const std::vector<int>& own, int P, double* min_out = nullptr) {
  std::vector<double> load((size_t)P, 0.0);
  for (size_t J = 0; J < own.size(); ++J) load[(size_t)own[J]] += cost[J];
  if (min_out) *min_out = *std::min_element(load.begin(), load.end());
  return *std::max_element(load.begin(), load.end());
}

// own[J] in [0, P): every round of P consecutive panels (the last one may be short) uses every rank at most once
inline std::vector<int> balanced_owners(const std::vector<double>& cost, int P) {
  const long npan = (long)cost.size();
  std::vector<int> own((size_t)npan, 0);
  if (P <= 1) return own;
  std::vector<double> load((size_t)P, 0.0);
  std::vector<int> ranks((size_t)P);
  std::vector<long> pan;
  for (long r0 = 0; r0 < npan; r0 += P) {
    const long r1 = std::min<long>(npan, r0 + P);
    pan.clear();
    for (long J = r0; J < r1; ++J) pan.push_back(J);
    // stable sorts: with equal costs and equal loads this is the cyclic deal
    std::stable_sort(pan.begin(), pan.end(), [&](long a, long b) { return cost[(size_t)a] > cost[(size_t)b]; });
    std::iota(ranks.begin(), ranks.end(), 0);
    if (r0 == 0) {
      // first round: nothing to balance against yet -- keep the cyclic deal (panel 0 on rank 0, as every caller expects)
      for (long J = r0; J < r1; ++J) own[(size_t)J] = (int)(J - r0);
    } else {
      std::stable_sort(ranks.begin(), ranks.end(), [&](int a, int b) { return load[(size_t)a] < load[(size_t)b]; });
      for (size_t q = 0; q < pan.size(); ++q) own[(size_t)pan[q]] = ranks[q];
    }
    for (long J = r0; J < r1; ++J) load[(size_t)own[(size_t)J]] += cost[(size_t)J];
  }
  // refinement: swap the owners of two panels of one round while that lowers the largest of the two loads involved
  for (int pass = 0; pass < 64; ++pass) {
    bool moved = false;
    for (long r0 = 0; r0 < npan; r0 += P) {
      const long r1 = std::min<long>(npan, r0 + P);
      for (long a = r0; a < r1; ++a)
        for (long b = a + 1; b < r1; ++b) {
          const int ia = own[(size_t)a], ib = own[(size_t)b];
          const double ca = cost[(size_t)a], cb = cost[(size_t)b];
          const double before = std::max(load[(size_t)ia], load[(size_t)ib]);
          const double la = load[(size_t)ia] - ca + cb, lb = load[(size_t)ib] - cb + ca;
          if (std::max(la, lb) < before * (1.0 - 1e-12)) {
            own[(size_t)a] = ib;
            own[(size_t)b] = ia;
            load[(size_t)ia] = la;
            load[(size_t)ib] = lb;
            moved = true;
          }
        }
    }
    if (!moved) break;
  }
  // (greedy: on adversarial cost vectors it can end above the cyclic deal -- then the cyclic deal it is)
  std::vector<int> cyc((size_t)npan);
  for (long J = 0; J < npan; ++J) cyc[(size_t)J] = (int)(J % P);
  if (max_load(cost, cyc, P) <= max_load(cost, own, P)) return cyc;
  return own;
}

// a caller-supplied table (tests, SGP_MULTI_OWNERS): any map panel -> rank is a valid ownership
inline bool owners_valid(const std::vector<int>& own, int P) {
  for (int o : own)
    if (o < 0 || o >= P) return false;
  return true;
}

}  // namespace sgp
