// Host check of the dataflow factorisation's task order (stheno.jl_amd/csrc/df_tasks.h), see tests/test_df_tasks_host.py.
#include "../stheno.jl_amd/csrc/df_tasks.h"
#include <algorithm>
#include <cstdio>
#include <vector>

using namespace sgp;

// Replay of chol_df.hip's task loop with W workgroups over nb equally shaped matrices of which only the first T_f tile
// columns are FACTORED (T_f == T_c: the plain factorisation): a free workgroup takes the next id; a held task finishes once
// its inputs are final -- tiles (i, k) and (j, k) for k < min(j, T_f), and the diagonal tile (j, j) for i > j in a factored
// column.  A task of a factored column publishes prog[i] = j + 1; an update-only task (j >= T_f) publishes nothing.  Rounds of
// "finish everything that can finish, then hand out ids" must end with every task done, whatever W.  Odd rounds retire ONE
// held task only (rotating start), so that the replay also visits schedules in which a single late workgroup is the one
// able to move.
static bool replay(int T_r, int T_c, int T_f, int nb, int W) {
  const long per = df_ntasks(T_r, T_c), nt = per * nb;
  std::vector<std::vector<int>> prog(nb, std::vector<int>(T_r, 0));   // final tiles of tile row i of matrix b
  std::vector<long> held;
  long head = 0, done = 0, round = 0;
  std::vector<char> seen((size_t)nt, 0);
  while (done < nt) {
    while ((int)held.size() < W && head < nt) held.push_back(head++);
    bool any = false;
    const size_t n0 = held.size();
    for (size_t k = 0; k < n0 && k < held.size();) {
      const size_t h = (k + (size_t)round) % held.size();
      int b = 0;
      long ql = held[h];
      if (nb > 1) df_batch_task(held[h], nb, b, ql);
      if (b < 0 || b >= nb || ql < 0 || ql >= per) return false;
      int i, j;
      df_task_tile(ql, T_r, T_c, i, j);
      const int jend = std::min(j, T_f);
      std::vector<int>& p = prog[b];
      const bool fin = j < T_f;
      const bool ready = p[i] >= jend && p[j] >= jend && (!fin || i == j || p[j] >= j + 1);
      if (ready) {
        if (seen[(size_t)held[h]]) return false;
        seen[(size_t)held[h]] = 1;
        if (fin) {
          if (p[i] != j) return false;   // row i's tiles must become final in column order
          p[i] = j + 1;
        }
        held[h] = held.back();
        held.pop_back();
        ++done;
        any = true;
        if (round & 1) break;
      } else {
        ++k;
      }
    }
    if (!any) return false;               // nobody can move: a deadlock
    ++round;
  }
  for (int b = 0; b < nb; ++b)
    for (int i = 0; i < T_r; ++i)
      if (prog[b][i] != std::min(i + 1, T_f)) return false;
  return true;
}

int main() {
  long shapes = 0, bad = 0;
  // exhaustive: every id of every shape up to 160 tile columns with 0 .. 3 bordered tile rows
  for (int T_c = 1; T_c <= 160; ++T_c)
    for (int border = 0; border <= 3; ++border) {
      const int T_r = T_c + border;
      const long nt = df_ntasks(T_r, T_c);
      long q = 0;
      bool ok = true;
      for (int j = 0; j < T_c && ok; ++j)
        for (int i = j; i < T_r; ++i, ++q) {
          int gi, gj;
          df_task_tile(q, T_r, T_c, gi, gj);
          if (gi != i || gj != j) {
            ok = false;
            break;
          }
        }
      if (q != nt) ok = false;
      ++shapes;
      if (!ok) {
        ++bad;
        printf("BAD decode T_r=%d T_c=%d\n", T_r, T_c);
      }
    }
  // large shapes (N up to 4 million columns would be T = 32768): the column boundaries and their neighbours
  for (int T_c : {511, 512, 1000, 2048, 8191, 32767})
    for (int border : {0, 1, 77}) {
      const int T_r = T_c + border;
      bool ok = true;
      for (int j = 0; j < T_c; j += (T_c > 3000 ? 37 : 1)) {
        const long s0 = df_col_start(T_r, j);
        for (long q : {s0, s0 + 1, s0 + (T_r - j) - 1}) {
          if (q < s0 || q >= s0 + (T_r - j)) continue;
          int gi, gj;
          df_task_tile(q, T_r, T_c, gi, gj);
          if (gj != j || gi != j + (int)(q - s0)) ok = false;
        }
      }
      ++shapes;
      if (!ok) {
        ++bad;
        printf("BAD decode (large) T_r=%d T_c=%d\n", T_r, T_c);
      }
    }
  long replays = 0, part = 0, batch = 0;
  for (int T_c : {1, 2, 3, 7, 16, 33})
    for (int border : {0, 1, 3})
      for (int W : {1, 2, 3, 8, 64, 512, 5000}) {
        ++replays;
        if (!replay(T_c + border, T_c, T_c, 1, W)) {
          ++bad;
          printf("BAD replay T_r=%d T_c=%d W=%d\n", T_c + border, T_c, W);
        }
      }
  // panel launches that factor T_f of their T_c columns (the sharded factorisation's sub-panels: 4 of 8, 2 of 8, ...) over tall
  // panels
  for (int T_c : {2, 4, 8, 16})
    for (int T_f = 1; T_f <= T_c; T_f += std::max(1, T_c / 4))
      for (int border : {0, 1, 40, 500})
        for (int W : {1, 2, 7, 64, 256, 5000}) {
          ++part;
          if (!replay(T_c + border, T_c, T_f, 1, W)) {
            ++bad;
            printf("BAD partial replay T_r=%d T_c=%d T_f=%d W=%d\n", T_c + border, T_c, T_f, W);
          }
        }
  // batches of independent matrices dealt round robin
  for (int nb : {2, 3, 8, 16})
    for (int T_c : {1, 2, 5, 16, 32})
      for (int border : {0, 1})
        for (int W : {1, 2, 3, 8, 64, 256, 5000}) {
          ++batch;
          if (!replay(T_c + border, T_c, T_c, nb, W)) {
            ++bad;
            printf("BAD batch replay nb=%d T_r=%d T_c=%d W=%d\n", nb, T_c + border, T_c, W);
          }
        }
  printf("shapes %ld replays %ld partial %ld batch %ld bad %ld\n", shapes, replays, part, batch, bad);
  return bad ? 1 : 0;
}
