// Host check of the dataflow factorisation's task order (stheno.jl_amd/csrc/df_order.h), see tests/test_df_order_host.py.
#include "../stheno.jl_amd/csrc/df_order.h"
#include <algorithm>
#include <cstdio>
#include <vector>

using namespace sgp;

// Replay with W workgroups: a free workgroup takes the next id; a held task finishes once its inputs are final --
// tiles (i, k) and (j, k) for k < j, and the diagonal tile (j, j) for i > j.  Rounds of "finish everything that can
// finish, then hand out ids" must end with every tile final, whatever W.
static bool replay(int T_r, int T_c, int W) {
  const long nt = df_ntasks(T_r, T_c);
  std::vector<int> prog(T_r, 0);          // final tiles of tile row i (they become final in column order)
  std::vector<long> held;                 // ids held by workgroups
  long head = 0, done = 0;
  while (done < nt) {
    while ((int)held.size() < W && head < nt) held.push_back(head++);
    bool any = false;
    for (size_t h = 0; h < held.size();) {
      int i, j;
      df_task_tile(held[h], T_r, T_c, i, j);
      const bool ready = prog[i] >= j && prog[j] >= j && (i == j || prog[j] >= j + 1);
      if (ready) {
        if (prog[i] != j) return false;   // row i's tiles must become final in column order
        prog[i] = j + 1;
        held[h] = held.back();
        held.pop_back();
        ++done;
        any = true;
      } else {
        ++h;
      }
    }
    if (!any) return false;               // nobody can move: a deadlock
  }
  for (int i = 0; i < T_r; ++i)
    if (prog[i] != std::min(i + 1, T_c)) return false;
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
  long replays = 0;
  for (int T_c : {1, 2, 3, 7, 16, 33})
    for (int border : {0, 1, 3})
      for (int W : {1, 2, 3, 8, 64, 512, 5000}) {
        ++replays;
        if (!replay(T_c + border, T_c, W)) {
          ++bad;
          printf("BAD replay T_r=%d T_c=%d W=%d\n", T_c + border, T_c, W);
        }
      }
  printf("shapes %ld replays %ld bad %ld\n", shapes, replays, bad);
  return bad ? 1 : 0;
}
