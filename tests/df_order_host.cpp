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

// XCD-affine order (df_build_queues): every tile exactly once, and a replay in which workgroup w serves queue w % 8 IN
// ORDER and moves on to the next queue only when its own is exhausted -- as the kernel does -- must finish for any number
// of workgroups >= DF_NQ.  Odd rounds retire ONE held task only (rotating start), so that the replay also visits schedules
// in which a single late workgroup is the one able to move.
static bool replay_queues(int T_r, int T_c, int pr, int pc, int W) {
  std::vector<uint32_t> tasks;
  int qs[DF_NQ + 1];
  df_build_queues(T_r, T_c, pr, pc, tasks, qs);
  const long nt = df_ntasks(T_r, T_c);
  if ((long)tasks.size() != nt || qs[DF_NQ] != (int)nt) return false;
  std::vector<char> seen((size_t)T_r * T_c, 0);
  for (int x = 0; x < DF_NQ; ++x)
    for (int q = qs[x]; q < qs[x + 1]; ++q) {
      int i, j;
      df_unpack(tasks[q], i, j);
      if (i < j || i >= T_r || j >= T_c || seen[(size_t)i * T_c + j]) return false;
      if (df_row_queue(i) != x) return false;   // a tile row lives in ONE queue
      seen[(size_t)i * T_c + j] = 1;
    }
  std::vector<int> prog(T_r, 0), head(DF_NQ, 0), cur(W), left(W, DF_NQ);
  std::vector<long> held(W, -1);
  for (int w = 0; w < W; ++w) cur[w] = w % DF_NQ;
  long done = 0, round = 0;
  while (done < nt) {
    for (int w = 0; w < W; ++w) {
      if (held[w] >= 0) continue;
      while (left[w] > 0) {
        const int len = qs[cur[w] + 1] - qs[cur[w]];
        if (head[cur[w]] < len) {
          held[w] = tasks[qs[cur[w]] + head[cur[w]]++];
          break;
        }
        cur[w] = (cur[w] + 1) % DF_NQ;
        --left[w];
      }
    }
    bool any = false;
    for (int k = 0; k < W; ++k) {
      const int w = (int)((k + round) % W);
      if (held[w] < 0) continue;
      int i, j;
      df_unpack((uint32_t)held[w], i, j);
      const bool ready = prog[i] >= j && prog[j] >= j && (i == j || prog[j] >= j + 1);
      if (!ready) continue;
      if (prog[i] != j) return false;
      prog[i] = j + 1;
      held[w] = -1;
      ++done;
      any = true;
      if (round & 1) break;
    }
    if (!any) return false;
    ++round;
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
  long qreplays = 0;
  for (int T_c : {1, 2, 7, 8, 9, 16, 33, 70})
    for (int border : {0, 1, 3, 40})
      for (int pc : {1, 2, 4, 8})
        for (int pr : {1, 3, 8, 16, 64})
          for (int W : {8, 9, 16, 64, 512, 5000}) {
            ++qreplays;
            if (!replay_queues(T_c + border, T_c, pr, pc, W)) {
              ++bad;
              printf("BAD queue replay T_r=%d T_c=%d pr=%d pc=%d W=%d\n", T_c + border, T_c, pr, pc, W);
            }
          }
  // the production shapes: N = 32768 / 65536 with one bordered tile row, the automatic patches
  for (int T_c : {256, 512})
    for (int pc : {2, 4, 8}) {
      ++qreplays;
      if (!replay_queues(T_c + 1, T_c, 64 / pc, pc, 512)) {
        ++bad;
        printf("BAD queue replay (large) T_c=%d pc=%d\n", T_c, pc);
      }
    }
  printf("shapes %ld replays %ld queue replays %ld bad %ld\n", shapes, replays, qreplays, bad);
  return bad ? 1 : 0;
}
