// Task order of the dataflow factorisation (chol_df.hip) as plain integer functions that also compile for the host:
// tests/df_order_host.cpp enumerates whole factorisations on the CPU, checks that the ids map onto the tiles one to one in
// column-major order, and replays the schedule with a simulated pool of workgroups to check the progress argument (every
// task's inputs belong to tasks with smaller ids, so the holder of the smallest unfinished id can always finish).
#pragma once
#ifndef __HIPCC__
#include <cmath>
#define __host__
#define __device__
#define __forceinline__ inline
#endif
#include <algorithm>
#include <cstdint>
#include <vector>

namespace sgp {

// tasks of a T_r x T_c tile grid (T_r >= T_c tile rows; the tiles on and below the diagonal of the first T_c columns)
__host__ __device__ __forceinline__ long df_ntasks(int T_r, int T_c) { return (long)T_c * T_r - (long)T_c * (T_c - 1) / 2; }
// first task id of tile column j: columns 0 .. j - 1 hold T_r, T_r - 1, ..., T_r - j + 1 tasks
__host__ __device__ __forceinline__ long df_col_start(int T_r, int j) { return (long)j * T_r - (long)j * (j - 1) / 2; }

// task id q -> tile (i, j): column-major, column j = tiles (j, j), (j + 1, j), ..., (T_r - 1, j).  The square root only
// seeds the search; the two loops make the answer exact whatever its rounding.
__host__ __device__ __forceinline__ void df_task_tile(long q, int T_r, int T_c, int& i, int& j) {
  const double b = 2.0 * T_r + 1.0;
  int c = (int)((b - sqrt(b * b - 8.0 * (double)q)) * 0.5);
  if (c < 0) c = 0;
  if (c >= T_c) c = T_c - 1;
  while (c > 0 && df_col_start(T_r, c) > q) --c;
  while (c + 1 < T_c && df_col_start(T_r, c + 1) <= q) ++c;
  j = c;
  i = c + (int)(q - df_col_start(T_r, c));
}


// ---- XCD-affine order (round 4) -------------------------------------------------------------------------------------------
// The column-major order above hands neighbouring workgroups tiles of ONE column whose contractions start at different times,
// so every workgroup streams its two operand row panels from the fabric on its own (L2 hit 0.11 at 32768 columns, 82 x the
// algorithmic bytes).  Here the tasks are dealt into DF_NQ = 8 in-order queues -- workgroup id % 8 is the XCD the hardware
// dispatches it to, and that is the queue it serves -- such that the workgroups of one XCD hold, at any time, a PATCH of
// pr owned tile rows x pc consecutive tile columns: pr + pc operand panels for pr * pc tiles, contracted over the same k
// range at the same time (the tiles of a patch are taken together and start at k = 0 together), i.e. shared through that
// XCD's L2 exactly as the 8 x 8 patches of the launch-based update kernel are (gemm_nt.hip).
//   * tile row i belongs to ONE queue, df_row_queue(i) (rows 8m .. 8m + 7 go to distinct queues, in zig-zag order so that
//     the triangular work balances): a row's A panel only ever travels through one XCD's L2;
//   * block column J = tile columns [pc J, pc J + pc); queue x walks the block columns in order and, inside one, its owned
//     rows >= pc J in chunks of pr (a patch), the tiles of a patch column by column, rows ascending.
// Key (J, patch, column, row) is a total order on the tiles that respects every dependency (tile (i, j) needs (i, k) and
// (j, k), k < j, and (j, j): same row = same queue and an earlier column; row j lies in [pc J, pc J + pc) and is therefore
// in patch 0 of its queue with columns <= j), and every queue is a subsequence of it.  So the earliest unfinished tile is
// either held (then it can finish) or at the head of its queue with every earlier task of that queue finished -- a
// workgroup of that queue is free to take it: no deadlock for any number of workgroups >= DF_NQ, no residency requirement.
// A workgroup whose own queue is exhausted serves the other queues (in order, too).  tests/df_order_host.cpp replays it.
constexpr int DF_NQ = 8;
__host__ __device__ __forceinline__ int df_row_queue(int i) { return ((i >> 3) & 1) ? 7 - (i & 7) : (i & 7); }
__host__ __device__ __forceinline__ uint32_t df_pack(int i, int j) { return ((uint32_t)i << 16) | (uint32_t)j; }
__host__ __device__ __forceinline__ void df_unpack(uint32_t t, int& i, int& j) {
  i = (int)(t >> 16);
  j = (int)(t & 0xffffu);
}

// tasks of all queues back to back; qstart[x] .. qstart[x + 1] = queue x.  T_r <= 65535 (8.4 million rows).
// pend (optional): for every task, the queue-relative index one past the last task of its patch (the soft gang start of
// chol_df.hip waits until the queue's head has passed it, i.e. until every tile of the patch has been taken).
inline void df_build_queues(int T_r, int T_c, int pr, int pc, std::vector<uint32_t>& tasks, int (&qstart)[DF_NQ + 1],
                            std::vector<uint32_t>* pend = nullptr) {
  std::vector<uint32_t> q[DF_NQ], pe[DF_NQ];
  std::vector<int> rows[DF_NQ];
  pr = std::max(pr, 1);
  pc = std::max(pc, 1);
  for (int J0 = 0; J0 < T_c; J0 += pc) {
    const int J1 = std::min(J0 + pc, T_c);
    for (int x = 0; x < DF_NQ; ++x) rows[x].clear();
    for (int i = J0; i < T_r; ++i) rows[df_row_queue(i)].push_back(i);
    for (int x = 0; x < DF_NQ; ++x)
      for (size_t p0 = 0; p0 < rows[x].size(); p0 += (size_t)pr) {
        const size_t p1 = std::min(p0 + (size_t)pr, rows[x].size());
        const size_t first = q[x].size();
        for (int j = J0; j < J1; ++j)
          for (size_t r = p0; r < p1; ++r)
            if (rows[x][r] >= j) q[x].push_back(df_pack(rows[x][r], j));
        pe[x].resize(q[x].size(), (uint32_t)q[x].size());
        (void)first;
      }
  }
  tasks.clear();
  if (pend) pend->clear();
  for (int x = 0; x < DF_NQ; ++x) {
    qstart[x] = (int)tasks.size();
    tasks.insert(tasks.end(), q[x].begin(), q[x].end());
    if (pend) pend->insert(pend->end(), pe[x].begin(), pe[x].end());
  }
  qstart[DF_NQ] = (int)tasks.size();
}

}  // namespace sgp
