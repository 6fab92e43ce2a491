// Task order of the dataflow factorisation (chol_df.hip) as plain integer functions that also compile for the host:
// tests/df_tasks_host.cpp enumerates whole factorisations on the CPU, checks that the ids map onto the tiles one to one in
// column-major order, and replays the schedule with a simulated pool of workgroups to check the progress argument (every
// task's inputs belong to tasks with smaller ids, so the holder of the smallest unfinished id can always finish) -- also
// for a launch that only FACTORS its first T_f tile columns and merely updates the others (the sharded factorisation's
// sub-panel launches, csrc/multi.hip) and for a batch of independent matrices dealt out round robin (sgp_logpdf_batch).
// (Round 4's XCD-affine task queues lived here as df_order.h; they lost -- profiles/r04_experiments/dataflow_xcd_queues.md
// -- and were removed in round 6.)
#pragma once
#ifndef __HIPCC__
#include <cmath>
#define __host__
#define __device__
#define __forceinline__ inline
#endif
#include <algorithm>
#include <cstdint>

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
// a batch of nb equally shaped matrices: id q -> matrix q % nb, task q / nb of it (every matrix sees its own tasks in
// column-major order; the chains of the nb matrices sit on different workgroups at any time and hide each other's latency)
__host__ __device__ __forceinline__ void df_batch_task(long q, int nb, int& b, long& ql) {
  b = (int)(q % nb);
  ql = q / nb;
}

__host__ __device__ __forceinline__ uint32_t df_pack(int i, int j) { return ((uint32_t)i << 16) | (uint32_t)j; }
__host__ __device__ __forceinline__ void df_unpack(uint32_t t, int& i, int& j) {
  i = (int)(t >> 16);
  j = (int)(t & 0xffffu);
}

}  // namespace sgp
