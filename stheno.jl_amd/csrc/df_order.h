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

}  // namespace sgp
