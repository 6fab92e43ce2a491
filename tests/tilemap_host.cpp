// Host-side exhaustive check of the workgroup -> tile enumeration of the GEMM launches
// (stheno.jl_amd/csrc/tilemap.h; compiled by g++ in tests/test_tilemap_host.py -- no GPU involved).
// For every launch shape (n_tr x n_tc tiles; lower-triangular mask_off == 0 or rectangular):
//   * every live tile (tr >= tc + mask_off) is produced by exactly one workgroup id,
//   * no id produces a tile outside the grid or above the mask (dead ids return false),
//   * id % 8 is the XCD the hardware places the workgroup on: tile row tr belongs to the XCD the
//     boustrophedon rule gives it (one row of every 8 per XCD), so an A row panel goes through one L2,
//   * lower-triangular square launches give every XCD the same number of live tiles when n_tr % 16 == 0,
//   * id 0 is tile (0, 0): the fused update + potrf_diag kernel relies on the next diagonal block being
//     the first workgroup dispatched,
//   * the number of ids exceeds the live tiles only by the documented slack.
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../stheno.jl_amd/csrc/tilemap.h"

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

static void check_shape(long n_tr, long n_tc, long mask_off, long* ids_out, long* live_out) {
  const long ids = sgp::tile_ids(n_tr, n_tc, mask_off);
  std::vector<int> hit((size_t)(n_tr * n_tc), 0);
  long per_xcd[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long live = 0;
  for (long id = 0; id < ids; ++id) {
    long tr = -1, tc = -1;
    if (!sgp::tile_of_id(id, n_tr, n_tc, mask_off, tr, tc)) continue;
    CHECK(tr >= 0 && tr < n_tr && tc >= 0 && tc < n_tc, "id %ld -> (%ld, %ld) outside %ld x %ld", id, tr, tc, n_tr, n_tc);
    if (!(tr >= 0 && tr < n_tr && tc >= 0 && tc < n_tc)) continue;
    CHECK(tr >= tc + mask_off, "id %ld -> masked tile (%ld, %ld), shape %ld x %ld", id, tr, tc, n_tr, n_tc);
    const long xcd = id & 7, j = tr / 8;
    CHECK(tr % 8 == ((j & 1) ? 7 - xcd : xcd), "tile row %ld on XCD %ld, shape %ld x %ld", tr, xcd, n_tr, n_tc);
    ++hit[(size_t)(tr * n_tc + tc)];
    ++per_xcd[xcd];
    ++live;
    if (id == 0) CHECK(tr == 0 && tc == 0, "id 0 -> (%ld, %ld), shape %ld x %ld mask %ld", tr, tc, n_tr, n_tc, mask_off);
  }
  long want = 0;
  for (long tr = 0; tr < n_tr; ++tr)
    for (long tc = 0; tc < n_tc; ++tc) {
      const bool is_live = tr >= tc + mask_off;
      want += is_live;
      CHECK(hit[(size_t)(tr * n_tc + tc)] == (is_live ? 1 : 0), "tile (%ld, %ld) produced %d times, shape %ld x %ld mask %ld", tr, tc,
            hit[(size_t)(tr * n_tc + tc)], n_tr, n_tc, mask_off);
    }
  CHECK(live == want, "live %ld != %ld, shape %ld x %ld", live, want, n_tr, n_tc);
  if (mask_off == 0 && n_tr == n_tc && n_tr % 16 == 0)
    for (int x = 1; x < 8; ++x) CHECK(per_xcd[x] == per_xcd[0], "XCD %d has %ld tiles, XCD 0 %ld, n = %ld", x, per_xcd[x], per_xcd[0], n_tr);
  *ids_out = ids;
  *live_out = live;
}

int main() {
  const long NOMASK = -(1L << 40);
  long shapes = 0, worst_num = 0, worst_den = 1, worst_tr = 0, worst_tc = 0;
  // lower-triangular launches: square (trailing updates), tall (look-ahead column updates, K = 128 inner updates,
  // the multi-GPU panel updates), and wider than tall (never launched, must still be exact)
  for (long n_tr = 1; n_tr <= 260; ++n_tr)
    for (long n_tc = 1; n_tc <= 260; n_tc += (n_tc < 70 ? 1 : 7)) {
      long ids, live;
      check_shape(n_tr, n_tc, 0, &ids, &live);
      ++shapes;
      // slack of the launched big shapes (never wider than tall, at least 64 tile columns = 8192 matrix columns): the
      // groups without a closed form -- tile columns beyond a multiple of 64, the partial last group.  (Below 64 tile
      // columns the enumeration is the plain rectangle: up to half of the ids of a SQUARE launch are dead there, a few
      // hundred workgroups that exit at once on an otherwise idle chip.)
      if (n_tc <= n_tr && n_tc >= 64 && (ids - live) * worst_den > worst_num * live) {
        worst_num = ids - live, worst_den = live;
        worst_tr = n_tr, worst_tc = n_tc;
      }
    }
  double worst_big = 0.0;
  for (long n : {512L, 513L, 520L, 768L, 1023L, 1024L, 1026L})   // N = 65536 .. 131328: the BASELINE sizes and beyond
    for (long n_tc : {8L, n, n - 8}) {
      long ids, live;
      check_shape(n, n_tc, 0, &ids, &live);
      ++shapes;
      const double f = (double)(ids - live) / (double)live;
      worst_big = f > worst_big ? f : worst_big;
    }
  // rectangular launches (row solves' updates, L Z, cin products)
  for (long n_tr = 1; n_tr <= 150; n_tr += (n_tr < 40 ? 1 : 5))
    for (long n_tc = 1; n_tc <= 40; ++n_tc) {
      long ids, live;
      check_shape(n_tr, n_tc, NOMASK, &ids, &live);
      ++shapes;
    }
  std::printf("shapes %ld failures %ld worst_dead_fraction %.4f at %ld x %ld worst_dead_fraction_big %.4f\n", shapes, failures,
              (double)worst_num / (double)worst_den, worst_tr, worst_tc, worst_big);
  return failures ? 1 : 0;
}
