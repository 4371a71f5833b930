// Host-side check of gemm5.hip's tile order and ownership (mmt_amd/csrc/g5_own.h), compiled with hipcc and run on the CPU by
// tests/test_host_cpu.py.  For a sweep of (tile rows, tile columns, grid) it asserts that the blocks of a launch own every
// live tile exactly once, that a tile id maps to a distinct in-range (row, column), and that in the last partial round no
// XCD (block id & 7) gets more than one tile over an eighth of the rest.  Prints "ok <cases>" or the first violation.
#include <cstdio>
#include <vector>
#include <hip/hip_runtime.h>
#include "../../mmt_amd/csrc/g5_own.h"

template <int BN>
static int check(int tile_rows, int tiles_n, int G) {
  const int live = tile_rows * tiles_n;
  std::vector<int> owner(live, -1), cell(live, 0), per_xcd(8, 0);
  for (int bid = 0; bid < G; ++bid) {
    const G5Own o = g5_own(live, G, bid);
    for (int i = 0; i < o.n; ++i) {
      const int id = o.id(i);
      if (id < 0 || id >= live) { std::printf("id %d out of range: rows %d cols %d G %d bid %d\n", id, tile_rows, tiles_n, G, bid); return 1; }
      if (owner[id] >= 0) { std::printf("tile %d owned twice (%d, %d): rows %d cols %d G %d\n", id, owner[id], bid, tile_rows, tiles_n, G); return 1; }
      owner[id] = bid;
      if (i >= o.full) per_xcd[bid & 7]++;
      int m0, n0;
      g5_tile<BN>(id, tiles_n, tile_rows, m0, n0);
      if (m0 % 128 || n0 % BN || m0 / 128 >= tile_rows || n0 / BN >= tiles_n || m0 < 0 || n0 < 0) {
        std::printf("tile %d -> (%d, %d) out of range: rows %d cols %d\n", id, m0, n0, tile_rows, tiles_n); return 1;
      }
      cell[(m0 / 128) * tiles_n + n0 / BN]++;
    }
  }
  for (int t = 0; t < live; ++t) {
    if (owner[t] < 0) { std::printf("tile %d unowned: rows %d cols %d G %d\n", t, tile_rows, tiles_n, G); return 1; }
    if (cell[t] != 1) { std::printf("cell %d hit %d times: rows %d cols %d G %d\n", t, cell[t], tile_rows, tiles_n, G); return 1; }
  }
  const int rem = live % G, cap = (rem + 7) / 8;
  for (int x = 0; x < 8; ++x)
    if (per_xcd[x] > cap) { std::printf("XCD %d has %d of %d tail tiles: rows %d cols %d G %d\n", x, per_xcd[x], rem, tile_rows, tiles_n, G); return 1; }
  return 0;
}

int main() {
  int cases = 0;
  const int grids[] = {8, 16, 120, 224, 232, 256, 304};
  for (int G : grids)
    for (int rows = 0; rows <= 60; ++rows)
      for (int cols : {1, 2, 4, 8, 9, 12, 24, 32}) {
        if (check<128>(rows, cols, G) || check<64>(rows, cols, G)) return 1;
        cases += 2;
      }
  if (check<128>(177, 4, 256) || check<64>(29, 8, 256) || check<128>(29, 4, 224) || check<128>(163, 32, 256)) return 1;
  std::printf("ok %d\n", cases + 4);
  return 0;
}
