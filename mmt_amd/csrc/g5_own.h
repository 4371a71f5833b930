// Tile order and tile ownership of the persistent GEMM (gemm5.hip), host-callable so that tests/test_host_cpu.py can check
// the exact-cover and XCD-balance properties on the CPU (hipcc-compiled harness: tests/helpers/g5_own_check.hip).
#pragma once

// tile id -> (m0, n0): bands of 8 tile rows, column by column inside a band (gemm2.hip)
// Narrow outputs (<= 8 tile columns: N = 512) go row by row instead: an XCD's chunk of consecutive ids is then a few WHOLE tile
// rows, every A row-tile -- the long-K GEMMs' big operand -- is fetched from far memory by one XCD only and shared by its
// column tiles through that XCD's L2 (column by column inside a band, 29 ids are 8 rows x 3.6 columns: A comes in 2.2 times).
template <int BN>
__host__ __device__ __forceinline__ void g5_tile(int id, int tiles_n, int tile_rows, int& m0, int& n0) {
  if (tiles_n <= 8) {
    const int r = id / tiles_n;
    m0 = r * 128;
    n0 = (id - r * tiles_n) * BN;
    return;
  }
  const int band = id / (8 * tiles_n), first = band * 8;
  const int rows_here = (tile_rows - first < 8 ? tile_rows - first : 8);
  const int within = id - band * 8 * tiles_n;
  m0 = (first + within % rows_here) * 128;
  n0 = (within / rows_here) * BN;
}

// Which tiles a block owns.  Round r of G tiles gives XCD x (= bid & 7: consecutive blocks go to consecutive XCDs) the ids
// r G + x G/8 + (bid >> 3); the LAST, partial round is cut into eight equal chunks instead, so that a launch with fewer tiles
// than blocks (the packed K = 3072 GEMMs: 116 or 232 tiles) still runs on all eight XCDs' L2s and memory paths rather than
// on the first few (r05: 39-42 us inside the step against 28 us with everything in four XCDs' reach -- DESIGN section 7).
struct G5Own {
  int full, n, body0, tail_id, G;
  __host__ __device__ __forceinline__ int id(int i) const { return i < full ? i * G + body0 : tail_id; }
};
__host__ __device__ __forceinline__ G5Own g5_own(int live, int G, int bid) {
  G5Own o;
  const int x = bid & 7, j = bid >> 3;
  o.G = G;
  o.full = live / G;
  const int rem = live - o.full * G, per = (rem + 7) >> 3;
  o.body0 = x * (G >> 3) + j;
  o.tail_id = o.full * G + x * per + j;
  o.n = o.full + ((j < per && x * per + j < rem) ? 1 : 0);
  return o;
}

