// Block order of the attention backward under token packing (r04).
//
// attn_bwd_kernel is launched with one block per (role, tile, sample, head) slot of the DENSE sequence length; with packed
// batches about half of those have no tile (they exit after reading the sample's length) and the live ones differ 2x in
// length (1..3 key / query tiles at the MSRVTT fill).  s_memtime budget of the r04 kernel (profiles/r04_attn_budget.txt):
// 536 live blocks on 512 resident slots, the dead blocks of the first half of the grid hold their slots for ~3 k cycles
// each, the dQ half of the grid starts 11 k cycles into the kernel, and the blocks that do not fit the first round are the
// LAST of the grid = the dQ tiles of the longest samples (22.7 k cycles each): the span is 55 k cycles for a mean of 32.6 k
// per CU.  The work list below fixes the order instead: longest first (iterations descending, dK/dV before dQ -- close to
// longest-processing-time-first), dead slots at the END of the grid, every block of one (sample, head) on the SAME XCD
// (position 8 p + x goes to XCD x: its K / V / Q / dO tiles are then re-read from that XCD's L2), and the sample's offset
// and length ride in the item, so a block's first dependent load (cu_seqlens) is gone.
//
// item (int32 x 4): { sample b, head | role << 8 | tile << 16, first row of the sample, its length };  x < 0: no work.
#pragma once
#include <stdint.h>

struct AttnSched {
  const int32_t* cu;  // [B + 1] cumulative sequence lengths (device)
  int32_t* work;      // [(2 * tiles * B * H)][4] out
  int B, H, tiles;    // tiles = ceil(dense sequence length / 64); (B * H) % 8 == 0, B * H <= 8 * 64 * ATT_SCHED_CHUNKS
};
#define ATT_SCHED_CHUNKS 4

// one block of >= 64 threads; wave w builds the lists of XCDs w, w + nwaves, ...
__device__ __forceinline__ void attn_schedule_block(const AttnSched& s) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  const int per_x = s.B * s.H / 8, slots_x = 2 * s.tiles * per_x;  // (sample, head) pairs and list positions per XCD
  for (int x = wave; x < 8; x += nwaves) {
    int nt[ATT_SCHED_CHUNKS], off[ATT_SCHED_CHUNKS], len[ATT_SCHED_CHUNKS];
#pragma unroll
    for (int c = 0; c < ATT_SCHED_CHUNKS; ++c) {
      const int i = c * 64 + lane;
      nt[c] = 0; off[c] = 0; len[c] = 0;
      if (i < per_x) {
        const int b = (x + 8 * i) / s.H;
        off[c] = s.cu[b]; len[c] = s.cu[b + 1] - off[c];
        nt[c] = min((len[c] + 63) >> 6, s.tiles);
      }
    }
    int p = 0;  // next free position of this XCD's list (wave-uniform)
    for (int it = s.tiles; it >= 1; --it)
      for (int role = 0; role < 2; ++role) {
#pragma unroll
        for (int c = 0; c < ATT_SCHED_CHUNKS; ++c) {
          if (c * 64 >= per_x) break;
          const bool sel = nt[c] == it;
          const unsigned long long m = __ballot(sel);
          if (sel) {
            const int before = __builtin_popcountll(m & ((1ull << lane) - 1ull));
            const int bh = x + 8 * (c * 64 + lane), b = bh / s.H, h = bh % s.H;
            for (int t = 0; t < it; ++t) {
              int32_t* w = s.work + ((int64_t)(p + before * it + t) * 8 + x) * 4;
              w[0] = b; w[1] = h | (role << 8) | (t << 16); w[2] = off[c]; w[3] = len[c];
            }
          }
          p += __builtin_popcountll(m) * it;
        }
      }
    for (int q = p + lane; q < slots_x; q += 64) s.work[((int64_t)q * 8 + x) * 4] = -1;
  }
}
