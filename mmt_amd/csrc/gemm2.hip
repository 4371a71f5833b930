// Wide-tile bf16 MFMA NT GEMM for the MMT hot path (gfx950):  C[M,N] = A[M,K] . B[N,K]^T (+ fused epilogue).
//
// Why a second kernel: a 128x128 tile moves (128+128)*64*2 B = 32 KiB from L2 into LDS per 64-deep K-step
// and pays it back with 512 MFMA cycles per SIMD; the L2->CU path delivers ~64 B/clk/CU (34.5 TB/s / 256 CUs,
// MI355X_MICROARCH.md), i.e. 512 clk for those 32 KiB -- the 128^2 tile is L2-bandwidth-bound at <= 50 % MFMA.
// 256x128 / 256x256 tiles halve / quarter the bytes per flop.  Structure (cdna guide section 5, "glds, 2 LDS
// buffers, BK=64, vmcnt(0) + one barrier per K-step"):
//   * up to 8 waves, each owning a 64 x WTN sub-tile as 2 x (WTN/32) fragments of mfma_f32_32x32x16_bf16
//     (operands swapped so a lane ends up with 4 consecutive output columns of one row);
//   * LDS-DMA staging (global_load_lds_dwordx4), lane-linear image, XOR swizzle chunk ^ ((row>>1)&7) applied
//     to the SOURCE address and to the fragment read: conflict-free for the 32-row ds_read_b128 fragments;
//   * epilogue through LDS: the accumulators are transposed into a row-major fp32 image 64 rows at a time,
//     and every global access of the epilogue (bias, residual, GELU aux, outputs) is then a row-contiguous
//     8-16 B/lane access instead of a 32-byte-per-row scatter.
#include "mmt_common.h"
#include "../../include/mmt_hip.h"
#include "gemm_epi.h"
#include "adam_unit.h"

#define BK 64
template <int ROWS, int NW>
__device__ __forceinline__ void stage2(const bf16_t* __restrict__ G, int64_t ld, int row0, int row_max, int k0,
                                       bf16_t* lds_tile, int wave, int lane) {
  constexpr int PER = ROWS / 8 / NW;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int rbase = (wave * PER + i) * 8;  // 8 rows of 128 B per wave-instruction
    const int r = rbase + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    const int gr = min(row0 + r, row_max);
    const bf16_t* src = G + (int64_t)gr * ld + k0 + c * 8;
    __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(lds_tile + rbase * BK), 16, 0, 0);
  }
}

// ---- "NN" operand form: B given as [K, N] row-major (a weight matrix W[out, in] used for the input gradient
// dX = dY . W: the contraction index is W's ROW).  The tile is staged as it lies in memory, [BK rows][BN columns], and
// the MFMA fragments (8 contraction values per output column) come from the hardware transpose read
// ds_read_b64_tr_b16 -- no W^T copy in HBM, no transposing pack kernel.
template <int BN> __device__ __forceinline__ int swz_kn(int r) {
  return BN == 128 ? (((r & 7) << 1) | ((r >> 3) & 1)) : ((r >> 1) & 7);  // 16 / 8 chunks of 16 B per tile row
}
template <int BN, int NW>
__device__ __forceinline__ void stage2_kn(const bf16_t* __restrict__ G, int64_t ld, int k0, int n0, bf16_t* lds_tile,
                                          int wave, int lane) {
  constexpr int CHN = BN / 8, RPI = 64 / CHN, PER = CHN / NW;  // chunks per row, rows per wave-instruction
  static_assert(CHN % NW == 0, "stage2_kn geometry");
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int rbase = (wave * PER + i) * RPI;
    const int r = rbase + lane / CHN;
    const int c = (lane % CHN) ^ swz_kn<BN>(r);
    const bf16_t* src = G + (int64_t)(k0 + r) * ld + n0 + c * 8;
    __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(lds_tile + rbase * BN), 16, 0, 0);
  }
}
// fragment of mfma_f32_32x32x16: lane (column colbase + (lane & 31), half lh = lane >> 5) gets the contraction rows
// kk*16 + lh*8 + 0..7 in natural order (two 4-row transpose reads), matching the A fragment's k order.
template <int BN>
__device__ __forceinline__ bf16x8_t frag_kn(const bf16_t* tile, int kk, int colbase, int lane) {
  const int t = lane & 15;
  const int col = colbase + ((lane >> 4) & 1) * 16 + 4 * (t & 3);
  const int r0 = kk * 16 + (lane >> 5) * 8 + (t >> 2), r1 = r0 + 4;
  const int ch = col >> 3, w = col & 7;
  const bf16_t* p0 = tile + r0 * BN + ((ch ^ swz_kn<BN>(r0)) << 3) + w;
  const bf16_t* p1 = tile + r1 * BN + ((ch ^ swz_kn<BN>(r1)) << 3) + w;
  bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4_t*)LDS_PTR(p0));
  bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4_t*)LDS_PTR(p1));
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// PH ("phased"): TWO groups of WGM x WGN waves, each covering the whole BM x BN tile, take ALTERNATE K-steps (K-step kt
// belongs to group kt & 1 and lives in stage kt % 4 of one four-deep ring): in half-step h group h & 1 runs the fragment
// reads + MFMAs of K-step h while the other group issues the LDS-DMA loads of its own K-step h + 3 -- the scheme of the
// weight-gradient kernel (gemm.hip: wgrad_phased_kernel).  Against the spatial split over 8 waves it doubles the wave
// tile (half the LDS bytes per MAC), halves the barriers per MAC, and hides the ~90 cycles every LDS-DMA instruction
// stalls its wave under the partner wave's MFMAs; the price is one LDS round trip of a partial tile at the end.
template <int BM, int BN, int WGM, int WGN, int NS, int EPI, bool BKN = false, bool PH = false>
__device__ __forceinline__ void gemm2_body(
    const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ B, int64_t ldb,
    void* __restrict__ Cout, int64_t ldc, int M, int N, int K, const MmtEpilogue& epi,
    const int32_t* __restrict__ n_rows_dev, const int bid, const int nblk) {
  constexpr int GW = WGM * WGN;  // waves that tile the output once
  constexpr int NW = PH ? 2 * GW : GW, NT = NW * 64, WTM = BM / WGM, WTN = BN / WGN, MI = WTM / 32, NJ = WTN / 32;
  static_assert(!PH || NS == 4 || NS == 6, "the phased loop runs on a four- or six-deep ring");
  // 8-wave blocks run as two groups in opposite phase (waves w and w+4 share a SIMD): group 0 issues the next
  // stage's LDS-DMA and THEN computes, group 1 computes and THEN issues.  An LDS-DMA instruction stalls its wave for
  // ~100 cycles at issue (measured, tools/gemm_instr.py); staggered, that stall hides under the partner wave's MFMAs.
  constexpr bool STAG = !PH && NW == 8 && NS >= 3;
#ifdef MMT_GEMM2_INSTR
  const long long t_entry = clock64(), w_entry = wall_clock64();  // shader clock (per CU, for durations) and the chip-wide 100 MHz counter
#endif
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16_t* smem = (bf16_t*)smem_raw;

  const int tiles_n = N / BN;
  const int nrows = n_rows_dev ? *n_rows_dev : M;
  const int tid = threadIdx.x;
  // Variable-length packing: only the first ceil(nrows/BM) tile rows are live.  The XCD-aware remap runs over the
  // LIVE tiles only -- remapping the whole (dense-sized) grid would hand every live tile to the first few XCDs
  // and leave the others with nothing but dead tiles.
  const int live_tiles = min(nblk, ((min(nrows, M) + BM - 1) / BM) * tiles_n);
  if (bid >= live_tiles) {  // dead tile: nothing to compute
    if constexpr (EPI == MMT_EPI_DGELU) {
      if (epi.colsum && BM >= 128 && bid < ((M + BM - 1) / BM) * tiles_n) {
        const int dm0 = (bid / tiles_n) * BM, dn0 = (bid % tiles_n) * BN;
        for (int h = 0; h < BM / 128; ++h)
          if (dm0 + h * 128 < M && tid < BN) epi.colsum[(int64_t)(dm0 / 128 + h) * N + dn0 + tid] = 0.f;
      }
    }
    // r06: a block without a tile works on the optimizer queue the launch carries (adam_unit.h) until the launch's own
    // blocks are in their last round -- Adam's HBM streaming under the MFMA-bound tiles, on CUs that would sit idle
    if constexpr (NT % 256 == 0) {
      if (epi.rider)
        adam_rider_run<NT>(epi.rider, epi.rider_limit, epi.rider_slot, live_tiles * (int)gridDim.y,
                           (bid - live_tiles) * (int)gridDim.y + (int)blockIdx.y, epi.rider_cap, smem_raw);
    }
    return;
  }
  const int id = xcd_remap(bid, live_tiles);
  // Wide outputs (N >= 1024: QKV, FFN-up, dGELU) walk the tile grid in bands of GROUP tile rows, column by column inside a
  // band: the ~live_tiles / 8 consecutive ids an XCD owns then touch GROUP A panels + a third of the B panels instead of
  // every B panel (the whole weight matrix) + 4 A panels.  Narrow outputs (<= 8 tile columns) stay row-major.
  int tm, tn;
  if (tiles_n * BN >= 1024) {
    constexpr int GROUP = 8;
    const int tile_rows = live_tiles / tiles_n;           // live tile rows (live_tiles is a multiple of tiles_n)
    const int band = id / (GROUP * tiles_n), first = band * GROUP;
    const int rows_here = min(GROUP, tile_rows - first);  // the last band may be shorter
    const int within = id - band * GROUP * tiles_n;
    tm = first + within % rows_here;
    tn = within / rows_here;
  } else {
    tm = id / tiles_n;
    tn = id % tiles_n;
  }
  const int m0 = tm * BM, n0 = tn * BN;
  const int lane = tid & 63, wave8 = tid >> 6;
  const int kg = PH ? wave8 / GW : 0;           // group (phased mode)
  const int wave = PH ? wave8 % GW : wave8;     // wave inside its group
  const int wm = wave / WGN, wn = wave % WGN;
  const int l31 = lane & 31, lh = lane >> 5;
  const bool late_issue = STAG && wave >= NW / 2;

  f32x16 acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  constexpr int STAGE = (BM + BN) * BK;
  constexpr int L = (BM + BN) / 8 / (PH ? GW : NW);  // LDS-DMA instructions per wave per stage
  const int KT = K / BK;
  const int amax = M - 1, bmax = N - 1;
  // NS-deep LDS ring, counted vmcnt: stage kt+NS-1 is issued while stage kt is consumed and the DMA queue is never
  // drained inside the loop (raw s_barrier -- __syncthreads() would add vmcnt(0), cdna guide section 5).
  auto issue_ph = [&](int kt) {  // phased mode: the four waves of group kt & 1 load K-step kt
    if (kt < KT) {
      bf16_t* base = smem + (kt % NS) * STAGE;  // (phased mode: NS = ring depth, 4 or 6)
      stage2<BM, GW>(A, lda, m0, amax, kt * BK, base, wave, lane);
      if constexpr (BKN) stage2_kn<BN, GW>(B, ldb, kt * BK, n0, base + BM * BK, wave, lane);
      else stage2<BN, GW>(B, ldb, n0, bmax, kt * BK, base + BM * BK, wave, lane);
    }
  };
  if constexpr (PH) {
    // group g has requested its first NS / 2 own K-steps (g, g + 2, ..) but the last, which goes out in the first half-step
    if (kg == 0) { issue_ph(0); issue_ph(2); if (NS >= 6) issue_ph(4); } else { issue_ph(1); if (NS >= 6) issue_ph(3); }
  } else {
#pragma unroll
  for (int s0 = 0; s0 < NS - 1; ++s0)
    if (s0 < KT) {
      stage2<BM, NW>(A, lda, m0, amax, s0 * BK, smem + s0 * STAGE, wave, lane);
      if constexpr (BKN) stage2_kn<BN, NW>(B, ldb, s0 * BK, n0, smem + s0 * STAGE + BM * BK, wave, lane);
      else stage2<BN, NW>(B, ldb, n0, bmax, s0 * BK, smem + s0 * STAGE + BM * BK, wave, lane);
    }
  }
  // per-lane LDS offsets of this wave's fragments (row r, 16-byte chunk c lives at chunk c ^ ((r>>1)&7))
  int aoff[MI], boff[NJ], asw[MI], bsw[NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int r = wm * WTM + i * 32 + l31;
    aoff[i] = r * BK; asw[i] = (r >> 1) & 7;
  }
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int r = wn * WTN + j * 32 + l31;
    boff[j] = BM * BK + r * BK; bsw[j] = (r >> 1) & 7;
  }
  int cur = 0;
#ifdef MMT_GEMM2_INSTR
  long long t_wait = 0, t_bar = 0, t_issue = 0, t_comp = 0, t0 = clock64(), tp = t0;
#define TICK(acc) do { const long long tn_ = clock64(); acc += tn_ - tp; tp = tn_; } while (0)
#else
#define TICK(acc) do {} while (0)
#endif
  for (int kt = 0; kt < KT; ++kt) {
    bool do_issue = false;
    int nxt = 0;
    if constexpr (PH) {
      const bool mine = (kt & 1) == kg;
      if (mine) {  // outstanding loads of this wave: K-steps kt, kt + 2 (and kt + 4 in the six-deep ring), L instructions each
        if (NS >= 6 && kt + 4 < KT) wait_vmcnt<2 * L>();
        else if (kt + 2 < KT) wait_vmcnt<L>();
        else wait_vmcnt<0>();
      }
      TICK(t_wait);
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      TICK(t_bar);
      cur = kt % NS;
      if (!mine) {
        issue_ph(kt + NS - 1);
        TICK(t_issue);
        continue;
      }
    } else {
    const int ahead = min(KT - kt - 1, NS - 2);  // stages issued after kt that may stay in flight
    // (r06: a six-deep ring -- five stages in flight, 144 KiB -- for the launches that do not fill the chip, the text tower's
    // and the compact last layer's, measured neutral: tower step 4.098 / 4.105 vs 4.109 / 4.104 ms, headline + 2 us; their
    // K-step is the issue cost of the LDS-DMA instructions + the barrier, not a prefetch distance.  Removed again.)
    if (NS >= 4 && ahead >= 2) wait_vmcnt<2 * L>();
    else if (NS >= 3 && ahead >= 1) wait_vmcnt<L>();
    else wait_vmcnt<0>();
    TICK(t_wait);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    TICK(t_bar);
    nxt = cur + NS - 1;
    if (nxt >= NS) nxt -= NS;
    do_issue = kt + NS - 1 < KT;
    }
    if (do_issue && !late_issue) {
      stage2<BM, NW>(A, lda, m0, amax, (kt + NS - 1) * BK, smem + nxt * STAGE, wave, lane);
      if constexpr (BKN) stage2_kn<BN, NW>(B, ldb, (kt + NS - 1) * BK, n0, smem + nxt * STAGE + BM * BK, wave, lane);
      else stage2<BN, NW>(B, ldb, n0, bmax, (kt + NS - 1) * BK, smem + nxt * STAGE + BM * BK, wave, lane);
    }
    TICK(t_issue);
    const bf16_t* st = smem + cur * STAGE;
    if constexpr (PH) {
      // One wave per SIMD computes here, so nothing else covers LDS latency: ALL fragment reads of the K-step go out first
      // and the MFMAs of k-sub-step kk wait only for ITS fragments (in-order returns).  The reads are inline asm -- left to
      // the compiler, each read is sunk in front of its MFMA (a full LDS round trip per MFMA) or all are waited for at
      // once -- and a fragment's address for sub-step kk is the sub-step-0 address XOR 32 kk (chunk index bits 1-2).
      static_assert(!BKN, "phased mode: NT operand form");
      u32x4 pa[4][MI], pb[4][NJ];
      unsigned abase[MI], bbase[NJ];
      const unsigned sbase = (unsigned)(uintptr_t)LDS_PTR(st);
#pragma unroll
      for (int i = 0; i < MI; ++i) abase[i] = sbase + (unsigned)(aoff[i] * 2) + (unsigned)((lh ^ asw[i]) << 4);
#pragma unroll
      for (int j = 0; j < NJ; ++j) bbase[j] = sbase + (unsigned)(boff[j] * 2) + (unsigned)((lh ^ bsw[j]) << 4);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
        for (int i = 0; i < MI; ++i) asm volatile("ds_read_b128 %0, %1" : "=v"(pa[kk][i]) : "v"(abase[i] ^ (unsigned)(kk << 5)));
#pragma unroll
        for (int j = 0; j < NJ; ++j) asm volatile("ds_read_b128 %0, %1" : "=v"(pb[kk][j]) : "v"(bbase[j] ^ (unsigned)(kk << 5)));
      }
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        // (MI + NJ) reads per sub-step: sub-step kk has landed when at most (3 - kk) (MI + NJ) are outstanding
        constexpr int PER = MI + NJ;
        if (kk == 0) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(3 * PER) : "memory");
        if (kk == 1) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * PER) : "memory");
        if (kk == 2) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(1 * PER) : "memory");
        if (kk == 3) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < MI; ++i) asm volatile("" : "+v"(pa[kk][i]));  // consumers stay below the wait
#pragma unroll
        for (int j = 0; j < NJ; ++j) asm volatile("" : "+v"(pb[kk][j]));
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, pb[kk][j]),
                                                                __builtin_bit_cast(bf16x8_t, pa[kk][i]), acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);  // (or the scheduler sinks these MFMAs below the later waits)
      }
#ifdef MMT_GEMM2_INSTR
      asm volatile("s_nop 0" ::"v"(acc[0][0][0]), "v"(acc[MI - 1][NJ - 1][15]));
#endif
      TICK(t_comp);
      continue;
    }
    // fragment reads software-pipelined one k-substep ahead of the MFMAs that consume them
    bf16x8_t af[2][MI], bfr[2][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i) af[0][i] = *(const bf16x8_t*)(st + aoff[i] + ((lh ^ asw[i]) << 3));
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      if constexpr (BKN) bfr[0][j] = frag_kn<BN>(st + BM * BK, 0, wn * WTN + j * 32, lane);
      else bfr[0][j] = *(const bf16x8_t*)(st + boff[j] + ((lh ^ bsw[j]) << 3));
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      if (kk < 3) {
        const int c = (kk + 1) * 2 + lh;
#pragma unroll
        for (int i = 0; i < MI; ++i) af[(kk + 1) & 1][i] = *(const bf16x8_t*)(st + aoff[i] + ((c ^ asw[i]) << 3));
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          if constexpr (BKN) bfr[(kk + 1) & 1][j] = frag_kn<BN>(st + BM * BK, kk + 1, wn * WTN + j * 32, lane);
          else bfr[(kk + 1) & 1][j] = *(const bf16x8_t*)(st + boff[j] + ((c ^ bsw[j]) << 3));
        }
      }
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[kk & 1][j], af[kk & 1][i], acc[i][j], 0, 0, 0);
    }
    if (do_issue && late_issue) {
      stage2<BM, NW>(A, lda, m0, amax, (kt + NS - 1) * BK, smem + nxt * STAGE, wave, lane);
      if constexpr (BKN) stage2_kn<BN, NW>(B, ldb, (kt + NS - 1) * BK, n0, smem + nxt * STAGE + BM * BK, wave, lane);
      else stage2<BN, NW>(B, ldb, n0, bmax, (kt + NS - 1) * BK, smem + nxt * STAGE + BM * BK, wave, lane);
    }
    if constexpr (!PH) cur = cur + 1 == NS ? 0 : cur + 1;
#ifdef MMT_GEMM2_INSTR
    asm volatile("s_nop 0" ::"v"(acc[0][0][0]), "v"(acc[MI - 1][NJ - 1][15]));
#endif
    TICK(t_comp);
  }
  if constexpr (PH) wait_vmcnt<0>();
  // (riders of this launch -- adam_unit.h -- stop claiming work when the first block of the last round gets here: the
  // epilogue that follows is the notice they need to finish the pass they are in)
  if (epi.rider && tid == 0 && blockIdx.y == 0 && ((epi.rider_cap >> 12) & 15) == 0 &&
      bid == adam_rider_signal_x(live_tiles, (int)gridDim.y, epi.rider_slot))
    adam_rider_host_done(epi.rider, epi.rider_slot);
#ifdef MMT_GEMM2_INSTR
  const long long t_loop_end = clock64();
#endif

  // ---- epilogue: 64 rows at a time through a row-major fp32 LDS image (gemm_epi.h) -----------------------------
#ifdef MMT_GEMM2_INSTR
  long long eticks[3] = {0, 0, 0};
  gemm_tile_epilogue<BM, BN, WGM, WGN, NT, EPI, PH>(acc, smem_raw, m0, n0, M, N, nrows, Cout, ldc, epi, wm, wn, kg, tid, eticks);
  const long long e_head = eticks[2] - t_loop_end, e_stage = eticks[0], e_sweep = eticks[1];
#else
  gemm_tile_epilogue<BM, BN, WGM, WGN, NT, EPI, PH>(acc, smem_raw, m0, n0, M, N, nrows, Cout, ldc, epi, wm, wn, kg, tid, nullptr);
#endif
#ifdef MMT_GEMM2_INSTR
  if (epi.row_index == nullptr && epi.seed_dev != nullptr && tid == 0) {  // lab: seed_dev doubles as the debug buffer
    long long* dbgbuf = (long long*)epi.seed_dev + (int64_t)bid * 16;
    dbgbuf[0] = t_wait; dbgbuf[1] = t_bar; dbgbuf[2] = t_issue; dbgbuf[3] = t_comp;
    dbgbuf[4] = t_loop_end - t0; dbgbuf[5] = clock64() - t_loop_end; dbgbuf[7] = KT;
    dbgbuf[8] = t0 - t_entry; dbgbuf[9] = e_head; dbgbuf[10] = e_stage; dbgbuf[11] = e_sweep; dbgbuf[12] = w_entry;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the tile's stores have left the wave
    dbgbuf[6] = clock64() - t_entry;  // whole block, shader cycles
    dbgbuf[13] = wall_clock64();
    // where the block ran: XCC_ID (hwreg 20) and HW_ID (hwreg 4: cu_id [11:8], sh_id [12], se_id [15:13])
    dbgbuf[14] = (long long)__builtin_amdgcn_s_getreg((31 << 11) | 20);
    dbgbuf[15] = (long long)__builtin_amdgcn_s_getreg((31 << 11) | 4);
  }
#endif
}

template <int BM, int BN, int WGM, int WGN, int NS, int EPI, bool BKN = false, bool PH = false>
__global__ __launch_bounds__((PH ? 2 : 1) * WGM * WGN * 64) void gemm2_kernel(
    const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ B, int64_t ldb,
    void* __restrict__ Cout, int64_t ldc, int M, int N, int K, MmtEpilogue epi,
    const int32_t* __restrict__ n_rows_dev) {
  gemm2_body<BM, BN, WGM, WGN, NS, EPI, BKN, PH>(A, lda, B, ldb, Cout, ldc, M, N, K, epi, n_rows_dev, (int)blockIdx.x,
                                                 (int)gridDim.x);
}

// Split-K for skinny problems (few output tiles, long K: the last layer's read-out rows): blockIdx.y = K-slice, every
// slice writes a raw fp32 partial tile into its own slab; splitk_epilogue_kernel sums the slabs in a fixed order and
// applies the epilogue.  A 16-tile x 48-K-step GEMM is otherwise one 40 us chain of dependent K-steps.
template <int BM, int BN, int WGM, int WGN, int NS, bool BKN = false>
__global__ __launch_bounds__(WGM * WGN * 64) void gemm2_splitk_kernel(
    const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ B, int64_t ldb, float* __restrict__ ws,
    int64_t slab_stride, int64_t ldws, int M, int N, int K, int kchunk, const int32_t* __restrict__ n_rows_dev,
    const void* rider, int rider_limit, int rider_slot, int rider_cap) {
  const int z = blockIdx.y;
  const int kb = z * kchunk;
  const int kl = min(kchunk, K - kb);
  MmtEpilogue e = {};
  e.rider = rider; e.rider_limit = rider_limit; e.rider_slot = rider_slot; e.rider_cap = rider_cap;  // (blocks without a tile: adam_unit.h)
  gemm2_body<BM, BN, WGM, WGN, NS, MMT_EPI_F32, BKN>(A + kb, lda, BKN ? B + (int64_t)kb * ldb : B + kb, ldb,
                                                      ws + (int64_t)z * slab_stride, ldws, M, N, kl, e, n_rows_dev,
                                                      (int)blockIdx.x, (int)gridDim.x);
}

template <int EPI>
__global__ __launch_bounds__(256) void splitk_epilogue_kernel(const float* __restrict__ ws, int64_t slab_stride, int64_t ldws,
                                                              int splits, void* __restrict__ Cout, int64_t ldc, int M, int N,
                                                              MmtEpilogue epi, const int32_t* __restrict__ n_rows_dev) {
  if (n_rows_dev) M = min(M, *n_rows_dev);
  const int n4 = N >> 2;
  unsigned dkey = 0;
  if constexpr (EPI == MMT_EPI_BIAS_DROP_RES) dkey = eff_key(epi.drop_key, epi.seed_dev);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < (int64_t)M * n4; i += (int64_t)gridDim.x * blockDim.x) {
    const int row = (int)(i / n4), col = (int)(i % n4) * 4;
    const float* p0 = ws + (int64_t)row * ldws + col;
    f32x4 v = *(const f32x4*)p0;
    int s = 1;
    for (; s + 3 < splits; s += 4) {  // fixed association (deterministic), four loads in flight
      const f32x4 a0 = *(const f32x4*)(p0 + (int64_t)s * slab_stride), a1 = *(const f32x4*)(p0 + (int64_t)(s + 1) * slab_stride);
      const f32x4 a2 = *(const f32x4*)(p0 + (int64_t)(s + 2) * slab_stride), a3 = *(const f32x4*)(p0 + (int64_t)(s + 3) * slab_stride);
      v += (a0 + a1) + (a2 + a3);
    }
    for (; s < splits; ++s) v += *(const f32x4*)(p0 + (int64_t)s * slab_stride);
    if constexpr (EPI == MMT_EPI_BIAS_DROP_RES || EPI == MMT_EPI_BIAS_F32) v += *(const f32x4*)(epi.bias + col);
    if constexpr (EPI == MMT_EPI_BIAS_DROP_RES) {
      if (epi.drop_thr16) {
        const int orow = epi.row_index ? epi.row_index[row] : row;
        bool k[4];
        keep4(dkey, (unsigned long long)orow * (unsigned)N + (unsigned)col, epi.drop_thr16, k);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = k[e] ? v[e] * epi.drop_scale : 0.f;
      }
    }
    if constexpr (EPI == MMT_EPI_BIAS_DROP_RES || EPI == MMT_EPI_ADD_F32)
      v += *(const f32x4*)(epi.res + (int64_t)row * epi.ldres + col);
    if constexpr (EPI == MMT_EPI_BF16) {
      u32x2 o = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
      *(u32x2*)((bf16_t*)Cout + (int64_t)row * ldc + col) = o;
      if (epi.dot_out) {  // (N % 64 == 0: the 16 lanes of a 64-column group share the row and run the loop together)
        const u32x2 c = *(const u32x2*)((const bf16_t*)epi.dot_src + (int64_t)row * epi.lddot + col);
        float part = bf2f((bf16_t)(o[0] & 0xffff)) * bf2f((bf16_t)(c[0] & 0xffff)) + bf2f((bf16_t)(o[0] >> 16)) * bf2f((bf16_t)(c[0] >> 16)) +
                     bf2f((bf16_t)(o[1] & 0xffff)) * bf2f((bf16_t)(c[1] & 0xffff)) + bf2f((bf16_t)(o[1] >> 16)) * bf2f((bf16_t)(c[1] >> 16));
        part += __shfl_xor(part, 1, 64); part += __shfl_xor(part, 2, 64);
        part += __shfl_xor(part, 4, 64); part += __shfl_xor(part, 8, 64);
        if ((threadIdx.x & 15) == 0) epi.dot_out[(int64_t)row * (N >> 6) + (col >> 6)] = part;
      }
    } else {
      *(f32x4*)((float*)Cout + (int64_t)row * ldc + col) = v;
    }
  }
}

// geometry of the slabs mmt_gemm_nt_splitk_ex(..., no_epilogue = 1) leaves behind: *splits slabs of *slab_stride floats,
// leading dimension N (for consumers that fold the reduction into their own pass)
extern "C" int mmt_gemm_splitk_geometry(int M, int N, int K, int splits_requested, int* splits, int64_t* slab_stride) {
  if (M <= 0 || N <= 0 || K <= 0 || K % BK || !splits || !slab_stride) return MMT_ERR_ARG;
  const int ksteps = K / BK;
  int sp = splits_requested <= 0 ? 16 : splits_requested;
  if (sp > ksteps) sp = ksteps;
  if (sp > 16) sp = 16;
  const int per = (ksteps + sp - 1) / sp;
  *splits = (ksteps + per - 1) / per;
  *slab_stride = (int64_t)((M + 127) / 128 * 128) * N;
  return 0;
}

extern "C" int64_t mmt_gemm_nt_splitk_workspace_floats(int M, int N, int K) {
  const int splits = K / BK < 16 ? K / BK : 16;
  return (int64_t)(splits < 1 ? 1 : splits) * ((M + 127) / 128 * 128) * N;
}

template <int BM, int BN, int WGM, int WGN, int NS, bool BKN = false>
static int launch_splitk(const bf16_t* A, int64_t lda, const bf16_t* B, int64_t ldb, float* ws, int64_t slab, int M, int Mpad,
                         int N, int K, int splits, int per, const int32_t* nr, const MmtEpilogue& e, hipStream_t s) {
  constexpr size_t lds = (size_t)NS * (BM + BN) * BK * 2;
  constexpr int NT = WGM * WGN * 64;
  static bool configured = false;
  if (!configured) {
    hipError_t rc = hipFuncSetAttribute((const void*)gemm2_splitk_kernel<BM, BN, WGM, WGN, NS, BKN>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (rc != hipSuccess) return (int)rc;
    configured = true;
  }
  int gx = ((Mpad + BM - 1) / BM) * (N / BN), slot = 0;
  const void* rider = e.rider;
  if (rider) {  // (as launch2: extra blocks -- per K-slice row of the grid -- that run optimizer queue entries)
    if (NT % 256 || lds < (size_t)adam_rider_lds_bytes<(NT % 256 ? 256 : NT)>()) {
      rider = nullptr;
    } else {
      const int by_lds = (int)((size_t)160 * 1024 / lds), by_threads = 2048 / NT;
      const int per_cu = by_lds < by_threads ? (by_lds < 1 ? 1 : by_lds) : by_threads;
      slot = (e.rider_slot & 0xffff) | ((256 * per_cu) << 16);
      const int cap = (e.rider_cap & 0xfff) > 0 ? (e.rider_cap & 0xfff) : MMT_RIDER_CAP;
      gx += ((cap < 256 * per_cu ? cap : 256 * per_cu) + splits - 1) / splits;
    }
  }
  hipLaunchKernelGGL((gemm2_splitk_kernel<BM, BN, WGM, WGN, NS, BKN>), dim3(gx, splits), dim3(NT),
                     lds, s, A, lda, B, ldb, ws, slab, (int64_t)N, M, N, K, per * BK, nr, rider, e.rider_limit, slot, e.rider_cap);
  return 0;
}

// epilogue: MMT_EPI_BF16 / F32 / BIAS_F32 / ADD_F32 / BIAS_DROP_RES.  ws: mmt_gemm_nt_splitk_workspace_floats() floats
// (slab s = ws + s * round_up(M,128) * N, leading dimension N).
// splits <= 0: as many as there are K-steps, at most 16 (skinny problems).  wide: 128x128 tiles (N % 128 == 0) instead of
// 128x64.  n_rows_dev (nullable): device count of live rows (token packing).  no_epilogue: leave the partial slabs for a
// consumer that sums them itself (mmt_ln_fwd / mmt_ln_bwd with a slab source); C and epi are then unused.
static int splitk_impl(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int M, int N, int K,
                       int epilogue, const MmtEpilogue* epi, float* ws, int splits, int wide, const int32_t* n_rows_dev,
                       int no_epilogue, bool b_kn, void* stream) {
  if (!A || !B || (!C && !no_epilogue) || !ws || M <= 0 || N <= 0 || K <= 0 || K % BK || N % 64 || (wide && N % 128))
    return MMT_ERR_ARG;
  if ((lda % 8) || (ldb % 8) || (ldc % 4) || ((uintptr_t)A & 15) || ((uintptr_t)B & 15) || ((uintptr_t)C & 15))
    return MMT_ERR_ALIGN;
  MmtEpilogue e = {};
  if (epi) e = *epi;
  // the row-dot sums exist in the bf16 epilogue only and need their partner matrix
  if (e.dot_out && !no_epilogue && (!e.dot_src || epilogue != MMT_EPI_BF16)) return MMT_ERR_ARG;
  const int ksteps = K / BK;
  if (splits <= 0) splits = 16;
  if (splits > ksteps) splits = ksteps;
  if (splits > 16) splits = 16;
  const int per = (ksteps + splits - 1) / splits;       // K-steps per slice
  splits = (ksteps + per - 1) / per;
  const int Mpad = (M + 127) / 128 * 128;
  const int64_t slab = (int64_t)Mpad * N;
  hipStream_t s = (hipStream_t)stream;
  const bf16_t *a = (const bf16_t*)A, *b = (const bf16_t*)B;
  int rc;
  if (b_kn)
    rc = wide ? launch_splitk<128, 128, 2, 4, 2, true>(a, lda, b, ldb, ws, slab, M, Mpad, N, K, splits, per, n_rows_dev, e, s)
              : launch_splitk<128, 64, 4, 2, 3, true>(a, lda, b, ldb, ws, slab, M, Mpad, N, K, splits, per, n_rows_dev, e, s);
  else if (wide == 2)  // 256x128 tiles: half the L2 -> LDS bytes per MAC of the 128x128 tile (the N = 512 GEMMs re-read
                       // their operands from L2 ~10x; at ~15 TB/s aggregate that traffic is what bounds them)
    rc = launch_splitk<256, 128, 4, 2, 3>(a, lda, b, ldb, ws, slab, M, Mpad, N, K, splits, per, n_rows_dev, e, s);
  else
    rc = wide ? launch_splitk<128, 128, 2, 4, 2>(a, lda, b, ldb, ws, slab, M, Mpad, N, K, splits, per, n_rows_dev, e, s)
              : launch_splitk<128, 64, 4, 2, 3>(a, lda, b, ldb, ws, slab, M, Mpad, N, K, splits, per, n_rows_dev, e, s);
  if (rc) return rc;
  if (no_epilogue) return (int)hipGetLastError();
  const int64_t items = (int64_t)M * (N / 4);
  const int grid = (int)((items + 255) / 256 < 2048 ? (items + 255) / 256 : 2048);
#define SK_EPI(E) hipLaunchKernelGGL(splitk_epilogue_kernel<E>, dim3(grid), dim3(256), 0, s, ws, slab, (int64_t)N, splits, C, ldc, M, N, e, n_rows_dev)
  switch (epilogue) {
    case MMT_EPI_BF16: SK_EPI(MMT_EPI_BF16); break;
    case MMT_EPI_F32: SK_EPI(MMT_EPI_F32); break;
    case MMT_EPI_BIAS_F32: if (!e.bias) return MMT_ERR_ARG; SK_EPI(MMT_EPI_BIAS_F32); break;
    case MMT_EPI_ADD_F32: if (!e.res) return MMT_ERR_ARG; SK_EPI(MMT_EPI_ADD_F32); break;
    case MMT_EPI_BIAS_DROP_RES: if (!e.bias || !e.res) return MMT_ERR_ARG; SK_EPI(MMT_EPI_BIAS_DROP_RES); break;
    default: return MMT_ERR_ARG;
  }
#undef SK_EPI
  return (int)hipGetLastError();
}

extern "C" int mmt_gemm_nt_splitk_ex(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int M,
                                     int N, int K, int epilogue, const MmtEpilogue* epi, float* ws, int splits, int wide,
                                     const int32_t* n_rows_dev, int no_epilogue, void* stream) {
  return splitk_impl(A, lda, B, ldb, C, ldc, M, N, K, epilogue, epi, ws, splits, wide, n_rows_dev, no_epilogue, false, stream);
}
// the same with B given as [K, N] row-major (see mmt_gemm_nn_bf16)
extern "C" int mmt_gemm_nn_splitk_ex(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int M,
                                     int N, int K, int epilogue, const MmtEpilogue* epi, float* ws, int splits, int wide,
                                     const int32_t* n_rows_dev, int no_epilogue, void* stream) {
  return splitk_impl(A, lda, B, ldb, C, ldc, M, N, K, epilogue, epi, ws, splits, wide, n_rows_dev, no_epilogue, true, stream);
}

extern "C" int mmt_gemm_nt_splitk(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int M, int N,
                                  int K, int epilogue, const MmtEpilogue* epi, float* ws, void* stream) {
  return mmt_gemm_nt_splitk_ex(A, lda, B, ldb, C, ldc, M, N, K, epilogue, epi, ws, 0, 0, nullptr, 0, stream);
}

// Several independent small GEMMs (the per-expert ReduceDim projections, model/model.py:426-437) in ONE launch:
// block -> (problem, tile) through a prefix table; each problem keeps its own XCD-aware tile order.
struct GemmGroupTable { MmtGemmItem item[MMT_GEMM_GROUP_MAX]; int count; };
template <int BM, int BN, int WGM, int WGN, int NS, int EPI>
__global__ __launch_bounds__(WGM * WGN * 64) void gemm2_grouped_kernel(GemmGroupTable tab) {
  int p = 0;
#pragma unroll 1
  for (int q = 1; q < tab.count; ++q)
    if ((int)blockIdx.x >= tab.item[q].tile_begin) p = q;
  const MmtGemmItem& it = tab.item[p];
  MmtEpilogue e = {};
  e.bias = it.bias;
  const int nblk = ((it.M + BM - 1) / BM) * (it.N / BN);
  gemm2_body<BM, BN, WGM, WGN, NS, EPI>((const bf16_t*)it.A, it.lda, (const bf16_t*)it.B, it.ldb, it.C, it.ldc, it.M, it.N,
                                        it.K, e, it.n_rows_dev, (int)blockIdx.x - it.tile_begin, nblk);
}

extern "C" int mmt_gemm_nt_grouped(const MmtGemmItem* items, int n, int epilogue, void* stream) {
  if (!items || n <= 0 || n > MMT_GEMM_GROUP_MAX) return MMT_ERR_ARG;
  if (epilogue != MMT_EPI_BIAS_F32 && epilogue != MMT_EPI_F32) return MMT_ERR_ARG;
  constexpr int BM = 128, BN = 64, WGM = 4, WGN = 2, NS = 3;
  GemmGroupTable tab;
  tab.count = n;
  int tiles = 0;
  for (int i = 0; i < n; ++i) {
    MmtGemmItem it = items[i];
    if (!it.A || !it.B || !it.C || it.M <= 0 || it.N <= 0 || it.K <= 0 || it.K % BK || it.N % BN) return MMT_ERR_ARG;
    if (epilogue == MMT_EPI_BIAS_F32 && !it.bias) return MMT_ERR_ARG;
    if ((it.lda % 8) || (it.ldb % 8) || (it.ldc % 4) || ((uintptr_t)it.A & 15) || ((uintptr_t)it.B & 15) ||
        ((uintptr_t)it.C & 15))
      return MMT_ERR_ALIGN;
    it.tile_begin = tiles;
    tiles += ((it.M + BM - 1) / BM) * (it.N / BN);
    tab.item[i] = it;
  }
  constexpr size_t lds = (size_t)NS * (BM + BN) * BK * 2;
  static bool configured = false;
  if (!configured) {
    hipError_t rc = hipFuncSetAttribute((const void*)gemm2_grouped_kernel<BM, BN, WGM, WGN, NS, MMT_EPI_BIAS_F32>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (rc == hipSuccess)
      rc = hipFuncSetAttribute((const void*)gemm2_grouped_kernel<BM, BN, WGM, WGN, NS, MMT_EPI_F32>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (rc != hipSuccess) return (int)rc;
    configured = true;
  }
  if (epilogue == MMT_EPI_BIAS_F32)
    hipLaunchKernelGGL((gemm2_grouped_kernel<BM, BN, WGM, WGN, NS, MMT_EPI_BIAS_F32>), dim3(tiles), dim3(512), lds,
                       (hipStream_t)stream, tab);
  else
    hipLaunchKernelGGL((gemm2_grouped_kernel<BM, BN, WGM, WGN, NS, MMT_EPI_F32>), dim3(tiles), dim3(512), lds,
                       (hipStream_t)stream, tab);
  return (int)hipGetLastError();
}

template <int BM, int BN, int WGM, int WGN, int NS, int EPI, bool BKN = false, bool PH = false>
static int launch2(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int M, int N, int K,
                   const MmtEpilogue& e, const int32_t* nr, hipStream_t s) {
  constexpr int NT = (PH ? 2 : 1) * WGM * WGN * 64;
  constexpr int CB = (NT * 4) % BN == 0 ? BN : 64;
  constexpr int RG = NT / (CB / 4);
  constexpr int CH = BM < 64 ? BM : 64;
  static_assert(CH % RG == 0 && (BM + BN) % (8 * WGM * WGN) == 0 && BM % (8 * WGM * WGN) == 0 && BN % (8 * WGM * WGN) == 0,
                "tile geometry");
  constexpr size_t stage_bytes = (size_t)NS * (BM + BN) * BK * 2;
  constexpr size_t epi_bytes = (size_t)(CH * (BN + 4) + RG * BN + MMT_GELU_LUT_N) * 4;  // image + column sums + GELU table
  constexpr size_t lds = stage_bytes > epi_bytes ? stage_bytes : epi_bytes;
  static bool configured = false;
  if (!configured) {
    hipError_t rc = hipFuncSetAttribute((const void*)gemm2_kernel<BM, BN, WGM, WGN, NS, EPI, BKN, PH>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (rc != hipSuccess) return (int)rc;
    configured = true;
  }
  // (N <= 1024 with K >= 2048: eight blocks past the last tile, which exit at once -- the FFN down-projection / its input
  // gradient then differ from the K = hidden GEMMs of the same template in their GRID, so that a profile can tell them apart)
  int grid = ((M + BM - 1) / BM) * (N / BN) + (N <= 1024 && K >= 2048 ? 8 : 0);
  // r06: with an optimizer queue attached (MmtEpilogue.rider) the launch gets one residency round of extra blocks: they (and
  // the tiles past the live row count) run queue entries while the tiles compute; the slot word tells them how many blocks of
  // this launch are resident at once, i.e. where its last round begins (adam_unit.h: adam_rider_run)
  MmtEpilogue e2 = e;
  if (e.rider) {
    if (NT % 256 || lds < (size_t)adam_rider_lds_bytes<(NT % 256 ? 256 : NT)>()) {
      e2.rider = nullptr;
    } else {
      const int by_lds = (int)((size_t)160 * 1024 / lds), by_threads = 2048 / NT;
      const int per_cu = by_lds < by_threads ? (by_lds < 1 ? 1 : by_lds) : by_threads;
      e2.rider_slot = (e.rider_slot & 0xffff) | ((256 * per_cu) << 16);
      const int cap = (e.rider_cap & 0xfff) > 0 ? (e.rider_cap & 0xfff) : MMT_RIDER_CAP;
      grid += cap < 256 * per_cu ? cap : 256 * per_cu;
    }
  }
  hipLaunchKernelGGL((gemm2_kernel<BM, BN, WGM, WGN, NS, EPI, BKN, PH>), dim3(grid), dim3(NT), lds, s, (const bf16_t*)A, lda,
                     (const bf16_t*)B, ldb, C, ldc, M, N, K, e2, nr);
  return (int)hipGetLastError();
}

template <int EPI>
static int pick2(int tile, const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int M, int N,
                 int K, const MmtEpilogue& e, const int32_t* nr, hipStream_t s) {
#define G2(BM_, BN_, WGM_, WGN_, NS_) \
  return launch2<BM_, BN_, WGM_, WGN_, NS_, EPI>(A, lda, B, ldb, C, ldc, M, N, K, e, nr, s)
  if (EPI == MMT_EPI_DGELU && e.colsum && (tile & 0xff) == 12) return MMT_ERR_ARG;  // 64-row tiles: no column sums
  // The tiles the dispatcher selects on its own (gemm.hip: dispatch_tile): 13, 14, 18 (+ 21 = gemm3.hip, 24 = gemm5.hip).
  // Everything else was measured and lost and is compiled into the LAB library only
  // (python -m mmt_amd.build --lab: -DMMT_LAB_TILES), where tools/gemm_lab.py and the MMT_TILE_* switches reach it.
  switch (tile & 0xff) {
    case 13: if (N % 64 == 0) G2(128, 64, 4, 2, 3); break;    // 8 waves on 128x64 (wave 32x32), staggered, 2 blocks/CU
    case 14: if (N % 128 == 0) G2(128, 128, 2, 4, 2); break;  // 8 waves on 128x128, in phase, 2 blocks/CU
    case 18:  // 2 x 4 waves on 128x64, PHASED: the groups take alternate K-steps (wave tile 64x32), 1-2 blocks/CU
      if (N % 64 == 0) return launch2<128, 64, 2, 2, 4, EPI, false, true>(A, lda, B, ldb, C, ldc, M, N, K, e, nr, s);
      break;
#ifdef MMT_LAB_TILES
    case 3: if (N % 128 == 0) G2(256, 128, 4, 2, 3); break;   // 8 waves, staggered
    case 4: if (N % 256 == 0) G2(256, 256, 4, 2, 2); break;   // 8 waves
    case 5: if (N % 128 == 0) G2(128, 128, 2, 2, 2); break;   // 4 waves, 2 blocks/CU
    case 7: if (N % 64 == 0) G2(128, 64, 2, 2, 3); break;     // 4 waves
    case 10: if (N % 128 == 0) G2(256, 128, 4, 2, 2); break;  // 8 waves, in phase
    case 11: if (N % 128 == 0) G2(128, 128, 2, 4, 3); break;  // 8 waves on 128x128 (wave 64x32), staggered
    case 12: if (N % 128 == 0) G2(64, 128, 2, 4, 3); break;   // 8 waves on 64x128 (wave 32x32), staggered, 2 blocks/CU
    case 15: if (N % 192 == 0) G2(256, 192, 4, 2, 2); break;  // 8 waves on 256x192 (wave 64x96), 1 block/CU: N = 3072 at
                                                              // <= 4096 live rows is ONE round of <= 256 tiles
    case 16: if (N % 192 == 0) G2(128, 192, 4, 2, 3); break;  // 8 waves on 128x192 (wave 32x96), staggered
    case 17: if (N % 192 == 0) G2(128, 192, 2, 2, 3); break;  // 4 waves on 128x192 (wave 64x96)
    case 22:  // tile 18 with a SIX-deep ring (144 KiB): every group keeps two of its own K-steps in flight beside the one it waits for
      if (N % 64 == 0) return launch2<128, 64, 2, 2, 6, EPI, false, true>(A, lda, B, ldb, C, ldc, M, N, K, e, nr, s);
      break;
    case 20: if (N % 192 == 0) G2(128, 192, 4, 2, 2); break;  // 8 waves on 128x192 (wave 32x96), in phase, TWO-deep ring:
                                                              // 80 KiB of LDS = two blocks per CU (tile 16's three stages allow one)
    case 19:  // 2 x 4 waves on 128x128, PHASED (wave tile 64x64)
      if (N % 128 == 0) return launch2<128, 128, 2, 2, 4, EPI, false, true>(A, lda, B, ldb, C, ldc, M, N, K, e, nr, s);
      break;
#endif
  }
#undef G2
  return MMT_ERR_ARG;
}

int mmt_gemm3_dispatch(int epilogue, const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int M, int N,
                       int K, const MmtEpilogue& e, const int32_t* nr, hipStream_t s);
int mmt_gemm4_dispatch(int epilogue, const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int M, int N,
                       int K, const MmtEpilogue& e, const int32_t* nr, hipStream_t s);
int mmt_gemm5_dispatch(int epilogue, int bn, const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int M,
                       int N, int K, const MmtEpilogue& e, const int32_t* nr, hipStream_t s);

// tile: 3 = 256x128, 4 = 256x256, 5 = 128x128, 6 = 128x256 (see MmtEpilogue.reserved); 21 = the 256x256 eight-phase kernel
// of gemm3.hip
int mmt_gemm2_dispatch(int tile, int epilogue, const void* A, int64_t lda, const void* B, int64_t ldb, void* C,
                       int64_t ldc, int M, int N, int K, const MmtEpilogue& e, const int32_t* nr, hipStream_t s) {
  if ((tile & 0xff) == 21) return mmt_gemm3_dispatch(epilogue, A, lda, B, ldb, C, ldc, M, N, K, e, nr, s);
  if ((tile & 0xff) == 24) {  // persistent, wave-specialised 128 x 128 (gemm5.hip) where its geometry allows, else tile 14 / 13
    const int rc = mmt_gemm5_dispatch(epilogue, 128, A, lda, B, ldb, C, ldc, M, N, K, e, nr, s);
    if (rc != MMT_ERR_ARG) return rc;
    tile = (N % 128 == 0 && !(epilogue == MMT_EPI_DGELU && e.colsum && M < 128)) ? 14 : 13;
  }
  if ((tile & 0xff) == 25) {  // the same kernel on 128 x 64 tiles (long K, narrow outputs), else tile 13
    const int rc = mmt_gemm5_dispatch(epilogue, 64, A, lda, B, ldb, C, ldc, M, N, K, e, nr, s);
    if (rc != MMT_ERR_ARG) return rc;
    tile = 13;
  }
#ifdef MMT_LAB_TILES
  if ((tile & 0xff) == 23) return mmt_gemm4_dispatch(epilogue, A, lda, B, ldb, C, ldc, M, N, K, e, nr, s);  // producer / consumer 128 x 64 (gemm4.hip)
#endif
  switch (epilogue) {
    case MMT_EPI_BF16: return pick2<MMT_EPI_BF16>(tile, A, lda, B, ldb, C, ldc, M, N, K, e, nr, s);
    case MMT_EPI_BIAS_BF16: return pick2<MMT_EPI_BIAS_BF16>(tile, A, lda, B, ldb, C, ldc, M, N, K, e, nr, s);
    case MMT_EPI_BIAS_GELU: return pick2<MMT_EPI_BIAS_GELU>(tile, A, lda, B, ldb, C, ldc, M, N, K, e, nr, s);
    case MMT_EPI_BIAS_DROP_RES: return pick2<MMT_EPI_BIAS_DROP_RES>(tile, A, lda, B, ldb, C, ldc, M, N, K, e, nr, s);
    case MMT_EPI_DGELU: return pick2<MMT_EPI_DGELU>(tile, A, lda, B, ldb, C, ldc, M, N, K, e, nr, s);
    case MMT_EPI_ADD_F32: return pick2<MMT_EPI_ADD_F32>(tile, A, lda, B, ldb, C, ldc, M, N, K, e, nr, s);
    case MMT_EPI_F32: return pick2<MMT_EPI_F32>(tile, A, lda, B, ldb, C, ldc, M, N, K, e, nr, s);
    case MMT_EPI_BIAS_F32: return pick2<MMT_EPI_BIAS_F32>(tile, A, lda, B, ldb, C, ldc, M, N, K, e, nr, s);
  }
  return MMT_ERR_ARG;
}

// ---- NN form: C[M,N] = A[M,K] . B[K,N]  (B row-major [K, N]: a weight matrix W[out = K, in = N] as stored) ----------
// The input-gradient GEMMs of every nn.Linear on the path (dX = dY . W; reference autograd of model/bert.py:146-150,
// 186, 218, 234) without a transposed weight copy.  Epilogues: BF16, F32, ADD_F32, DGELU.  Tiles as the NT form:
// 128x128 / 8 waves for wide outputs, 128x64 / 8 staggered waves otherwise and for short batches.
template <int EPI>
static int pick_nn(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int M, int N, int K,
                   const MmtEpilogue& e, const int32_t* nr, hipStream_t s) {
  const int forced = e.reserved & 0xff;
  const bool wide = forced ? forced == 14 : (M > 1024 && N >= 1024 && N % 128 == 0);
  if (wide) return launch2<128, 128, 2, 4, 2, EPI, true>(A, lda, B, ldb, C, ldc, M, N, K, e, nr, s);
  return launch2<128, 64, 4, 2, 3, EPI, true>(A, lda, B, ldb, C, ldc, M, N, K, e, nr, s);
}

extern "C" int mmt_gemm_nn_bf16(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int M, int N,
                                int K, int epilogue, const MmtEpilogue* epi, const int32_t* n_rows_dev, void* stream) {
  if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0 || K % BK || N % 64) return MMT_ERR_ARG;
  if ((lda % 8) || (ldb % 8) || (ldc % 4) || ((uintptr_t)A & 15) || ((uintptr_t)B & 15) || ((uintptr_t)C & 15))
    return MMT_ERR_ALIGN;
  MmtEpilogue e = {};
  if (epi) e = *epi;
  hipStream_t s = (hipStream_t)stream;
  switch (epilogue) {
    case MMT_EPI_BF16: return pick_nn<MMT_EPI_BF16>(A, lda, B, ldb, C, ldc, M, N, K, e, n_rows_dev, s);
    case MMT_EPI_F32: return pick_nn<MMT_EPI_F32>(A, lda, B, ldb, C, ldc, M, N, K, e, n_rows_dev, s);
    case MMT_EPI_ADD_F32:
      if (!e.res) return MMT_ERR_ARG;
      return pick_nn<MMT_EPI_ADD_F32>(A, lda, B, ldb, C, ldc, M, N, K, e, n_rows_dev, s);
    case MMT_EPI_DGELU:
      if (!e.aux || e.colsum) return MMT_ERR_ARG;
      return pick_nn<MMT_EPI_DGELU>(A, lda, B, ldb, C, ldc, M, N, K, e, n_rows_dev, s);
  }
  return MMT_ERR_ARG;
}
