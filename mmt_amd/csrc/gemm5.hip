// Persistent wave-specialised NT GEMM on gfx950, written for the wide K = hidden GEMMs (QKV, FFN up-projection + GELU, dGELU)
// (tile id 24 of the dispatcher):   C[M,N] = A[M,K] . B[N,K]^T (+ the fused epilogues of gemm2), 128 x 128 tiles.
//
// Why: profiles/r04_gemm2_budget.txt -- at 3.6 k live rows x N = 3072 x K = 512 a 128x128 block of gemm2.hip is prologue
// 2.2 k + K-loop 15.6 k (8 K-steps) + epilogue 8.8 k cycles, two such blocks per CU overlap only partly (1.47 resident on
// average), 696 tiles on 512 slots leave a 12 % tail round, and inside the K-loop the waves that run the MFMAs also issue
// the LDS-DMA requests (two-deep ring: a step cannot be shorter than issue + latency).  Here ONE block per CU stays for the
// whole launch and its twelve waves have three roles:
//   * waves 8..11  PRODUCERS: nothing but LDS-DMA (buffer_load ... lds, descriptor + fixed per-lane offset + scalar K
//     offset) into a FOUR-deep ring of 32 KiB stages (A 128 x 64, B 128 x 64), two stages in flight, counted vmcnt;
//     the stream of stages runs on across tile boundaries, so a tile's first K-steps arrive under the previous tile's last;
//   * waves 0..3 and 4..7  two CONSUMER GROUPS (one wave of each per SIMD, wave tile 64 x 64 = 2 x 2 fragments of
//     mfma_f32_32x32x16_bf16) that take ALTERNATE tiles: while one group runs the K-loop of tile i (only fragment reads
//     and MFMAs), the other runs the epilogue of tile i - 1 -- the VALU-bound sweep (bias / GELU table / bf16 packing /
//     stores) hides under the other group's matrix work instead of following it;
//   * one workgroup barrier per K-step ("tick") hands a landed stage to the consumers and a consumed one back to the
//     producers.  The epilogue needs no barrier of its own: every wave stages ITS 64 x 64 sub-tile through a private LDS
//     region (32 rows x 32 columns at a time) and sweeps it itself, so the epilogue group simply executes the tick barriers
//     at evenly spaced points of its instruction stream.
// With ONE wave per SIMD in the K-loop nothing but its own instruction order hides LDS latency, so a tick is one asm
// statement: every MFMA is followed by one fragment read, TWO k-sub-steps ahead of the sub-step being multiplied, the last
// two sub-steps read the first two of the NEXT stage (which is why the ring is four deep: a stage has landed a whole tick
// before its tick), and the fragment addresses advance in the MFMA shadows (g5_tick below; budgets:
// profiles/r05_g5_budget*.txt).  Blocks walk the live tiles in an XCD-aware order (g5_tile / g5_own below): bands of eight
// tile rows for wide outputs, row by row for narrow ones, the last partial round of tiles spread over all eight XCDs.
//
// The same kernel serves the long-K GEMMs with narrow outputs (N = 512, K = 1536 / 3072: FFN down-projection, input
// gradients): on 128 x 128 tiles where those fill a round of the chip, and as a second instantiation on 128 x 64 tiles
// (tile id 25: wave tile 64 x 32, five-deep ring of 24 KiB stages, six producer waves, sixteen waves per block -- G5Lay<64>)
// for row counts where they do not.  What bounds these inside a training step is not this file: with the operands coming
// from Infinity Cache / HBM a CU gets ~18 B/clk whatever the ring depth, the number of producers or the tile order
// (profiles/r05_g5_narrow_budget_cold.txt; DESIGN.md section 7, "r05 (late)").
#include <type_traits>
#include "mmt_common.h"
#include "../../include/mmt_hip.h"
#include "gemm_epi.h"
#include "g5_own.h"

#define G5_RING 4
#ifndef G5_LAB_AUX_A   // lab: cache-policy bits of the LDS-DMA requests (1 sc0, 2 nt, 16 sc1) for the A / B operand
#define G5_LAB_AUX_A 0
#endif
#ifndef G5_LAB_AUX_B
#define G5_LAB_AUX_B 0
#endif
#define G5_HALF 16384                                    // one operand of a stage: 128 rows x 128 B
// LDS: [A slots 0..3 | GELU table | epilogue images | column-sum scratch | B slots 0..3].  The table sits at 64 KiB so that
// (table - 4 * MMT_GELU_LUT_LO) fits the 16-bit ds offset field (a gather is then med3 + shift + ds_read_b32).
#define G5_LUT_OFF (G5_RING * G5_HALF)
#define G5_ST_OFF (G5_LUT_OFF + MMT_GELU_LUT_N * 4)      // wave-private epilogue images: 4 waves x 32 rows x 36 floats (only one
                                                         // group is inside an epilogue at any time: the groups share them)
#define G5_ST_P 36
#define G5_ST_WAVE (32 * G5_ST_P * 4)
#define G5_RED_OFF (G5_ST_OFF + 4 * G5_ST_WAVE)          // [2 groups][2 wave rows][128] column sums (DGELU colsum)
#define G5_BOFF (G5_RED_OFF + 2 * 2 * 128 * 4)
#define G5_LDS (G5_BOFF + G5_RING * G5_HALF)

static_assert(G5_LDS <= 163840 && G5_BOFF % 1024 == 0, "gemm5: LDS budget / alignment");
// The 128 x 64 tile (a stage = 16 KiB of A + 8 KiB of B) runs a FIVE-deep ring: what bounds a long-K GEMM whose operands come
// from HBM / Infinity Cache is the bytes a CU keeps in flight (ring - 2 stages: r05 budgets, 1.39 k cycles per K-step cold
// with two 24 KiB stages in flight against 0.79 k warm).  LDS: [A slots 0..4 | B slots 0..4], both 16 KiB apart (one address
// step for all fragment pointers); the unused upper 8 KiB of B slots 0..3 hold the four wave-private epilogue images.  No
// GELU table, no column sums (the host refuses those epilogues for this tile).  10 x 16 KiB = all of a CU's LDS.
template <int BN> struct G5Lay {
  static constexpr int RING = BN == 64 ? 5 : 4;
  static constexpr int BOFF = BN == 64 ? RING * G5_HALF : G5_BOFF;
  static constexpr int LDS = BN == 64 ? 2 * RING * G5_HALF : G5_LDS;
  static constexpr int ST_OFF = BN == 64 ? BOFF + G5_HALF / 2 : G5_ST_OFF;
  static constexpr int ST_WAVE = BN == 64 ? G5_HALF : G5_ST_WAVE;
  // producer waves.  A wave gets its requests answered at ~1 piece per 120 cycles when the operands come from far memory
  // (r05 cold budgets: 8 pieces per ~1000 cycles per wave, with three waves as with four), so the narrow tile -- whose
  // consumers need < 128 registers -- spreads a stage over six waves of four pieces (sixteen waves per block).
  static constexpr int NP = BN == 64 ? 8 : 4;
  static constexpr int THREADS = 512 + 64 * NP;
};
static_assert(G5Lay<64>::LDS <= 163840 && G5_ST_WAVE <= G5_HALF / 2, "gemm5: LDS budget of the 128 x 64 tile");
template <int N> __device__ __forceinline__ void g5_vmwait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void g5_barrier() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

#ifdef MMT_G5_INSTR
#define G5_TICK(acc) do { const long long n_ = clock64(); acc += n_ - g5p; g5p = n_; } while (0)
#else
#define G5_TICK(acc) do {} while (0)
#endif

// The epilogue of one wave's 64 x 64 sub-tile at (row0, col0).  nbar tick barriers are spread evenly over its 16 sweep
// steps (0: the launch's last epilogue -- nobody is left to meet).
//   * The sub-tile goes through the wave's LDS image one 32 x 32 fragment at a time (4 passes); a pass is swept in 4 steps
//     of 8 rows (8 lanes x 4 columns = a 64-byte bf16 row segment per row).  (16 rows x 64 columns per pass -- whole
//     128-byte lines per row -- was measured slower: half the lanes idle in the staging writes; r05 lab)
//   * ONE wave per SIMD runs it, so nothing but the code itself hides LDS latency: the sweep goes in PAIRS of steps, the
//     image rows of the next pair are requested before the current pair is worked on (also across a tick barrier), and a
//     pair is one basic block, so the compiler batches its eight GELU-table gathers.
//   * Global accesses are MUBUF with a descriptor cut at the end of the matrix: rows past M are dropped (stores) / read as
//     zero (loads) by the hardware's range check -- no predicates, no 64-bit address arithmetic (a per-lane 32-bit byte
//     offset + a scalar step).  The host refuses matrices whose byte offsets do not fit 31 bits.
template <int EPI, int NJ>
__device__ __forceinline__ void g5_epilogue(f32x16 (&acc)[2][NJ], unsigned char* smem_raw, int row0, int col0, int grp, int w4,
                                            int lane, int M, int N, int nrows, void* __restrict__ Cout, int ldc,
                                            const MmtEpilogue& epi, int nbar, long long& t_work, long long& t_bar) {
  constexpr int P = G5_ST_P;
  constexpr bool OUT16 = EPI == MMT_EPI_BF16 || EPI == MMT_EPI_BIAS_BF16 || EPI == MMT_EPI_BIAS_GELU || EPI == MMT_EPI_DGELU;
  constexpr int ESZ = OUT16 ? 2 : 4;
  constexpr bool RES = EPI == MMT_EPI_BIAS_DROP_RES || EPI == MMT_EPI_ADD_F32;
  constexpr bool BIAS = EPI == MMT_EPI_BIAS_BF16 || EPI == MMT_EPI_BIAS_GELU || EPI == MMT_EPI_BIAS_DROP_RES || EPI == MMT_EPI_BIAS_F32;
  float* st = (float*)(smem_raw + G5Lay<32 * NJ * 2>::ST_OFF + w4 * G5Lay<32 * NJ * 2>::ST_WAVE);
  const float* lut = (const float*)(smem_raw + G5_LUT_OFF);
  const int l31 = lane & 31, lh = lane >> 5;
  const int rq = lane >> 3, c4 = (lane & 7) * 4;  // sweep: row 8 t + rq of the 32-row image, columns c4 .. c4 + 3 of its 32
#ifdef MMT_G5_INSTR
  long long g5p = clock64();
#endif
  constexpr int FLAGS = 0x00020000;
  const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(Cout, 0, ((M - 1) * ldc + N) * ESZ, FLAGS);
  // byte offset of (row0 + rq, col0 + c4) in C and in the second matrix of the epilogue (GELU output / residual /
  // pre-activations); pass (i, j), step t adds (32 i + 8 t) rows and 32 j columns
  const unsigned off_c = (unsigned)(((row0 + rq) * ldc + col0 + c4) * ESZ);
  const unsigned row8_c = (unsigned)(8 * ldc * ESZ);
  __amdgpu_buffer_rsrc_t r2 = rc;
  unsigned off_2 = 0, row8_2 = 0;
  constexpr int ESZ2 = RES ? 4 : 2;
  if constexpr (EPI == MMT_EPI_BIAS_GELU) {
    r2 = __builtin_amdgcn_make_buffer_rsrc(epi.out2, 0, ((M - 1) * (int)epi.ldout2 + N) * 2, FLAGS);
    off_2 = (unsigned)(((row0 + rq) * (int)epi.ldout2 + col0 + c4) * 2); row8_2 = (unsigned)(8 * (int)epi.ldout2 * 2);
  } else if constexpr (RES) {
    r2 = __builtin_amdgcn_make_buffer_rsrc((void*)epi.res, 0, ((M - 1) * (int)epi.ldres + N) * 4, FLAGS);
    off_2 = (unsigned)(((row0 + rq) * (int)epi.ldres + col0 + c4) * 4); row8_2 = (unsigned)(8 * (int)epi.ldres * 4);
  } else if constexpr (EPI == MMT_EPI_DGELU) {
    r2 = __builtin_amdgcn_make_buffer_rsrc((void*)epi.aux, 0, ((M - 1) * (int)epi.ldaux + N) * 2, FLAGS);
    off_2 = (unsigned)(((row0 + rq) * (int)epi.ldaux + col0 + c4) * 2); row8_2 = (unsigned)(8 * (int)epi.ldaux * 2);
  }
  f32x4 bias4[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  if constexpr (BIAS) {
    bias4[0] = *(const f32x4*)(epi.bias + col0 + c4);
    if constexpr (NJ == 2) bias4[1] = *(const f32x4*)(epi.bias + col0 + 32 + c4);
  }
  unsigned dkey = 0;
  // dropout: the original row numbers (RNG coordinate) of this wave's 8 sweep rows per lane, requested HERE -- before the
  // first store of the epilogue and without a branch (a null table is a descriptor of zero records): a load behind a
  // condition makes the compiler wait with vmcnt(0) at every use, which waits for every store issued so far as well (r05:
  // 43 us for the FFN down-projection inside the unpacked step against 28 us for the same GEMM without dropout)
  int orow_all[EPI == MMT_EPI_BIAS_DROP_RES ? 8 : 1];
  if constexpr (EPI == MMT_EPI_BIAS_DROP_RES) {
    dkey = eff_key(epi.drop_key, epi.seed_dev);
    const bool use_ri = epi.drop_thr16 != 0 && epi.row_index != nullptr;
    const __amdgpu_buffer_rsrc_t rri =
        __builtin_amdgcn_make_buffer_rsrc(use_ri ? (void*)epi.row_index : (void*)Cout, 0, use_ri ? M * 4 : 0, FLAGS);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int row = row0 + (k >> 2) * 32 + 8 * (k & 3) + rq;
      const int v = (int)__builtin_amdgcn_raw_buffer_load_b32(rri, row * 4, 0, 0);
      orow_all[k] = use_ri ? v : row;
    }
  }
  float csum[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  float dotp[4] = {0.f, 0.f, 0.f, 0.f};  // BF16 + dot_out: the j = 0 half of a row's dot product waits for the j = 1 half
  int done = 0;
  // DGELU: the pre-activations of the whole sub-tile are requested before the first store of this epilogue and waited for
  // once (a wait placed between the stores would wait for the stores too: vmcnt counts both, in order); the second half's go
  // out once the first staging pass has freed its accumulators
  u32x2 pf_aux[EPI == MMT_EPI_DGELU ? 16 : 1];
  auto aux_index = [](int i, int j, int t) { return (i * 2 + j) * 4 + t; };
  if constexpr (EPI == MMT_EPI_DGELU) {
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int t = 0; t < 4; ++t)
        pf_aux[aux_index(0, j, t)] = __builtin_amdgcn_raw_buffer_load_b64(r2, (int)(off_2 + (unsigned)t * row8_2 + (unsigned)(j * 32 * 2)), 0, 0);
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int pass = i * NJ + j;
      // residual rows / original row numbers of this pass: all 4 steps' worth go out before the staging pass
      u32x4 pf_res[RES ? 4 : 1];
      if constexpr (RES) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          pf_res[t] = __builtin_amdgcn_raw_buffer_load_b128(r2, (int)(off_2 + (unsigned)(i * 4 + t) * row8_2 + (unsigned)(j * 32 * ESZ2)), 0, 0);
        }
      }
      // staging: fragment (i, j) of the wave tile -> row-major fp32 image (same-wave LDS traffic is in order)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
        *(f32x4*)(st + l31 * P + 8 * q + 4 * lh) = v;
      }
      if constexpr (EPI == MMT_EPI_DGELU) {
        if (pass == 0) {
#pragma unroll
          for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
            for (int t = 0; t < 4; ++t)
              pf_aux[aux_index(1, jj, t)] =
                  __builtin_amdgcn_raw_buffer_load_b64(r2, (int)(off_2 + (unsigned)(4 + t) * row8_2 + (unsigned)(jj * 32 * 2)), 0, 0);
#pragma unroll
          for (int t = 0; t < 16; ++t) asm volatile("" : "+v"(pf_aux[t]));
        }
      }
      if (pass == 0) {  // the bias has arrived HERE, on every path (else every later use waits vmcnt(0): stores too)
        if constexpr (EPI == MMT_EPI_BIAS_DROP_RES) {
#pragma unroll
          for (int k = 0; k < 8; ++k) asm volatile("" : "+v"(orow_all[k]));
        }
        asm volatile("" : "+v"(bias4[0]) :: "memory");
        if constexpr (NJ == 2) asm volatile("" : "+v"(bias4[1]) :: "memory");
      }
      f32x4 nxt[2] = {*(const f32x4*)(st + rq * P + c4), *(const f32x4*)(st + (8 + rq) * P + c4)};
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        f32x4 cur[2] = {nxt[0], nxt[1]};
        if (p == 0) {  // the next pair's image rows: in flight while this pair is worked on (and across its barrier)
          nxt[0] = *(const f32x4*)(st + (16 + rq) * P + c4);
          nxt[1] = *(const f32x4*)(st + (24 + rq) * P + c4);
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int t = 2 * p + h;                  // step inside the pass: rows 8 t + rq
          const int row = row0 + i * 32 + 8 * t + rq;
          const int col = col0 + j * 32 + c4;
          f32x4 v = cur[h] + bias4[j];
          const int o_c = (int)(off_c + (unsigned)(i * 4 + t) * row8_c + (unsigned)(j * 32 * ESZ));
          const int o_2 = (int)(off_2 + (unsigned)(i * 4 + t) * row8_2 + (unsigned)(j * 32 * ESZ2));
          if constexpr (EPI == MMT_EPI_BF16 || EPI == MMT_EPI_BIAS_BF16) {
            u32x2 o = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
            __builtin_amdgcn_raw_buffer_store_b64(o, rc, o_c, 0, 0);
            if constexpr (EPI == MMT_EPI_BF16) {
              if (epi.dot_out) {  // sum of out * dot_src over the 64 columns of this wave: 8 lanes x 4 columns, two passes
                const u32x2 c = *(const u32x2*)((const bf16_t*)epi.dot_src + (int64_t)min(row, M - 1) * epi.lddot + col);
                float part = bf2f((bf16_t)(o[0] & 0xffff)) * bf2f((bf16_t)(c[0] & 0xffff)) + bf2f((bf16_t)(o[0] >> 16)) * bf2f((bf16_t)(c[0] >> 16)) +
                             bf2f((bf16_t)(o[1] & 0xffff)) * bf2f((bf16_t)(c[1] & 0xffff)) + bf2f((bf16_t)(o[1] >> 16)) * bf2f((bf16_t)(c[1] >> 16));
                part += __shfl_xor(part, 1, 64); part += __shfl_xor(part, 2, 64); part += __shfl_xor(part, 4, 64);
                if (j == 0) dotp[t] = part;
                else if (row < M && (lane & 7) == 0) epi.dot_out[(int64_t)row * (N >> 6) + (col0 >> 6)] = dotp[t] + part;
              }
            }
          } else if constexpr (EPI == MMT_EPI_BIAS_GELU) {
            u32x2 o = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
            __builtin_amdgcn_raw_buffer_store_b64(o, rc, o_c, 0, 0);
            u32x2 g = {pack_bf2(gelu_lut(lut, o[0] & 0xffff), gelu_lut(lut, o[0] >> 16)),
                       pack_bf2(gelu_lut(lut, o[1] & 0xffff), gelu_lut(lut, o[1] >> 16))};
            __builtin_amdgcn_raw_buffer_store_b64(g, r2, o_2, 0, 0);
          } else if constexpr (EPI == MMT_EPI_BIAS_DROP_RES) {
            if (epi.drop_thr16) {
              bool k[4];
              keep4(dkey, (unsigned long long)orow_all[i * 4 + t] * (unsigned)N + (unsigned)col, epi.drop_thr16, k);
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = k[e] ? v[e] * epi.drop_scale : 0.f;
            }
            v += __builtin_bit_cast(f32x4, pf_res[t]);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rc, o_c, 0, 0);
          } else if constexpr (EPI == MMT_EPI_DGELU) {
            const u32x2 a = pf_aux[aux_index(i, j, t)];
            v[0] *= gelu_grad_lut(lut, a[0] & 0xffff);
            v[1] *= gelu_grad_lut(lut, a[0] >> 16);
            v[2] *= gelu_grad_lut(lut, a[1] & 0xffff);
            v[3] *= gelu_grad_lut(lut, a[1] >> 16);
            u32x2 o = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
            __builtin_amdgcn_raw_buffer_store_b64(o, rc, o_c, 0, 0);
            const float live = row < nrows ? 1.f : 0.f;
            csum[j][0] += live * bf2f((bf16_t)(o[0] & 0xffff)); csum[j][1] += live * bf2f((bf16_t)(o[0] >> 16));
            csum[j][2] += live * bf2f((bf16_t)(o[1] & 0xffff)); csum[j][3] += live * bf2f((bf16_t)(o[1] >> 16));
          } else if constexpr (EPI == MMT_EPI_ADD_F32) {
            v += __builtin_bit_cast(f32x4, pf_res[t]);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rc, o_c, 0, 0);
          } else {  // MMT_EPI_F32 / MMT_EPI_BIAS_F32
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rc, o_c, 0, 0);
          }
        }
        if constexpr (EPI == MMT_EPI_DGELU) {
          if (pass == 2 * NJ - 1 && p == 1 && epi.colsum) {  // column sums of this wave's 64 rows -> LDS, in front of the last barrier
            float* red = (float*)(smem_raw + G5_RED_OFF) + grp * 256 + (w4 >> 1) * 128 + (w4 & 1) * 64;
#pragma unroll
            for (int jj = 0; jj < NJ; ++jj) {
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                csum[jj][e] += __shfl_xor(csum[jj][e], 8, 64);
                csum[jj][e] += __shfl_xor(csum[jj][e], 16, 64);
                csum[jj][e] += __shfl_xor(csum[jj][e], 32, 64);
              }
              if (lane < 8) *(f32x4*)(red + jj * 32 + c4) = (f32x4){csum[jj][0], csum[jj][1], csum[jj][2], csum[jj][3]};
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (nbar == 0) g5_barrier();  // (the launch's last epilogue: only this group's waves are left to meet)
          }
        }
        const int target = ((pass * 4 + 2 * p + 2) * nbar) / (8 * NJ);  // tick barriers owed after this pair of steps
        if (done < target) {
          G5_TICK(t_work);
          while (done < target) { g5_barrier(); ++done; }
          G5_TICK(t_bar);
        }
      }
      asm volatile("" ::: "memory");
    }
  }
  G5_TICK(t_work);
}

// One tick of a consumer wave: the 16 MFMAs of a 64-deep K-step as ONE instruction stream in which every MFMA is followed by
// one ds_read_b128 that is needed TWO k-sub-steps later.  With one wave per SIMD nothing else overlaps LDS latency with the
// matrix pipe: 16 reads followed by 16 MFMAs cost the SUM of both (551 + 512 cycles measured); one sub-step of lead costs the
// same (the wait in front of a sub-step then sits ~100 cycles behind the issue of its last read).  Operands:
//   %0..%3   accumulators (a,b) = 00 01 10 11
//   %4..%19  fragment sets F0..F3 = k-sub-steps 0..3, each {a0, a1, b0, b1}; F0 / F1 cross the tick boundary: they are read
//            from the NEXT stage under the MFMAs of sub-steps 2 / 3 (the ring is four deep: that stage landed a tick ago)
//   %20..%23 fragment addresses of sub-step 0 in the CURRENT stage: A0 A1 B0 B1 (sub-step kk: XOR 32 kk, into %24..%27);
//            moved on to the next ring slot (+ %28) in the MFMA shadows once the current stage's last reads are out
// Variants: FIRST tick of a tile (nothing prefetched: the reads of sub-steps 0 and 1 go out up front), MIDDLE, LAST (no next
// stage: it belongs to the other group's tile).
#define G5_RD(dst, addr) "ds_read_b128 " dst ", " addr "\n\t"
#define G5_XR(t, k, addr) "v_xor_b32 " t ", " k ", " addr "\n\t"
#define G5_AD(addr) "v_add_u32 " addr ", %28, " addr "\n\t"
#define G5_MF(c, b, a) "v_mfma_f32_32x32x16_bf16 " c ", " b ", " a ", " c "\n\t"
#define G5_W4 "s_waitcnt lgkmcnt(4)\n\t"
#define G5_W0 "s_waitcnt lgkmcnt(0)\n\t"
#define G5_FILL01 /* sub-steps 0 and 1 of the current stage, up front */                                                  \
  G5_RD("%4", "%20") G5_RD("%6", "%22") G5_RD("%7", "%23") G5_RD("%5", "%21")                                              \
  G5_XR("%24", "32", "%20") G5_RD("%8", "%24") G5_XR("%25", "32", "%22") G5_RD("%10", "%25")                               \
  G5_XR("%26", "32", "%23") G5_RD("%11", "%26") G5_XR("%27", "32", "%21") G5_RD("%9", "%27")
#define G5_SUB01 /* MFMAs of sub-steps 0 / 1, reads of sub-steps 2 / 3 */                                                  \
  G5_W4 G5_MF("%0", "%6", "%4") G5_XR("%24", "64", "%20") G5_RD("%12", "%24")                                              \
  G5_MF("%1", "%7", "%4") G5_XR("%25", "64", "%22") G5_RD("%14", "%25")                                                   \
  G5_MF("%2", "%6", "%5") G5_XR("%26", "64", "%23") G5_RD("%15", "%26")                                                   \
  G5_MF("%3", "%7", "%5") G5_XR("%27", "64", "%21") G5_RD("%13", "%27")                                                   \
  G5_W4 G5_MF("%0", "%10", "%8") G5_XR("%24", "96", "%20") G5_RD("%16", "%24")                                             \
  G5_MF("%1", "%11", "%8") G5_XR("%25", "96", "%22") G5_RD("%18", "%25")                                                  \
  G5_MF("%2", "%10", "%9") G5_XR("%26", "96", "%23") G5_RD("%19", "%26")                                                  \
  G5_MF("%3", "%11", "%9") G5_XR("%27", "96", "%21") G5_RD("%17", "%27")
#define G5_SUB23_NEXT /* MFMAs of sub-steps 2 / 3, reads of sub-steps 0 / 1 of the NEXT stage */                           \
  G5_W4 G5_MF("%0", "%14", "%12") G5_AD("%20") G5_RD("%4", "%20")                                                         \
  G5_MF("%1", "%15", "%12") G5_AD("%22") G5_RD("%6", "%22")                                                               \
  G5_MF("%2", "%14", "%13") G5_AD("%23") G5_RD("%7", "%23")                                                               \
  G5_MF("%3", "%15", "%13") G5_AD("%21") G5_RD("%5", "%21")                                                               \
  G5_W4 G5_MF("%0", "%18", "%16") G5_XR("%24", "32", "%20") G5_RD("%8", "%24")                                             \
  G5_MF("%1", "%19", "%16") G5_XR("%25", "32", "%22") G5_RD("%10", "%25")                                                 \
  G5_MF("%2", "%18", "%17") G5_XR("%26", "32", "%23") G5_RD("%11", "%26")                                                 \
  G5_MF("%3", "%19", "%17") G5_XR("%27", "32", "%21") G5_RD("%9", "%27")
#define G5_SUB23_LAST                                                                                                     \
  G5_W4 G5_MF("%0", "%14", "%12") G5_MF("%1", "%15", "%12") G5_MF("%2", "%14", "%13") G5_MF("%3", "%15", "%13")            \
  G5_W0 G5_MF("%0", "%18", "%16") G5_MF("%1", "%19", "%16") G5_MF("%2", "%18", "%17") G5_MF("%3", "%19", "%17")            \
  "s_nop 15\n\ts_nop 15\n\t"
struct G5Frags { u32x4 f[4][4]; };  // [k-sub-step][a0, a1, b0, b1]
#define G5_OPERANDS                                                                                                        \
  : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]), "+v"(fr.f[0][0]), "+v"(fr.f[0][1]), "+v"(fr.f[0][2]), \
    "+v"(fr.f[0][3]), "+v"(fr.f[1][0]), "+v"(fr.f[1][1]), "+v"(fr.f[1][2]), "+v"(fr.f[1][3]), "=&v"(fr.f[2][0]),             \
    "=&v"(fr.f[2][1]), "=&v"(fr.f[2][2]), "=&v"(fr.f[2][3]), "=&v"(fr.f[3][0]), "=&v"(fr.f[3][1]), "=&v"(fr.f[3][2]),         \
    "=&v"(fr.f[3][3]), "+v"(ad[0]), "+v"(ad[1]), "+v"(ad[2]), "+v"(ad[3]), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)        \
  : "s"(delta)                                                                                                             \
  : "memory"
template <int VARIANT>  // 0: first tick of a tile, 1: middle, 2: last
__device__ __forceinline__ void g5_tick(f32x16 (&acc)[2][2], G5Frags& fr, unsigned (&ad)[4], int delta) {
  unsigned t0, t1, t2, t3;
  if constexpr (VARIANT == 0) asm volatile(G5_FILL01 G5_SUB01 G5_SUB23_NEXT G5_OPERANDS);
  else if constexpr (VARIANT == 1) asm volatile(G5_SUB01 G5_SUB23_NEXT G5_OPERANDS);
  else asm volatile(G5_SUB01 G5_SUB23_LAST G5_OPERANDS);
}

// The same tick for the 128 x 64 tile (tile id 25: wave tile 64 x 32 = 2 x 1 fragments, 8 MFMAs and 12 reads per tick; a stage
// is 24 KiB).  Operands: %0 %1 accumulators (a0, a1) | %2..%13 F0..F3 = {a0, a1, b0} | %14 %15 %16 addresses A0 A1 B0 |
// %17..%19 temporaries | %20 ring-slot step.  Three reads per sub-step: lgkmcnt(3) in front of a sub-step leaves exactly the
// NEXT sub-step's reads outstanding.
#define G5N_AD(addr) "v_add_u32 " addr ", %20, " addr "\n\t"
#define G5N_W3 "s_waitcnt lgkmcnt(3)\n\t"
#define G5N_FILL01                                                                                                        \
  G5_RD("%2", "%14") G5_RD("%4", "%16") G5_RD("%3", "%15")                                                                 \
  G5_XR("%17", "32", "%14") G5_RD("%5", "%17") G5_XR("%19", "32", "%16") G5_RD("%7", "%19")                                \
  G5_XR("%18", "32", "%15") G5_RD("%6", "%18")
#define G5N_SUB01                                                                                                         \
  G5N_W3 G5_MF("%0", "%4", "%2") G5_XR("%17", "64", "%14") G5_RD("%8", "%17") G5_XR("%19", "64", "%16") G5_RD("%10", "%19") \
  G5_MF("%1", "%4", "%3") G5_XR("%18", "64", "%15") G5_RD("%9", "%18")                                                     \
  G5N_W3 G5_MF("%0", "%7", "%5") G5_XR("%17", "96", "%14") G5_RD("%11", "%17") G5_XR("%19", "96", "%16") G5_RD("%13", "%19") \
  G5_MF("%1", "%7", "%6") G5_XR("%18", "96", "%15") G5_RD("%12", "%18")
#define G5N_SUB23_NEXT                                                                                                    \
  G5N_W3 G5_MF("%0", "%10", "%8") G5N_AD("%14") G5_RD("%2", "%14") G5N_AD("%16") G5_RD("%4", "%16")                         \
  G5_MF("%1", "%10", "%9") G5N_AD("%15") G5_RD("%3", "%15")                                                               \
  G5N_W3 G5_MF("%0", "%13", "%11") G5_XR("%17", "32", "%14") G5_RD("%5", "%17") G5_XR("%19", "32", "%16") G5_RD("%7", "%19") \
  G5_MF("%1", "%13", "%12") G5_XR("%18", "32", "%15") G5_RD("%6", "%18")
#define G5N_SUB23_LAST                                                                                                    \
  G5N_W3 G5_MF("%0", "%10", "%8") G5_MF("%1", "%10", "%9")                                                                 \
  G5_W0 G5_MF("%0", "%13", "%11") G5_MF("%1", "%13", "%12") "s_nop 15\n\ts_nop 15\n\t"
struct G5FragsN { u32x4 f[4][3]; };  // [k-sub-step][a0, a1, b0]
#define G5N_OPERANDS                                                                                                       \
  : "+v"(acc[0][0]), "+v"(acc[1][0]), "+v"(fr.f[0][0]), "+v"(fr.f[0][1]), "+v"(fr.f[0][2]), "+v"(fr.f[1][0]), "+v"(fr.f[1][1]), \
    "+v"(fr.f[1][2]), "=&v"(fr.f[2][0]), "=&v"(fr.f[2][1]), "=&v"(fr.f[2][2]), "=&v"(fr.f[3][0]), "=&v"(fr.f[3][1]),          \
    "=&v"(fr.f[3][2]), "+v"(ad[0]), "+v"(ad[1]), "+v"(ad[2]), "=&v"(t0), "=&v"(t1), "=&v"(t2)                               \
  : "s"(delta)                                                                                                             \
  : "memory"
template <int VARIANT>
__device__ __forceinline__ void g5_tick(f32x16 (&acc)[2][1], G5FragsN& fr, unsigned (&ad)[3], int delta) {
  unsigned t0, t1, t2;
  if constexpr (VARIANT == 0) asm volatile(G5N_FILL01 G5N_SUB01 G5N_SUB23_NEXT G5N_OPERANDS);
  else if constexpr (VARIANT == 1) asm volatile(G5N_SUB01 G5N_SUB23_NEXT G5N_OPERANDS);
  else asm volatile(G5N_SUB01 G5N_SUB23_LAST G5N_OPERANDS);
}

template <int EPI, int BN>
__global__ __launch_bounds__(G5Lay<BN>::THREADS) void gemm5_kernel(const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ B,
                                                    int64_t ldb, void* __restrict__ Cout, int64_t ldc, int M, int N, int K,
                                                    MmtEpilogue epi, const int32_t* __restrict__ n_rows_dev) {
  constexpr int NP = G5Lay<BN>::NP, PIECES = 32 / NP;  // producer waves; 1-KiB LDS-DMA pieces per producer wave and stage
  constexpr int THREADS = G5Lay<BN>::THREADS;
  constexpr int NJ = BN / 64;               // 32-column fragments of a consumer wave (wave tile 64 x 32 NJ)
  static_assert(BN == 128 || BN == 64, "gemm5: tile width");
  constexpr int RING = G5Lay<BN>::RING, BOFF = G5Lay<BN>::BOFF;
  constexpr int LA = RING - 1;              // stages requested ahead of the tick that consumes them
  extern __shared__ __attribute__((aligned(256))) unsigned char smem_raw[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int G = (int)gridDim.x, bid = (int)blockIdx.x;
  const int nrows = n_rows_dev ? min(*n_rows_dev, M) : M;
  const int tiles_n = N / BN, tile_rows = (nrows + 127) >> 7, live = tile_rows * tiles_n;
  const G5Own own = g5_own(live, G, bid);
  const int n = own.n;  // tiles of this block
  const int KT = K >> 6;                                        // (>= 2: the host's check)
  if constexpr (EPI == MMT_EPI_DGELU) {
    if (epi.colsum) {  // row tiles past the live rows: their column sums are zero
      const int64_t dead = (int64_t)((M + 127) / 128 - tile_rows) * N;
      for (int64_t i = (int64_t)bid * THREADS + tid; i < dead; i += (int64_t)G * THREADS) epi.colsum[(int64_t)tile_rows * N + i] = 0.f;
    }
  }
  if (n == 0) return;
  const int S = n * KT;  // stages of this block, in tile order
#ifdef MMT_G5_INSTR
  long long g5p = clock64();
  const long long t_entry = g5p;
#endif
  long long t_a = 0, t_b = 0, t_c = 0, t_d = 0;
  (void)t_a; (void)t_b; (void)t_c; (void)t_d;

  if (wave >= 8) {  // ------------------------------------ producers ------------------------------------
    const int p = wave - 8;
    if (p * PIECES >= 16 + BN / 8) return;  // (128 x 64: a stage has 24 pieces, the last waves have none; an exited wave is not waited for)
    const int g0 = p * PIECES;         // this wave's first 8-row group: 0..15 = A rows, 16..31 = B rows
    const bool is_b = g0 >= 16;
    const int rg0 = g0 & 15;
    const bf16_t* base_g = is_b ? B : A;
    const int64_t ld = is_b ? ldb : lda;
    const int rmax = (is_b ? N : M) - 1;
    unsigned voff[PIECES];
    int ti = 0, kt = 0;                // tile / K-step of the next stage to request
    const bf16_t* tile_base = base_g;
    auto setup = [&](int i) {
      int m0, n0;
      g5_tile<BN>(own.id(i), tiles_n, tile_rows, m0, n0);
      const int r0 = is_b ? n0 : m0;
      tile_base = base_g + (int64_t)r0 * ld;
#pragma unroll
      for (int q = 0; q < PIECES; ++q) {
        const int r = (rg0 + q) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);  // the XOR swizzle of the image goes onto the SOURCE address
        voff[q] = (unsigned)((int64_t)(min(r0 + r, rmax) - r0) * ld * 2 + c * 16);
      }
    };
    int slot = 0;  // ring slot of the next stage to request
    auto issue = [&]() {
      const __amdgpu_buffer_rsrc_t desc = __builtin_amdgcn_make_buffer_rsrc((void*)tile_base, 0, 0x7fffffff, 0x00020000);
      unsigned char* dst = smem_raw + slot * G5_HALF + (is_b ? BOFF : 0) + rg0 * 1024;
#ifndef G5_LAB_NO_DMA
#pragma unroll
      for (int q = 0; q < PIECES; ++q)
        if (G5_LAB_AUX_A == G5_LAB_AUX_B || !is_b)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(desc, LDS_PTR(dst + q * 1024), 16, (int)voff[q], kt * 128, 0, G5_LAB_AUX_A);
        else
          __builtin_amdgcn_raw_ptr_buffer_load_lds(desc, LDS_PTR(dst + q * 1024), 16, (int)voff[q], kt * 128, 0, G5_LAB_AUX_B);
#else
      (void)desc; (void)dst;
#endif
      slot = slot == RING - 1 ? 0 : slot + 1;
      if (++kt == KT) {
        kt = 0;
        if (++ti < n) setup(ti);
      }
    };
    // A consumer's tick tau reads stage tau and the first half of stage tau + 1: at the barrier that OPENS tick tau the
    // stages <= tau + 1 have landed and the stages <= tau - 1 are free, so the stages tau + 2 .. tau + LA are in flight or
    // being requested.  (S >= 2.)
    auto wait_all_but = [](int stages) {  // this wave's requests: PIECES per stage, answered in order
      if (stages >= 2 && LA >= 4) g5_vmwait<2 * PIECES>();
      else if (stages >= 1) g5_vmwait<PIECES>();
      else g5_vmwait<0>();
    };
    setup(0);
    const int ahead = min(S, LA);
    for (int q = 0; q < ahead; ++q) issue();
    wait_all_but(ahead - 2);
    G5_TICK(t_a);
    g5_barrier();  // stages 0 and 1 have landed
    for (int tau = 0; tau < S; ++tau) {
      if (tau + LA < S) issue();  // the slot of stage tau - 1 takes stage tau + LA
      G5_TICK(t_b);
      wait_all_but(min(LA - 2, S - 3 - tau));  // stage tau + 2 has landed (if there is one)
      G5_TICK(t_c);
      g5_barrier();
      G5_TICK(t_d);
    }
#ifdef MMT_G5_INSTR
    if (epi.row_index == nullptr && epi.seed_dev != nullptr && tid == 512) {
      long long* d = (long long*)epi.seed_dev + ((int64_t)bid * 3 + 2) * 8;
      d[0] = t_a; d[1] = t_b; d[2] = t_c; d[3] = t_d; d[4] = clock64() - t_entry; d[5] = S; d[6] = n;
    }
#endif
    return;
  }

  // ------------------------------------ consumers ------------------------------------
  const int grp = wave >> 2, w4 = wave & 3, wm = w4 >> 1, wn = w4 & 1;
  const int l31 = lane & 31, lh = lane >> 5;
  if constexpr (EPI == MMT_EPI_BIAS_GELU || EPI == MMT_EPI_DGELU) {
    const float* src = EPI == MMT_EPI_BIAS_GELU ? g_gelu_lut_cdf : g_gelu_lut_dgelu;
    float* lut = (float*)(smem_raw + G5_LUT_OFF);
    for (int i = tid; i < MMT_GELU_LUT_N; i += 512) lut[i] = src[i];
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  g5_barrier();  // stages 0 and 1 have landed; the GELU table is in LDS

  f32x16 acc[2][NJ];
  int slot = 0;  // ring slot of the stage the current tick consumes
  for (int i = 0; i <= n; ++i) {
    if ((i & 1) == grp) {
      if (i == n) break;  // nothing left for this group
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < NJ; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
      // per-lane LDS addresses of this wave's sub-step-0 fragments in the current ring slot: row r, 16-byte chunk lh ^
      // ((r >> 1) & 7).  Rebuilt per tile (not live across this group's epilogue); the ticks move them from slot to slot.
      unsigned ad[2 + NJ];
      {
        unsigned lds0 = (unsigned)(uintptr_t)LDS_PTR(smem_raw) + (unsigned)slot * G5_HALF;
        asm volatile("" : "+v"(lds0));
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int ra = wm * 64 + h * 32 + l31;
          ad[h] = lds0 + (unsigned)(ra * 128) + (unsigned)((lh ^ ((ra >> 1) & 7)) << 4);
        }
#pragma unroll
        for (int h = 0; h < NJ; ++h) {
          const int rb = wn * 32 * NJ + h * 32 + l31;
          ad[2 + h] = lds0 + (unsigned)BOFF + (unsigned)(rb * 128) + (unsigned)((lh ^ ((rb >> 1) & 7)) << 4);
        }
      }
      typename std::conditional<NJ == 2, G5Frags, G5FragsN>::type fr;
#pragma unroll
      for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int f = 0; f < 2 + NJ; ++f) asm volatile("" : "=v"(fr.f[k][f]));  // (defined, not initialised: the first tick fills them)
      auto delta_of = [](int s) { return __builtin_amdgcn_readfirstlane(s == RING - 1 ? -(RING - 1) * G5_HALF : G5_HALF); };
      auto next = [](int s) { return s == RING - 1 ? 0 : s + 1; };
      g5_tick<0>(acc, fr, ad, delta_of(slot));
      G5_TICK(t_a);
      g5_barrier();  // this stage may be overwritten; the next one has landed
      G5_TICK(t_b);
      slot = next(slot);
      for (int kt = 1; kt < KT - 1; ++kt) {
        g5_tick<1>(acc, fr, ad, delta_of(slot));
        G5_TICK(t_a);
        g5_barrier();
        G5_TICK(t_b);
        slot = next(slot);
      }
      g5_tick<2>(acc, fr, ad, 0);
      G5_TICK(t_a);
      g5_barrier();
      G5_TICK(t_b);
      slot = next(slot);
    } else {
      if (i >= 1) {
        int m0, n0;
        g5_tile<BN>(own.id(i - 1), tiles_n, tile_rows, m0, n0);
        const int nbar = i == n ? 0 : KT;
        g5_epilogue<EPI, NJ>(acc, smem_raw, m0 + wm * 64, n0 + wn * 32 * NJ, grp, w4, lane, M, N, nrows, Cout, (int)ldc, epi, nbar, t_c, t_d);
        if constexpr (EPI == MMT_EPI_DGELU && NJ == 2) {
          if (epi.colsum && wm == 0 && lane < 16) {  // both wave rows' sums have crossed a barrier: one row of sums per 128 output rows
            const float* red = (const float*)(smem_raw + G5_RED_OFF) + grp * 256 + wn * 64 + lane * 4;
            const f32x4 s = *(const f32x4*)red + *(const f32x4*)(red + 128);
            *(f32x4*)(epi.colsum + (int64_t)(m0 >> 7) * N + n0 + wn * 64 + lane * 4) = s;
          }
        }
      } else {
        for (int kt = 0; kt < KT; ++kt) g5_barrier();
      }
      slot = (slot + KT) % RING;
#ifdef MMT_G5_INSTR
      g5p = clock64();
#endif
    }
  }
#ifdef MMT_G5_INSTR
  if (epi.row_index == nullptr && epi.seed_dev != nullptr && (tid == 0 || tid == 256)) {
    long long* d = (long long*)epi.seed_dev + ((int64_t)bid * 3 + grp) * 8;
    d[0] = t_a; d[1] = t_b; d[2] = t_c; d[3] = t_d; d[4] = clock64() - t_entry; d[5] = S; d[6] = n;
  }
#endif
}

template <int EPI, int BN>
static int launch5(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int M, int N, int K,
                   const MmtEpilogue& e, const int32_t* nr, hipStream_t s) {
  static int cus = 0;
  if (!cus) {
    if (hipFuncSetAttribute((const void*)gemm5_kernel<EPI, BN>, hipFuncAttributeMaxDynamicSharedMemorySize, G5Lay<BN>::LDS) != hipSuccess) return MMT_ERR_ARG;
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return MMT_ERR_ARG;
    cus = n > 8 ? (n & ~7) : 8;
  }
  // one block per CU (157 KiB of LDS each); a problem with fewer tiles than CUs starts a block per tile (rounded up to 8)
  const int64_t tiles = (int64_t)((M + 127) / 128) * (N / BN);
  const int grid = tiles >= cus ? cus : (int)((tiles + 7) & ~(int64_t)7);
  hipLaunchKernelGGL((gemm5_kernel<EPI, BN>), dim3(grid), dim3(G5Lay<BN>::THREADS), G5Lay<BN>::LDS, s, (const bf16_t*)A, lda, (const bf16_t*)B, ldb, C, ldc, M, N, K, e, nr);
  return (int)hipGetLastError();
}

// tiles 24 (bn = 128) / 25 (bn = 64) of mmt_gemm2_dispatch: N % bn == 0, K % 64 == 0, K >= 128; A / B rows are clamped to
// M - 1 / N - 1; the epilogue addresses its matrices with 32-bit byte offsets (MUBUF range check instead of row predicates):
// each must stay below 2 GiB.  The 64-wide tile is for the long-K GEMMs with narrow outputs (FFN down-projection, its input
// gradient): no GELU epilogues, no row dots.
int mmt_gemm5_dispatch(int epilogue, int bn, const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int M,
                       int N, int K, const MmtEpilogue& e, const int32_t* nr, hipStream_t s) {
  if ((bn != 128 && bn != 64) || N % bn || K % 64 || K < 128) return MMT_ERR_ARG;
  const int64_t lim = (int64_t)1 << 31, rows = (int64_t)M + 128;
  if (rows * ldc * 4 >= lim || rows * e.ldres * 4 >= lim || rows * e.ldout2 * 2 >= lim || rows * e.ldaux * 2 >= lim) return MMT_ERR_ARG;
  if (bn == 64) {
    if (e.dot_out) return MMT_ERR_ARG;
    switch (epilogue) {
      case MMT_EPI_BF16: return launch5<MMT_EPI_BF16, 64>(A, lda, B, ldb, C, ldc, M, N, K, e, nr, s);
      case MMT_EPI_BIAS_BF16: return launch5<MMT_EPI_BIAS_BF16, 64>(A, lda, B, ldb, C, ldc, M, N, K, e, nr, s);
      case MMT_EPI_BIAS_DROP_RES: return launch5<MMT_EPI_BIAS_DROP_RES, 64>(A, lda, B, ldb, C, ldc, M, N, K, e, nr, s);
      case MMT_EPI_ADD_F32: return launch5<MMT_EPI_ADD_F32, 64>(A, lda, B, ldb, C, ldc, M, N, K, e, nr, s);
      case MMT_EPI_F32: return launch5<MMT_EPI_F32, 64>(A, lda, B, ldb, C, ldc, M, N, K, e, nr, s);
      case MMT_EPI_BIAS_F32: return launch5<MMT_EPI_BIAS_F32, 64>(A, lda, B, ldb, C, ldc, M, N, K, e, nr, s);
    }
    return MMT_ERR_ARG;
  }
  switch (epilogue) {
    case MMT_EPI_BF16: return launch5<MMT_EPI_BF16, 128>(A, lda, B, ldb, C, ldc, M, N, K, e, nr, s);
    case MMT_EPI_BIAS_BF16: return launch5<MMT_EPI_BIAS_BF16, 128>(A, lda, B, ldb, C, ldc, M, N, K, e, nr, s);
    case MMT_EPI_BIAS_GELU: return launch5<MMT_EPI_BIAS_GELU, 128>(A, lda, B, ldb, C, ldc, M, N, K, e, nr, s);
    case MMT_EPI_BIAS_DROP_RES: return launch5<MMT_EPI_BIAS_DROP_RES, 128>(A, lda, B, ldb, C, ldc, M, N, K, e, nr, s);
    case MMT_EPI_DGELU: return launch5<MMT_EPI_DGELU, 128>(A, lda, B, ldb, C, ldc, M, N, K, e, nr, s);
    case MMT_EPI_ADD_F32: return launch5<MMT_EPI_ADD_F32, 128>(A, lda, B, ldb, C, ldc, M, N, K, e, nr, s);
    case MMT_EPI_F32: return launch5<MMT_EPI_F32, 128>(A, lda, B, ldb, C, ldc, M, N, K, e, nr, s);
    case MMT_EPI_BIAS_F32: return launch5<MMT_EPI_BIAS_F32, 128>(A, lda, B, ldb, C, ldc, M, N, K, e, nr, s);
  }
  return MMT_ERR_ARG;
}
