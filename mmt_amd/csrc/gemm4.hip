// Producer / consumer NT GEMM for the packed N = hidden GEMMs (gfx950, r04 lab; tile id 23 of the dispatcher):
//   C[M,N] = A[M,K] . B[N,K]^T (+ the fused epilogues of gemm2), 128 x 64 tiles.
//
// tools/ubench/ingest_rate.hip: a CU pulls 45-57 B/clk out of L2 when >= 64 KiB are in flight and requests keep coming; the
// K-loops of gemm2.hip get 21-22, because the waves that run the MFMAs also issue the LDS-DMA pieces (70-140 cycles of stall
// per 1-KiB piece when the waves of a group issue together) and then wait for them behind the other group's burst
// (DESIGN section 7).  Here the roles are split: waves 4..7 do nothing but issue (6 pieces each per 64-deep K-step into a
// five-deep ring of 24 KiB stages, three K-steps in flight, counted vmcnt), waves 0-3 (64 x 32 wave tiles, one per SIMD) do
// nothing but read fragments and run MFMAs; one workgroup barrier per K-step hands a landed stage to the consumers and a
// consumed one back to the producers.  The producers leave before the epilogue (finished waves do not count in barriers).
#ifdef MMT_LAB_TILES  // (lab library only: python -m mmt_amd.build --lab)
#include "mmt_common.h"
#include "../../include/mmt_hip.h"
#include "gemm_epi.h"
#include <type_traits>

#define G4_BK 64
#define G4_RING 5
#define G4_NP 4  // producer waves (a wave gets one 1-KiB piece accepted per ~50 cycles: tools/ubench/ingest_rate.hip)
template <int ROWS>
__device__ __forceinline__ void g4_stage(const bf16_t* __restrict__ G, int64_t ld, int row0, int row_max, int k0, bf16_t* lds_tile,
                                         int pw, int lane) {  // producer wave pw of G4_NP
  constexpr int PER = ROWS / 8 / G4_NP;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int rbase = (pw * PER + i) * 8;  // 8 rows of 128 B per wave-instruction
    const int r = rbase + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    const int gr = min(row0 + r, row_max);
    const bf16_t* src = G + (int64_t)gr * ld + k0 + c * 8;
    __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(lds_tile + rbase * G4_BK), 16, 0, 0);
  }
}
template <int N> __device__ __forceinline__ void g4_vmwait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void g4_barrier() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

template <int EPI>
__global__ __launch_bounds__(256 + 64 * G4_NP) void gemm4_kernel(const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ B, int64_t ldb,
                                                                 void* __restrict__ Cout, int64_t ldc, int M, int N, int K, MmtEpilogue epi,
                                                                 const int32_t* __restrict__ n_rows_dev) {
  constexpr int BM = 128, BN = 64, MI = 2, NJ = 1, STAGE = (BM + BN) * G4_BK, PIECES = (BM + BN) / 8 / G4_NP;  // per producer wave and K-step
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16_t* smem = (bf16_t*)smem_raw;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_n = N / BN;
  const int nrows = n_rows_dev ? *n_rows_dev : M;
  const int bid = (int)blockIdx.x;
  const int live_tiles = min((int)gridDim.x, ((min(nrows, M) + BM - 1) / BM) * tiles_n);
  if (bid >= live_tiles) {  // dead tile (token packing): nothing to compute
    if constexpr (EPI == MMT_EPI_DGELU) {
      if (epi.colsum) {
        const int dm0 = (bid / tiles_n) * BM, dn0 = (bid % tiles_n) * BN;
        if (dm0 < M && tid < BN) epi.colsum[(int64_t)(dm0 / 128) * N + dn0 + tid] = 0.f;
      }
    }
    return;
  }
  const int id = xcd_remap(bid, live_tiles);
  const int tm = id / tiles_n, tn = id % tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int KT = K / G4_BK;

  if (wave >= 4) {  // ---------------- producers ----------------
    const int pw = wave - 4;
    auto issue = [&](int kt) {
      bf16_t* base = smem + (kt % G4_RING) * STAGE;
      g4_stage<BM>(A, lda, m0, M - 1, kt * G4_BK, base, pw, lane);
      g4_stage<BN>(B, ldb, n0, N - 1, kt * G4_BK, base + BM * G4_BK, pw, lane);
    };
#pragma unroll
    for (int s0 = 0; s0 < G4_RING - 1; ++s0)
      if (s0 < KT) issue(s0);
    for (int kt = 0; kt < KT; ++kt) {
      // K-step kt has landed when at most the K-steps requested after it are outstanding
      const int ahead = min(KT - 1 - kt, G4_RING - 2);
      if (ahead >= 4) g4_vmwait<4 * PIECES>();
      else if (ahead == 3) g4_vmwait<3 * PIECES>();
      else if (ahead == 2) g4_vmwait<2 * PIECES>();
      else if (ahead == 1) g4_vmwait<PIECES>();
      else g4_vmwait<0>();
      g4_barrier();  // consumers may read stage kt; they are done with stage kt - 1
      if (kt + G4_RING - 1 < KT) issue(kt + G4_RING - 1);
    }
    return;
  }

  // ---------------- consumers: waves 0..3, wave tile 64 x 32 ----------------
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, lh = lane >> 5;
  f32x16 acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][0][r] = 0.f;
  unsigned aoffb[MI], boffb[NJ];  // byte offsets of this lane's fragments inside a stage (k-sub-step 0)
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int r = wm * 64 + i * 32 + l31;
    aoffb[i] = (unsigned)(r * G4_BK * 2) + (unsigned)((lh ^ ((r >> 1) & 7)) << 4);
  }
  {
    const int r = wn * 32 + l31;
    boffb[0] = (unsigned)((BM * G4_BK + r * G4_BK) * 2) + (unsigned)((lh ^ ((r >> 1) & 7)) << 4);
  }
  const unsigned lds0 = (unsigned)(uintptr_t)LDS_PTR(smem);
  for (int kt = 0; kt < KT; ++kt) {
    g4_barrier();
    const unsigned sbase = lds0 + (unsigned)(kt % G4_RING) * (STAGE * 2);
    u32x4 pa[4][MI], pb[4][NJ];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
      for (int i = 0; i < MI; ++i) asm volatile("ds_read_b128 %0, %1" : "=v"(pa[kk][i]) : "v"((sbase + aoffb[i]) ^ (unsigned)(kk << 5)));
      asm volatile("ds_read_b128 %0, %1" : "=v"(pb[kk][0]) : "v"((sbase + boffb[0]) ^ (unsigned)(kk << 5)));
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      constexpr int PER = MI + NJ;
      if (kk == 0) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(3 * PER) : "memory");
      if (kk == 1) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * PER) : "memory");
      if (kk == 2) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(1 * PER) : "memory");
      if (kk == 3) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < MI; ++i) asm volatile("" : "+v"(pa[kk][i]));
      asm volatile("" : "+v"(pb[kk][0]));
#pragma unroll
      for (int i = 0; i < MI; ++i)
        acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, pb[kk][0]), __builtin_bit_cast(bf16x8_t, pa[kk][i]),
                                                            acc[i][0], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // the producers have left; the four consumer waves run the 4-wave epilogue of gemm2's 128 x 64 tile
  gemm_tile_epilogue<BM, BN, 2, 2, 256, EPI, false>(acc, smem_raw, m0, n0, M, N, nrows, Cout, ldc, epi, wm, wn, 0, tid, nullptr);
}

template <int EPI>
static int launch4(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int M, int N, int K,
                   const MmtEpilogue& e, const int32_t* nr, hipStream_t s) {
  constexpr int lds = G4_RING * (128 + 64) * G4_BK * 2;  // 24 KiB per stage
  static bool configured = false;
  if (!configured) {
    if (hipFuncSetAttribute((const void*)gemm4_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return MMT_ERR_ARG;
    configured = true;
  }
  const int grid = ((M + 127) / 128) * (N / 64);
  hipLaunchKernelGGL((gemm4_kernel<EPI>), dim3(grid), dim3(256 + 64 * G4_NP), lds, s, (const bf16_t*)A, lda, (const bf16_t*)B, ldb, C, ldc, M, N, K, e, nr);
  return (int)hipGetLastError();
}

// tile 23 of mmt_gemm2_dispatch: N % 64 == 0, K % 64 == 0
int mmt_gemm4_dispatch(int epilogue, const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int M, int N,
                       int K, const MmtEpilogue& e, const int32_t* nr, hipStream_t s) {
  if (N % 64 || K % 64) return MMT_ERR_ARG;
  switch (epilogue) {
    case MMT_EPI_BF16: return launch4<MMT_EPI_BF16>(A, lda, B, ldb, C, ldc, M, N, K, e, nr, s);
    case MMT_EPI_BIAS_BF16: return launch4<MMT_EPI_BIAS_BF16>(A, lda, B, ldb, C, ldc, M, N, K, e, nr, s);
    case MMT_EPI_BIAS_GELU: return launch4<MMT_EPI_BIAS_GELU>(A, lda, B, ldb, C, ldc, M, N, K, e, nr, s);
    case MMT_EPI_BIAS_DROP_RES: return launch4<MMT_EPI_BIAS_DROP_RES>(A, lda, B, ldb, C, ldc, M, N, K, e, nr, s);
    case MMT_EPI_DGELU: return launch4<MMT_EPI_DGELU>(A, lda, B, ldb, C, ldc, M, N, K, e, nr, s);
    case MMT_EPI_ADD_F32: return launch4<MMT_EPI_ADD_F32>(A, lda, B, ldb, C, ldc, M, N, K, e, nr, s);
    case MMT_EPI_F32: return launch4<MMT_EPI_F32>(A, lda, B, ldb, C, ldc, M, N, K, e, nr, s);
    case MMT_EPI_BIAS_F32: return launch4<MMT_EPI_BIAS_F32>(A, lda, B, ldb, C, ldc, M, N, K, e, nr, s);
  }
  return MMT_ERR_ARG;
}
#endif  // MMT_LAB_TILES
