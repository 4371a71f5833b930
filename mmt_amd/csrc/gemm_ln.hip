// Full-row NT GEMM with the LayerNorm in its epilogue (gfx950), for the hidden -> hidden projection of BertSelfOutput:
//   z = dropout(A . W^T + bias) + residual            model/bert.py:185-188  (dense, dropout, + input_tensor)
//   h = LayerNorm(z)                                   model/bert.py:188      (fp32 statistics, eps 1e-12)
// One launch instead of the N = hidden GEMM (MMT_EPI_BIAS_DROP_RES) + mmt_ln_fwd: LayerNorm needs whole rows and a GEMM tile
// of gemm2.hip holds 64 of the 512 columns, so every N = hidden GEMM was followed by a LayerNorm launch that re-reads z
// (7 us under graph replay at 3.6 k rows, next to an 11 us GEMM).  Here a block owns 32 rows x ALL 512 columns:
//   * 8 waves, wave w = columns 64 w .. 64 w + 63 (1 x 2 fragments of mfma_f32_32x32x16_bf16); the B rows a wave stages
//     through LDS-DMA are the 64 weight rows it alone consumes, the 32 x 64 A tile is shared;
//   * every block streams the whole weight (d x K bf16 = 512 KiB at K = 512) out of its XCD's L2: 68 KiB per 64-deep
//     K-step, two stages, one barrier per step -- bound by what a CU ingests (~40 B/clk: ~1.7 k cycles per step), not by
//     its 256 MFMA cycles.  That only pays for K = hidden (8 steps); the K = intermediate projection (3 MiB per block) keeps
//     its own GEMM + LayerNorm launches;
//   * epilogue: accumulators -> row-major fp32 image in LDS (the ring is free by then), then ONE WAVE PER ROW exactly as
//     mmt_ln_fwd does it (lane = columns 4 lane .. + 3 and 256 + 4 lane .. + 3, same summation order): bias, dropout on the
//     ORIGINAL row number, residual, z stored, mean / rstd by wave reductions, h32 / h16 stored -- the results equal the two
//     launches' (the GEMM sums K-steps in ascending order like the un-phased tiles of gemm2.hip).
#include "mmt_common.h"
#include "../../include/mmt_hip.h"

typedef __attribute__((ext_vector_type(16))) float f32x16;

#define GL_D 512                      // output width = hidden size
#define GL_BM 32
#define GL_STAGE ((GL_BM + GL_D) * 128)  // bytes per 64-deep K-step: A 32 rows + B 512 rows of 128 B
#define GL_P (GL_D + 4)               // fp32 pitch of the epilogue image
#define GL_LDS (2 * GL_STAGE)         // (>= the 32 x 516 x 4 B image)

template <int N> __device__ __forceinline__ void gl_vmwait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

struct GemmLnArgs {
  const bf16_t* A; int64_t lda; const bf16_t* W; int64_t ldw;
  const float *bias, *res; int64_t ldres;
  const int32_t* row_index; uint32_t drop_key, thr16; float drop_scale; const uint32_t* seed_dev;
  float* z_out; const float *gamma, *beta; float eps;
  float* h32; bf16_t* h16; float *mean, *rstd;
  int M, K; const int32_t* n_rows_dev;
};

__global__ __launch_bounds__(512) void gemm_ln_kernel(GemmLnArgs a) {
  extern __shared__ __attribute__((aligned(256))) unsigned char smem_raw[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nrows = a.n_rows_dev ? min(*a.n_rows_dev, a.M) : a.M;
  const int m0 = (int)blockIdx.x * GL_BM;
  if (m0 >= nrows) return;  // (token packing: row blocks past the live rows)
  const int l31 = lane & 31, lh = lane >> 5;
  const int KT = a.K >> 6;

  // ---- LDS-DMA sources: a stage is [A 32 x 128 B | B 512 x 128 B] as 68 pieces of 8 rows; wave w moves B pieces 8 w .. 8 w + 7
  // (= its own 64 weight rows) and, for w < 4, A piece w.  Lane-linear image, chunk c of row r at chunk c ^ ((r >> 1) & 7).
  const int sub = lane >> 3, ch = lane & 7;
  unsigned ob[8], oa = 0;
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int r = wave * 64 + q * 8 + sub;
    ob[q] = (unsigned)((int64_t)r * a.ldw * 2 + (ch ^ ((r >> 1) & 7)) * 16);
  }
  if (wave < 4) {
    const int r = wave * 8 + sub;
    oa = (unsigned)((int64_t)(min(m0 + r, a.M - 1) - m0) * a.lda * 2 + (ch ^ ((r >> 1) & 7)) * 16);
  }
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)(a.A + (int64_t)m0 * a.lda), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)a.W, 0, 0x7fffffff, 0x00020000);
  auto issue = [&](int kt) {
    unsigned char* base = smem_raw + (kt & 1) * GL_STAGE;
    if (wave < 4) __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, LDS_PTR(base + wave * 1024), 16, (int)oa, kt * 128, 0, 0);
#pragma unroll
    for (int q = 0; q < 8; ++q)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, LDS_PTR(base + GL_BM * 128 + (wave * 8 + q) * 1024), 16, (int)ob[q], kt * 128, 0, 0);
  };

  // ---- fragment addresses (k-sub-step 0; sub-step kk: XOR 32 kk) ----
  const unsigned lds0 = (unsigned)(uintptr_t)LDS_PTR(smem_raw);
  const unsigned fa = lds0 + (unsigned)(l31 * 128) + (unsigned)((lh ^ ((l31 >> 1) & 7)) << 4);
  unsigned fb[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int r = wave * 64 + j * 32 + l31;
    fb[j] = lds0 + (unsigned)(GL_BM * 128 + r * 128) + (unsigned)((lh ^ ((r >> 1) & 7)) << 4);
  }

  f32x16 acc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  issue(0);
  for (int kt = 0; kt < KT; ++kt) {
    gl_vmwait<0>();  // stage kt (the only one outstanding) has landed
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();  // ... for every wave; and everybody is done reading the other buffer
    asm volatile("" ::: "memory");
    if (kt + 1 < KT) issue(kt + 1);
    const unsigned so = (unsigned)(kt & 1) * GL_STAGE;
    u32x4 pa[4], pb[4][2];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      asm volatile("ds_read_b128 %0, %1" : "=v"(pa[kk]) : "v"((fa + so) ^ (unsigned)(kk << 5)));
#pragma unroll
      for (int j = 0; j < 2; ++j) asm volatile("ds_read_b128 %0, %1" : "=v"(pb[kk][j]) : "v"((fb[j] + so) ^ (unsigned)(kk << 5)));
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      if (kk == 0) asm volatile("s_waitcnt lgkmcnt(9)" ::: "memory");
      if (kk == 1) asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
      if (kk == 2) asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory");
      if (kk == 3) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      asm volatile("" : "+v"(pa[kk]));
#pragma unroll
      for (int j = 0; j < 2; ++j) asm volatile("" : "+v"(pb[kk][j]));
#pragma unroll
      for (int j = 0; j < 2; ++j)
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, pb[kk][j]), __builtin_bit_cast(bf16x8_t, pa[kk]),
                                                         acc[j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  __syncthreads();  // every wave is done with the ring

  // ---- epilogue: accumulators -> fp32 image [32][516]; a lane holds row l31, columns 64 w + 32 j + 8 q + 4 lh .. + 3 ----
  float* st = (float*)smem_raw;
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 v = {acc[j][4 * q], acc[j][4 * q + 1], acc[j][4 * q + 2], acc[j][4 * q + 3]};
      *(f32x4*)(st + l31 * GL_P + wave * 64 + j * 32 + 8 * q + 4 * lh) = v;
    }
  __syncthreads();
  const unsigned dkey = eff_key(a.drop_key, a.seed_dev);
  f32x4 b4[2], g4[2], be4[2];
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const int col = c * 256 + lane * 4;
    b4[c] = *(const f32x4*)(a.bias + col);
    g4[c] = *(const f32x4*)(a.gamma + col);
    be4[c] = *(const f32x4*)(a.beta + col);
  }
  // the four rows of this wave: second operands first (residual rows, original row numbers), then row by row as mmt_ln_fwd
  f32x4 res[4][2];
  int orow[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = min(m0 + wave * 4 + i, nrows - 1);
    orow[i] = (a.thr16 && a.row_index) ? a.row_index[row] : row;
#pragma unroll
    for (int c = 0; c < 2; ++c) res[i][c] = *(const f32x4*)(a.res + (int64_t)row * a.ldres + c * 256 + lane * 4);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = wave * 4 + i, row = m0 + r;
    if (row >= nrows) break;  // (wave-uniform)
    f32x4 x[2];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int col = c * 256 + lane * 4;
      f32x4 v = *(const f32x4*)(st + r * GL_P + col);
      v += b4[c];
      if (a.thr16) {
        bool k[4];
        keep4(dkey, (unsigned long long)orow[i] * (unsigned)GL_D + (unsigned)col, a.thr16, k);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = k[e] ? v[e] * a.drop_scale : 0.f;
      }
      v += res[i][c];
      *(f32x4*)(a.z_out + (int64_t)row * GL_D + col) = v;
      x[c] = v;
      s += v[0] + v[1] + v[2] + v[3];
    }
    const float mean = wave_sum(s) / (float)GL_D;
    float var = 0.f;
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float t = x[c][e] - mean; var += t * t; }
    const float rstd = 1.0f / sqrtf(wave_sum(var) / (float)GL_D + a.eps);
    if (lane == 0) { a.mean[row] = mean; a.rstd[row] = rstd; }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int col = c * 256 + lane * 4;
      f32x4 y;
#pragma unroll
      for (int e = 0; e < 4; ++e) y[e] = (x[c][e] - mean) * rstd * g4[c][e] + be4[c][e];
      if (a.h32) *(f32x4*)(a.h32 + (int64_t)row * GL_D + col) = y;
      u32x2 o = {pack_bf2(y[0], y[1]), pack_bf2(y[2], y[3])};
      *(u32x2*)(a.h16 + (int64_t)row * GL_D + col) = o;
    }
  }
}

// z = dropout(A[M,K] . W[512,K]^T + bias) + res ; h = LN(z): model/bert.py:185-188.  A / W bf16 row-major (K % 64 == 0),
// res / z_out / h32 fp32 [M, 512] (z_out, h32 contiguous; h32 nullable), h16 bf16 [M, 512], mean / rstd [M].  Dropout as
// MMT_EPI_BIAS_DROP_RES: element (row_index[row] or row, column) of the stream hash(drop_key, *seed_dev); thr16 = 0: none.
// n_rows_dev (nullable): live rows of a packed batch (row blocks past them exit).
extern "C" int mmt_gemm_nt_ln_fwd(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, const float* res,
                                  int64_t ldres, const int32_t* row_index, uint32_t drop_key, uint32_t drop_thr16, float drop_scale,
                                  const uint32_t* seed_dev, float* z_out, const float* gamma, const float* beta, float eps,
                                  float* h32, void* h16, float* mean, float* rstd, int M, int N, int K,
                                  const int32_t* n_rows_dev, void* stream) {
  if (!A || !W || !bias || !res || !z_out || !gamma || !beta || !h16 || !mean || !rstd || M <= 0 || K <= 0) return MMT_ERR_ARG;
  if (N != GL_D || K % 64) return MMT_ERR_ARG;
  if ((lda % 8) || (ldw % 8) || (ldres % 4) || ((uintptr_t)A & 15) || ((uintptr_t)W & 15) || ((uintptr_t)res & 15) ||
      ((uintptr_t)z_out & 15) || ((uintptr_t)h16 & 7) || (h32 && ((uintptr_t)h32 & 15)))
    return MMT_ERR_ALIGN;
  if ((int64_t)GL_D * ldw * 2 >= ((int64_t)1 << 31) || (int64_t)GL_BM * lda * 2 >= ((int64_t)1 << 31)) return MMT_ERR_ARG;
  static bool configured = false;
  if (!configured) {
    if (hipFuncSetAttribute((const void*)gemm_ln_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, GL_LDS) != hipSuccess) return MMT_ERR_ARG;
    configured = true;
  }
  GemmLnArgs a;
  a.A = (const bf16_t*)A; a.lda = lda; a.W = (const bf16_t*)W; a.ldw = ldw; a.bias = bias; a.res = res; a.ldres = ldres;
  a.row_index = row_index; a.drop_key = drop_key; a.thr16 = drop_thr16; a.drop_scale = drop_scale; a.seed_dev = seed_dev;
  a.z_out = z_out; a.gamma = gamma; a.beta = beta; a.eps = eps; a.h32 = h32; a.h16 = (bf16_t*)h16; a.mean = mean; a.rstd = rstd;
  a.M = M; a.K = K; a.n_rows_dev = n_rows_dev;
  hipLaunchKernelGGL(gemm_ln_kernel, dim3((M + GL_BM - 1) / GL_BM), dim3(512), GL_LDS, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}
