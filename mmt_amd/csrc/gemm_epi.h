// LDS-staged epilogue shared by the NT GEMM kernels (gemm2.hip: 128/256-row tiles; gemm3.hip: the 256x256 eight-phase
// kernel): the accumulators (mfma_f32_32x32x16, operands swapped: a lane holds 4 consecutive output columns of one row per
// register quad) are transposed into a row-major fp32 image 64 rows at a time, and every global access of the epilogue
// (bias, residual, GELU aux, outputs) is then a row-contiguous 8-16 B/lane access instead of a 32-byte-per-row scatter.
#pragma once
#include "mmt_common.h"
#include "../../include/mmt_hip.h"
#include "gelu_lut.h"

typedef __attribute__((ext_vector_type(16))) float f32x16;

// erf-form GELU (model/bert.py:37-53) of a bf16-ROUNDED pre-activation, by table: the epilogues apply the activation to
// the value they store (backward differentiates exactly what was applied), so its argument is one of 65536 bf16 numbers
// and Phi(|x|) / gelu'(|x|) for the 2304 magnitudes in [2^-15, 8) are EXACT table entries (gelu_lut.h, double precision
// rounded to fp32; smaller / larger magnitudes clamp: relative error < 1.3e-5 / none).  Per element: and, med3, LDS read,
// sign fix, multiply -- the Abramowitz-Stegun form it replaces (|erf error| 1.5e-7) cost one v_rcp + one v_exp + a dozen
// fma: ~45 of the ~110 issue slots of a 4-element sweep step, and the sweep is VALU-bound (tools/gemm2_budget.py r04).
__device__ __forceinline__ float gelu_lut_abs(const float* __restrict__ lut, unsigned b) {
  const unsigned mag = b & 0x7fffu;
  const unsigned idx = min(max(mag, (unsigned)MMT_GELU_LUT_LO), (unsigned)(MMT_GELU_LUT_LO + MMT_GELU_LUT_N - 1)) - MMT_GELU_LUT_LO;
  return lut[idx];
}
// x Phi(x) for x = the bf16 number with bits b (lut = the Phi table in LDS)
__device__ __forceinline__ float gelu_lut(const float* __restrict__ lut, unsigned b) {
  const float t = gelu_lut_abs(lut, b);
  return bf2f((bf16_t)b) * ((b & 0x8000u) ? 1.0f - t : t);
}
// d/dx [x Phi(x)] (lut = the derivative table): g'(-x) = 1 - g'(x)
__device__ __forceinline__ float gelu_grad_lut(const float* __restrict__ lut, unsigned b) {
  const float t = gelu_lut_abs(lut, b);
  return (b & 0x8000u) ? 1.0f - t : t;
}

// acc[MI][NJ]: wave (wm, wn) of a WGM x WGN grid owns rows wm * WTM + 32 i .. and columns wn * WTN + 32 j .. of the BM x BN
// tile at (m0, n0).  PH: two wave groups (kg = 0 / 1) hold partial tiles over alternate K-steps; group 1 adds its partial
// onto group 0's image.  smem_raw: >= (64 (BN + 4) + RG BN) floats, no longer in use by the main loop.  ticks (lab build):
// [0] = LDS staging, [1] = row sweep, in s_memtime cycles.
template <int BM, int BN, int WGM, int WGN, int NT, int EPI, bool PH>
__device__ __forceinline__ void gemm_tile_epilogue(f32x16 (&acc)[BM / WGM / 32][BN / WGN / 32], unsigned char* smem_raw, int m0, int n0,
                                                   int M, int N, int nrows, void* __restrict__ Cout, int64_t ldc,
                                                   const MmtEpilogue& epi, int wm, int wn, int kg, int tid, long long* ticks) {
  constexpr int WTM = BM / WGM, WTN = BN / WGN, MI = WTM / 32, NJ = WTN / 32;
  constexpr int P = BN + 4;        // fp32 pitch of the epilogue image
  // the epilogue sweeps the image in column blocks of CB columns: the whole width when the thread count divides into
  // whole rows of it, 64-column blocks otherwise (BN = 192: 3 blocks of 16 lanes x 16 B per row)
  constexpr int CB = (NT * 4) % BN == 0 ? BN : 64;
  constexpr int NCB = BN / CB;     // column blocks
  constexpr int CG = CB / 4;       // 4-column groups per row of a column block
  constexpr int RG = NT / CG;      // rows covered per sweep of the block
  constexpr int CH = BM < 64 ? BM : 64;  // rows per epilogue chunk
  static_assert(BN % CB == 0 && NT % CG == 0 && CH % RG == 0, "epilogue geometry");
  const int lane = tid & 63, l31 = lane & 31, lh = lane >> 5;
  // ---- epilogue: 64 rows at a time through a row-major fp32 LDS image -------------------------------
  float* st = (float*)smem_raw;
  float* red = st + CH * P;  // [RG][BN] column-sum scratch (DGELU)
  const int cg = tid % CG, rg = tid / CG;
  f32x4 bias4[NCB];
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) {
    bias4[cb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if constexpr (EPI == MMT_EPI_BIAS_BF16 || EPI == MMT_EPI_BIAS_GELU || EPI == MMT_EPI_BIAS_DROP_RES ||
                  EPI == MMT_EPI_BIAS_F32)
      bias4[cb] = *(const f32x4*)(epi.bias + n0 + cb * CB + cg * 4);
  }
  unsigned dkey = 0;
  if constexpr (EPI == MMT_EPI_BIAS_DROP_RES) dkey = eff_key(epi.drop_key, epi.seed_dev);
  float csum[NCB][4];
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) csum[cb][0] = csum[cb][1] = csum[cb][2] = csum[cb][3] = 0.f;
  // Second operands of the epilogue (residual rows, GELU pre-activations, original row numbers of the dropout key) for
  // the WHOLE tile go out now, all at once: fetched inside the sweep, each sits behind the previous row's store -- a
  // chain of (BM / RG) global round trips per thread at the end of every tile.
  // Element offsets of this thread's first swept row (m0 + rg, column n0 + 4 cg) in every matrix the epilogue touches: row
  // m0 + rg + dr is that + dr * ld with dr a compile-time constant of the unrolled sweep -- a scalar multiply and one 64-bit
  // add instead of a 64-bit row x ld product per access (two quarter-rate v_mul_lo_u32 + v_mad_u64_u32: a fifth of the
  // sweep's instruction slots, tools/gemm2_budget.py r04).
  const int64_t rb = m0 + rg, cb0 = n0 + cg * 4;
  const int64_t off_c = rb * ldc + cb0, off_res = rb * epi.ldres + cb0, off_aux = rb * epi.ldaux + cb0,
                off_o2 = rb * epi.ldout2 + cb0, off_dot = rb * epi.lddot + cb0;
  constexpr int SW = CH / RG, NPF = (BM / CH) * SW * NCB;
  constexpr bool PF_FITS = NPF <= 8;  // (256-row tiles would spend > 64 registers on it: they prefetch chunk by chunk, below)
  constexpr bool PFC = !PF_FITS && SW * NCB <= 8;  // second operands of ONE 64-row chunk, fetched before its staging pass
  constexpr bool PF_RES = PF_FITS && (EPI == MMT_EPI_BIAS_DROP_RES || EPI == MMT_EPI_ADD_F32);
  constexpr bool PF_AUX = PF_FITS && (EPI == MMT_EPI_DGELU || EPI == MMT_EPI_BF16);  // (BF16: dot_src, when dot_out is set)
  f32x4 pf_res[PF_RES ? NPF : 1];
  u32x2 pf_aux[PF_AUX ? NPF : 1];
  int pf_orow[PF_RES && EPI == MMT_EPI_BIAS_DROP_RES ? (BM / CH) * SW : 1];
  if constexpr (PF_RES || PF_AUX) {
#pragma unroll
    for (int ch = 0; ch < BM / CH; ++ch)
#pragma unroll
      for (int sw = 0; sw < SW; ++sw) {
        const int dr = ch * CH + sw * RG;
        const bool in = m0 + dr + rg < M;  // rows past the matrix read its last row (never used)
        const int row = in ? m0 + dr + rg : M - 1;
        if constexpr (PF_RES && EPI == MMT_EPI_BIAS_DROP_RES)
          pf_orow[ch * SW + sw] = (epi.drop_thr16 && epi.row_index) ? epi.row_index[row] : row;
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
          if constexpr (PF_RES) {
            const int64_t o = in ? off_res + (int64_t)dr * epi.ldres + cb * CB : (int64_t)(M - 1) * epi.ldres + cb0 + cb * CB;
            pf_res[(ch * SW + sw) * NCB + cb] = *(const f32x4*)(epi.res + o);
          }
          if constexpr (PF_AUX && EPI == MMT_EPI_DGELU) {
            const int64_t o = in ? off_aux + (int64_t)dr * epi.ldaux + cb * CB : (int64_t)(M - 1) * epi.ldaux + cb0 + cb * CB;
            pf_aux[(ch * SW + sw) * NCB + cb] = *(const u32x2*)((const bf16_t*)epi.aux + o);
          }
          if constexpr (PF_AUX && EPI == MMT_EPI_BF16) {
            if (epi.dot_out) {
              const int64_t o = in ? off_dot + (int64_t)dr * epi.lddot + cb * CB : (int64_t)(M - 1) * epi.lddot + cb0 + cb * CB;
              pf_aux[(ch * SW + sw) * NCB + cb] = *(const u32x2*)((const bf16_t*)epi.dot_src + o);
            }
          }
        }
      }
  }
  // GELU tables: global -> registers now (the round trip overlaps the barrier and the first staging pass), -> LDS below
  constexpr bool LUT = EPI == MMT_EPI_BIAS_GELU || EPI == MMT_EPI_DGELU;
  constexpr int LUTN = (MMT_GELU_LUT_N + NT - 1) / NT;
  float* lut = red + RG * BN;
  float lutv[LUT ? LUTN : 1];
  if constexpr (LUT) {
    const float* src = EPI == MMT_EPI_BIAS_GELU ? g_gelu_lut_cdf : g_gelu_lut_dgelu;
#pragma unroll
    for (int i = 0; i < LUTN; ++i) lutv[i] = src[min(i * NT + tid, MMT_GELU_LUT_N - 1)];
  }
  __syncthreads();  // every wave is done with the stage buffers
  if constexpr (LUT) {
#pragma unroll
    for (int i = 0; i < LUTN; ++i)
      if (i * NT + tid < MMT_GELU_LUT_N) lut[i * NT + tid] = lutv[i];
  }
#ifdef MMT_GEMM2_INSTR
  long long e_stage = 0, e_sweep = 0, tp = clock64();
  const long long e_t0 = tp;
#define ETICK(acc) do { const long long tn_ = clock64(); acc += tn_ - tp; tp = tn_; } while (0)
#else
#define ETICK(acc) do {} while (0)
#endif
#pragma unroll
  for (int ch = 0; ch < BM / CH; ++ch) {
    f32x4 pc_res[PFC && (EPI == MMT_EPI_BIAS_DROP_RES || EPI == MMT_EPI_ADD_F32) ? SW * NCB : 1];
    u32x2 pc_aux[PFC && EPI == MMT_EPI_DGELU ? SW * NCB : 1];
    if constexpr (PFC && (EPI == MMT_EPI_BIAS_DROP_RES || EPI == MMT_EPI_ADD_F32 || EPI == MMT_EPI_DGELU)) {
#pragma unroll
      for (int sw = 0; sw < SW; ++sw) {
        const int dr = ch * CH + sw * RG;
        const bool in = m0 + dr + rg < M;
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
          if constexpr (EPI == MMT_EPI_DGELU) {
            const int64_t o = in ? off_aux + (int64_t)dr * epi.ldaux + cb * CB : (int64_t)(M - 1) * epi.ldaux + cb0 + cb * CB;
            pc_aux[sw * NCB + cb] = *(const u32x2*)((const bf16_t*)epi.aux + o);
          } else {
            const int64_t o = in ? off_res + (int64_t)dr * epi.ldres + cb * CB : (int64_t)(M - 1) * epi.ldres + cb0 + cb * CB;
            pc_res[sw * NCB + cb] = *(const f32x4*)(epi.res + o);
          }
        }
      }
    }
    if (kg == 0) {  // the fragment rows of this wave that belong to chunk ch (a wave tile may span several chunks)
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        if ((wm * WTM + i * 32) / CH != ch) continue;
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            f32x4 v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
            *(f32x4*)(st + ((wm * WTM + i * 32) % CH + l31) * P + wn * WTN + j * 32 + 8 * q + 4 * lh) = v;
          }
      }
    }
    if constexpr (PH) {  // the other group's partial tile (odd K-steps) is added into the image: even + odd, fixed order
      __syncthreads();
      if (kg == 1) {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          if ((wm * WTM + i * 32) / CH != ch) continue;
#pragma unroll
          for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              float* dst = st + ((wm * WTM + i * 32) % CH + l31) * P + wn * WTN + j * 32 + 8 * q + 4 * lh;
              f32x4 v = *(const f32x4*)dst;
              v += (f32x4){acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
              *(f32x4*)dst = v;
            }
        }
      }
    }
    __syncthreads();
    ETICK(e_stage);
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
      const int lcol = cb * CB + cg * 4;  // column inside the tile
      const int col = n0 + lcol;
#pragma unroll
      for (int r0 = 0; r0 < CH; r0 += RG) {
        const int r = r0 + rg;
        const int row = m0 + ch * CH + r;
        const int dr = ch * CH + r0;  // (compile-time after unrolling)
        const int64_t o_c = off_c + (int64_t)dr * ldc + cb * CB;
        if (row < M) {
          f32x4 v = *(const f32x4*)(st + r * P + lcol);
          v += bias4[cb];
          if constexpr (EPI == MMT_EPI_BF16 || EPI == MMT_EPI_BIAS_BF16) {
            u32x2 o = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
            *(u32x2*)((bf16_t*)Cout + o_c) = o;
            if constexpr (EPI == MMT_EPI_BF16) {
              if (epi.dot_out) {  // sums of out * dot_src over each 64-column group of the row: 16 neighbouring lanes x 4 columns
                static_assert(CG % 16 == 0, "a 64-column group is 16 lanes of one row");
                u32x2 c;
                if constexpr (PF_AUX) c = pf_aux[(ch * SW + r0 / RG) * NCB + cb];
                else c = *(const u32x2*)((const bf16_t*)epi.dot_src + off_dot + (int64_t)dr * epi.lddot + cb * CB);
                float part = bf2f((bf16_t)(o[0] & 0xffff)) * bf2f((bf16_t)(c[0] & 0xffff)) + bf2f((bf16_t)(o[0] >> 16)) * bf2f((bf16_t)(c[0] >> 16)) +
                             bf2f((bf16_t)(o[1] & 0xffff)) * bf2f((bf16_t)(c[1] & 0xffff)) + bf2f((bf16_t)(o[1] >> 16)) * bf2f((bf16_t)(c[1] >> 16));
                part += __shfl_xor(part, 1, 64); part += __shfl_xor(part, 2, 64);
                part += __shfl_xor(part, 4, 64); part += __shfl_xor(part, 8, 64);
                if ((cg & 15) == 0) epi.dot_out[(int64_t)row * (N >> 6) + (col >> 6)] = part;
              }
            }
          } else if constexpr (EPI == MMT_EPI_BIAS_GELU) {
            u32x2 o = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
            *(u32x2*)((bf16_t*)Cout + o_c) = o;
            // GELU of the bf16-rounded pre-activation: backward differentiates exactly what was applied
            u32x2 g = {pack_bf2(gelu_lut(lut, o[0] & 0xffff), gelu_lut(lut, o[0] >> 16)),
                       pack_bf2(gelu_lut(lut, o[1] & 0xffff), gelu_lut(lut, o[1] >> 16))};
            *(u32x2*)((bf16_t*)epi.out2 + off_o2 + (int64_t)dr * epi.ldout2 + cb * CB) = g;
          } else if constexpr (EPI == MMT_EPI_BIAS_DROP_RES) {
            if (epi.drop_thr16) {
              int orow;
              if constexpr (PF_RES) orow = pf_orow[ch * SW + r0 / RG];
              else orow = epi.row_index ? epi.row_index[row] : row;
              bool k[4];
              keep4(dkey, (unsigned long long)orow * (unsigned)N + (unsigned)col, epi.drop_thr16, k);
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = k[e] ? v[e] * epi.drop_scale : 0.f;
            }
            if constexpr (PF_RES) v += pf_res[(ch * SW + r0 / RG) * NCB + cb];
            else if constexpr (PFC) v += pc_res[(r0 / RG) * NCB + cb];
            else v += *(const f32x4*)(epi.res + off_res + (int64_t)dr * epi.ldres + cb * CB);
            *(f32x4*)((float*)Cout + o_c) = v;
          } else if constexpr (EPI == MMT_EPI_DGELU) {
            u32x2 a;
            if constexpr (PF_AUX) a = pf_aux[(ch * SW + r0 / RG) * NCB + cb];
            else if constexpr (PFC) a = pc_aux[(r0 / RG) * NCB + cb];
            else a = *(const u32x2*)((const bf16_t*)epi.aux + off_aux + (int64_t)dr * epi.ldaux + cb * CB);
            v[0] *= gelu_grad_lut(lut, a[0] & 0xffff);
            v[1] *= gelu_grad_lut(lut, a[0] >> 16);
            v[2] *= gelu_grad_lut(lut, a[1] & 0xffff);
            v[3] *= gelu_grad_lut(lut, a[1] >> 16);
            u32x2 o = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
            *(u32x2*)((bf16_t*)Cout + o_c) = o;
            if (row < nrows) {
              csum[cb][0] += bf2f((bf16_t)(o[0] & 0xffff)); csum[cb][1] += bf2f((bf16_t)(o[0] >> 16));
              csum[cb][2] += bf2f((bf16_t)(o[1] & 0xffff)); csum[cb][3] += bf2f((bf16_t)(o[1] >> 16));
            }
          } else if constexpr (EPI == MMT_EPI_ADD_F32) {
            if constexpr (PF_RES) v += pf_res[(ch * SW + r0 / RG) * NCB + cb];
            else if constexpr (PFC) v += pc_res[(r0 / RG) * NCB + cb];
            else v += *(const f32x4*)(epi.res + off_res + (int64_t)dr * epi.ldres + cb * CB);
            *(f32x4*)((float*)Cout + o_c) = v;
          } else {  // MMT_EPI_F32 / MMT_EPI_BIAS_F32
            *(f32x4*)((float*)Cout + o_c) = v;
          }
        }
      }
    }
    if constexpr (EPI == MMT_EPI_DGELU) {
      if (epi.colsum && BM >= 128 && (ch & 1)) {  // one partial row of column sums per 128 output rows
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
          *(f32x4*)(red + rg * BN + cb * CB + cg * 4) = (f32x4){csum[cb][0], csum[cb][1], csum[cb][2], csum[cb][3]};
          csum[cb][0] = csum[cb][1] = csum[cb][2] = csum[cb][3] = 0.f;
        }
        __syncthreads();
        const int half_row = m0 / 128 + (ch >> 1);
        if (tid < BN && half_row * 128 < M) {
          float s = 0.f;
#pragma unroll
          for (int g = 0; g < RG; ++g) s += red[g * BN + tid];
          epi.colsum[(int64_t)half_row * N + n0 + tid] = s;
        }
      }
    }
    __syncthreads();
    ETICK(e_sweep);
  }
#ifdef MMT_GEMM2_INSTR
  if (ticks) { ticks[0] = e_stage; ticks[1] = e_sweep; ticks[2] = e_t0; }
#endif
#undef ETICK
}
