// Row-sharded similarity + max-margin ranking loss for very large global batches (BASELINE.json configs[4]:
// n = 64k pairs over 8 ranks).  The reference materialises six length-2n^2 index vectors on
// the host (model/loss.py:55-63) and an [n, n, M] weight tensor (model/model.py:806-819) -- infeasible beyond a few
// thousand pairs.  Here each rank owns a ROW BLOCK: its b texts against all n videos.
//
//   S[t][v]   = <T'[t], V'[v]> / den(t, v),   T' = tw (.) T, V' = vw (.) V folded over K = M*d   -> ONE bf16 MFMA GEMM
//   pass 1    : per local row  rowcnt[t] = #{c != r : m - s_rr + s_rc > 0},  colcnt[c] += [m - s_cc + s_rc > 0],
//               loss partial   (integer atomics => deterministic; colcnt is all-reduced across ranks by the host)
//   pass 2    : G'[t][v] = dL/dS / den  (bf16, operand of the two backward GEMMs) and gs[t][m] = sum_v G' S vw[v][m]
//   backward  : P = G' V'  (b x Md),  Q = G'^T T'  (n x Md, reduce-scattered across ranks by the host),
//               dT[t][m] = tw P,  dtw[t][m] = <T[t][m], P[t][m]> - gs[t][m]      (and the mirror for V)
// Kernels here are the HBM-bound passes over the row block; the GEMMs reuse gemm2.hip / wgrad_grouped.
#include <type_traits>

#include "mmt_common.h"
#include "../../include/mmt_hip.h"

#define LS_MAXM MMT_MAX_EXPERTS

// out16[r][m*d + c] = bf16(w[r][m] * x[r][m][c]); rows r >= R (up to Rpad) are zero-filled
__global__ __launch_bounds__(256) void fold_bf16_kernel(const float* __restrict__ x, const float* __restrict__ w, int R,
                                                        int Rpad, int M, int d, bf16_t* __restrict__ out16) {
  const int d4 = d >> 2;
  const int64_t n4 = (int64_t)Rpad * M * d4;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t rm = i / d4;
    u32x2 o = {0u, 0u};
    if (rm < (int64_t)R * M) {
      const f32x4 v = ((const f32x4*)x)[i] * w[rm];
      o[0] = pack_bf2(v[0], v[1]); o[1] = pack_bf2(v[2], v[3]);
    }
    ((u32x2*)out16)[i] = o;
  }
}

// ---- the passes over the b x n row block (r04) --------------------------------------------------------------------
// One block = TR rows x up to LS_CPB columns; a thread owns 4 consecutive columns of every 1024-column chunk it sweeps and
// all TR rows of them: 16-byte loads of S (a 4 KiB row segment per block and load), the column data (video weights,
// diagonal) read ONCE per chunk into registers and reused for TR rows, the row data (text weights, s_rr) in LDS.  The r01-r03
// kernels walked a row per block with 4-byte loads and re-read vw[c][0..M) + diag[c] for every element (0.42 TB/s on the
// gradient pass; profiles/r03_pmc_config4.txt).  Per-row sums over the block's columns are reduced once per block and
// written as per-column-block partials (fixed order downstream => deterministic); the integer counts use atomics.
#define LS_CHUNK 1024
#define LS_CPB 8192  // columns per block (8 chunks)
static inline int ls_col_blocks(int n) { return (n + LS_CPB - 1) / LS_CPB; }
extern "C" int mmt_ls_col_blocks(int n) { return n > 0 ? ls_col_blocks(n) : MMT_ERR_ARG; }

// vwt (nullable): the video weights TRANSPOSED, [M, n] -- one coalesced 16-byte load per expert for the thread's four
// columns.  From the [n, M] layout the same data is 4 * M four-byte loads per thread, each instruction spread over
// 64 * M * 16 bytes: 28 load instructions x 56 cache lines per wave and chunk at M = 7 against 16 x 8 for the row block
// itself -- the sweeps ran at 1.5 TB/s on that, not on HBM.
template <int MM>
__device__ __forceinline__ void ls_load_cols(const float* __restrict__ vw, const float* __restrict__ vwt,
                                             const float* __restrict__ diag, int c0, int n, int M, float (&vwc)[4][MM], f32x4& dg) {
  if (vwt) {
#pragma unroll
    for (int m = 0; m < MM; ++m) {
      const f32x4 v = m < M ? *(const f32x4*)(vwt + (int64_t)m * n + c0) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 4; ++j) vwc[j][m] = v[j];
    }
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int m = 0; m < MM; ++m) vwc[j][m] = (vw && m < M && c0 + j < n) ? vw[(int64_t)(c0 + j) * M + m] : 0.f;
  }
  dg = diag ? *(const f32x4*)(diag + c0) : (f32x4){0.f, 0.f, 0.f, 0.f};
}

// similarity = numerator / den as ONE v_rcp_f32 and ONE v_mul_f32 (inline asm: opaque to -ffast-math, so the two
// instructions are the same wherever the similarity is recomputed -- the counting sweep, the gradient pass, the diagonal --
// and the hinge decisions of the passes agree bit for bit).  <= 1.5 ulp from the correctly rounded quotient.
// (v_rcp_f32 sits inside the asm with its wait state: the compiler's hazard recognizer does not look into inline asm, and a
// VALU instruction reading a transcendental result one slot later reads the OLD register on gfx950)
__device__ __forceinline__ float ls_rcp(float den) {
  float r;
  asm("v_rcp_f32 %0, %1\n\ts_nop 0" : "=v"(r) : "v"(den));
  return r;
}
__device__ __forceinline__ float ls_quot(float num, float rcp_den) {
  float r;
  asm("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(num), "v"(rcp_den));
  return r;
}
// den(t, v) = sum_m tw[t][m] vw[v][m] as a fixed fma chain; 0 -> 1e-5 (model.py:816)
template <int MM>
__device__ __forceinline__ float ls_den(const float (&twr)[MM], const float (&vwj)[MM], bool* zero) {
  float den = 0.f;
#pragma unroll
  for (int m = 0; m < MM; ++m) den = __builtin_fmaf(twr[m], vwj[m], den);
  *zero = den == 0.f;
  return *zero ? 1e-5f : den;
}

// raw numerators -> similarities (MODE 1, in place) and / or pass 1 (MODE 2: counts; MODE 3: both in one sweep; MODE 7: both,
// but the similarities are NOT written back -- the gradient pass divides again (RAW) and the row block is never rewritten).
//   rowcnt[t] = #{c != r : m - s_rr + s_rc > 0},  colcnt[c] += [m - s_cc + s_rc > 0],  loss_part[t][cb] = hinge sums
// Two copies of the chunk body: CHECKED (rows beyond b and the diagonal element tested per row / per element) for the last
// row block and for the few threads whose four columns cross the block's diagonal, and the plain one for everything else.
// The r04 first version tested everywhere: 46 VALU instructions per element, 512 VGPRs + spills, one wave per SIMD, and
// 1.5 TB/s.  Row counts are wave ballots (v_cmp + s_bcnt1: the scalar unit counts, nothing to reduce at the end).
template <int TR, int MM, int MODE>
__global__ __launch_bounds__(256, 2) void ls_sweep_kernel(float* __restrict__ S, int64_t ld, const float* __restrict__ diag,
                                                          const float* __restrict__ tw, const float* __restrict__ vw,
                                                          const float* __restrict__ vwt, int b, int n,
                                                          int M, int r0, float margin, int32_t* __restrict__ rowcnt,
                                                          int32_t* __restrict__ colcnt, float* __restrict__ loss_part) {
  constexpr bool FIN = MODE & 1, CNT = MODE & 2, STORE = !(MODE & 4);
  __shared__ float tws[TR][MM];
  __shared__ float ms_s[TR];  // margin - s_rr
  __shared__ float redf[4][TR];
  __shared__ int redi[4][TR];
  const int cb = blockIdx.x, ncb = gridDim.x, t0 = blockIdx.y * TR, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if constexpr (FIN) {
    for (int e = tid; e < TR * MM; e += 256) {
      const int r = e / MM, m = e % MM;
      tws[r][m] = (t0 + r < b && m < M) ? tw[(int64_t)(t0 + r) * M + m] : 0.f;
    }
  }
  if (CNT && tid < TR) ms_s[tid] = t0 + tid < b ? margin - diag[r0 + t0 + tid] : 0.f;
  __syncthreads();
  float acc[TR];
  int cnt[TR];  // wave totals (ballot counts: the same value in every lane)
#pragma unroll
  for (int r = 0; r < TR; ++r) { acc[r] = 0.f; cnt[r] = 0; }
  const bool full = t0 + TR <= b;
  const int rg0 = r0 + t0;  // global column of the block's first diagonal element
  const int c_end = min(n, (cb + 1) * LS_CPB);
  for (int c0 = cb * LS_CPB + tid * 4; c0 < c_end; c0 += LS_CHUNK) {
    // (the row data in LDS is loop-invariant: without this the compiler hoists all TR x MM of it into VGPRs and spills)
    asm volatile("" ::: "memory");
    float vwc[4][MM];
    f32x4 dg;
    ls_load_cols<MM>(FIN ? vw : nullptr, FIN ? vwt : nullptr, CNT ? diag : nullptr, c0, n, M, vwc, dg);
    f32x4 sv[TR];
#pragma unroll
    for (int r = 0; r < TR; ++r)  // (rows beyond b re-read row b - 1; nothing of them is kept)
      sv[r] = *(const f32x4*)(S + (int64_t)min(t0 + r, b - 1) * ld + c0);
    float md[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) md[j] = margin - dg[j];
    int cc[4] = {0, 0, 0, 0};
    auto rows = [&](auto checked_t) {
      constexpr bool CHECKED = decltype(checked_t)::value;
#pragma unroll
      for (int r = 0; r < TR; ++r) {
        if (CHECKED && t0 + r >= b) continue;  // (block-uniform)
        f32x4 sr = sv[r];
        if constexpr (FIN) {
          float twr[MM];
#pragma unroll
          for (int m = 0; m < MM; ++m) twr[m] = tws[r][m];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            bool zero;
            const float den = ls_den<MM>(twr, vwc[j], &zero);
            sr[j] = ls_quot(sr[j], ls_rcp(den));
          }
          if constexpr (STORE) *(f32x4*)(S + (int64_t)(t0 + r) * ld + c0) = sr;
        }
        if constexpr (CNT) {
          const float ms = ms_s[r];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const bool skip = CHECKED && c0 + j == rg0 + r;  // the diagonal element carries no hinge
            const float h1 = ms + sr[j], h2 = md[j] + sr[j];
            const bool p1 = h1 > 0.f && !skip, p2 = h2 > 0.f && !skip;
            acc[r] += (p1 ? h1 : 0.f) + (p2 ? h2 : 0.f);
            cnt[r] += __builtin_popcountll(__ballot(p1));  // (the branch below is wave-uniform: every lane is here)
            cc[j] += p2;
          }
        }
      }
    };
    // (wave-uniform choice: one lane crossing the diagonal sends its whole wave through the checked copy)
    if (full && !(CNT && __ballot(c0 < rg0 + TR && c0 + 4 > rg0))) rows(std::false_type{});
    else rows(std::true_type{});
    if constexpr (CNT) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (cc[j]) atomicAdd(colcnt + c0 + j, cc[j]);
    }
  }
  if constexpr (CNT) {
#pragma unroll
    for (int r = 0; r < TR; ++r) {
      const float a = wave_sum(acc[r]);
      if (lane == 0) { redf[wave][r] = a; redi[wave][r] = cnt[r]; }
    }
    __syncthreads();
    if (tid < TR && t0 + tid < b) {
      loss_part[(int64_t)(t0 + tid) * ncb + cb] = (redf[0][tid] + redf[1][tid]) + (redf[2][tid] + redf[3][tid]);
      atomicAdd(rowcnt + t0 + tid, redi[0][tid] + redi[1][tid] + redi[2][tid] + redi[3][tid]);
    }
  }
}

// diag_local[t] = S[t][r0 + t] / den(t, r0 + t) from the RAW numerators (before mmt_ls_counts_ex(finish = 1) divides them)
__global__ __launch_bounds__(256) void ls_diag_kernel(const float* __restrict__ S, int64_t ld, const float* __restrict__ tw,
                                                      const float* __restrict__ vw, int b, int M, int r0,
                                                      float* __restrict__ diag_local) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= b) return;
  float den = 0.f;
  for (int m = 0; m < M; ++m) den = __builtin_fmaf(tw[(int64_t)t * M + m], vw[(int64_t)(r0 + t) * M + m], den);
  if (den == 0.f) den = 1e-5f;
  diag_local[t] = ls_quot(S[(int64_t)t * ld + r0 + t], ls_rcp(den));
}

// pass 2: G'[t][v] (bf16) = g(t, v) / den(t, v) with g = ((h1 > 0) + (h2 > 0)) / norm off the diagonal and
// -(rowcnt[t] + colcnt[r]) / norm on it; gs_part[t][cb][m] = sum over the block's columns of G'[t][v] S[t][v] vw[v][m]
// RAW: S holds the raw numerators (mmt_ls_counts_ex(finish = 2)); the division is redone here, bit for bit (ls_quot).
// Checked / plain copies of the chunk body as in the sweep above.
template <int TR, int MM, bool RAW>
__global__ __launch_bounds__(256, 2) void ls_grad2_kernel(const float* __restrict__ S, int64_t ld, const float* __restrict__ diag,
                                                          const float* __restrict__ tw, const float* __restrict__ vw,
                                                          const float* __restrict__ vwt,
                                                          const int32_t* __restrict__ rowcnt, const int32_t* __restrict__ colcnt,
                                                          int b, int n, int M, int r0, float margin, float inv_norm,
                                                          bf16_t* __restrict__ G16, int64_t ldg, float* __restrict__ gs_part) {
  __shared__ float tws[TR][MM];
  __shared__ float ms_s[TR], gdiag_s[TR];
  __shared__ float red[4][TR][MM];
  const int cb = blockIdx.x, ncb = gridDim.x, t0 = blockIdx.y * TR, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int e = tid; e < TR * MM; e += 256) {
    const int r = e / MM, m = e % MM;
    tws[r][m] = (t0 + r < b && m < M) ? tw[(int64_t)(t0 + r) * M + m] : 0.f;
  }
  if (tid < TR && t0 + tid < b) {
    ms_s[tid] = margin - diag[r0 + t0 + tid];
    gdiag_s[tid] = -(float)(rowcnt[t0 + tid] + colcnt[r0 + t0 + tid]) * inv_norm;
  }
  __syncthreads();
  float acc[TR][MM];
#pragma unroll
  for (int r = 0; r < TR; ++r)
#pragma unroll
    for (int m = 0; m < MM; ++m) acc[r][m] = 0.f;
  const bool full = t0 + TR <= b;
  const int rg0 = r0 + t0;
  const int c_end = min(n, (cb + 1) * LS_CPB);
  for (int c0 = cb * LS_CPB + tid * 4; c0 < c_end; c0 += LS_CHUNK) {
    // (the row data in LDS is loop-invariant: without this the compiler hoists all TR x MM of it into VGPRs and spills)
    asm volatile("" ::: "memory");
    float vwc[4][MM];
    f32x4 dg;
    ls_load_cols<MM>(vw, vwt, diag, c0, n, M, vwc, dg);
    f32x4 sv[TR];
#pragma unroll
    for (int r = 0; r < TR; ++r) sv[r] = *(const f32x4*)(S + (int64_t)min(t0 + r, b - 1) * ld + c0);
    float md[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) md[j] = margin - dg[j];
    auto rows = [&](auto checked_t) {
      constexpr bool CHECKED = decltype(checked_t)::value;
#pragma unroll
      for (int r = 0; r < TR; ++r) {
        if (CHECKED && t0 + r >= b) continue;  // (block-uniform)
        const float ms = ms_s[r];
        float twr[MM];
#pragma unroll
        for (int m = 0; m < MM; ++m) twr[m] = tws[r][m];
        float gq[4], gpb[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          bool zero;
          const float rden = ls_rcp(ls_den<MM>(twr, vwc[j], &zero));
          const float sj = RAW ? ls_quot(sv[r][j], rden) : sv[r][j];
          float g = ((ms + sj > 0.f) ? inv_norm : 0.f) + ((md[j] + sj > 0.f) ? inv_norm : 0.f);
          if (CHECKED && c0 + j == rg0 + r) g = gdiag_s[r];
          gq[j] = ls_quot(g, rden);
          // the 1e-5 branch carries no normaliser gradient (model.py:816)
          gpb[j] = zero ? 0.f : sj;
        }
        const u32x2 o = {pack_bf2(gq[0], gq[1]), pack_bf2(gq[2], gq[3])};
        *(u32x2*)(G16 + (int64_t)(t0 + r) * ldg + c0) = o;
        gpb[0] *= __uint_as_float(o[0] << 16); gpb[1] *= __uint_as_float(o[0] & 0xffff0000u);
        gpb[2] *= __uint_as_float(o[1] << 16); gpb[3] *= __uint_as_float(o[1] & 0xffff0000u);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int m = 0; m < MM; ++m) acc[r][m] += gpb[j] * vwc[j][m];
      }
    };
    if (full && !__ballot(c0 < rg0 + TR && c0 + 4 > rg0)) rows(std::false_type{});
    else rows(std::true_type{});
  }
#pragma unroll
  for (int r = 0; r < TR; ++r)
#pragma unroll
    for (int m = 0; m < MM; ++m) {
      const float v = wave_sum(acc[r][m]);
      if (lane == 0) red[wave][r][m] = v;
    }
  __syncthreads();
  for (int e = tid; e < TR * MM; e += 256) {
    const int r = e / MM, m = e % MM;
    if (t0 + r < b && m < M)
      gs_part[((int64_t)(t0 + r) * ncb + cb) * M + m] = (red[0][r][m] + red[1][r][m]) + (red[2][r][m] + red[3][r][m]);
  }
}

// dX[r][m][:] = w[r][m] * P[r][m*d + :]; dw[r][m] = <X[r][m], P[r][m]> - gsub[r][m]   (one wave per (r, m))
__global__ __launch_bounds__(256) void ls_unfold_kernel(const float* __restrict__ P, int64_t ldp, const float* __restrict__ x,
                                                        const float* __restrict__ w, const float* __restrict__ gsub, int R,
                                                        int M, int d, float* __restrict__ dx, float* __restrict__ dw) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = blockIdx.x * 4 + wave; i < R * M; i += gridDim.x * 4) {
    const int r = i / M, m = i % M;
    const float wv = w[i];
    float dot = 0.f;
    for (int c = lane * 4; c < d; c += 256) {
      const f32x4 p = *(const f32x4*)(P + (int64_t)r * ldp + (int64_t)m * d + c);
      const f32x4 xv = *(const f32x4*)(x + (int64_t)i * d + c);
      dot += p[0] * xv[0] + p[1] * xv[1] + p[2] * xv[2] + p[3] * xv[3];
      if (dx) *(f32x4*)(dx + (int64_t)i * d + c) = p * wv;
    }
    dot = wave_sum(dot);
    if (lane == 0 && dw) dw[i] = dot - (gsub ? gsub[i] : 0.f);
  }
}

extern "C" int mmt_ls_fold_bf16(const float* x, const float* w, int R, int Rpad, int M, int d, void* out16, void* stream) {
  if (!x || !w || !out16 || R <= 0 || Rpad < R || M <= 0 || d <= 0 || (d & 3)) return MMT_ERR_ARG;
  const int64_t n4 = (int64_t)Rpad * M * (d >> 2);
  hipLaunchKernelGGL(fold_bf16_kernel, dim3((int)((n4 + 255) / 256 < 8192 ? (n4 + 255) / 256 : 8192)), dim3(256), 0,
                     (hipStream_t)stream, x, w, R, Rpad, M, d, (bf16_t*)out16);
  return (int)hipGetLastError();
}

template <int MODE>
static int ls_sweep(float* S, int64_t ld, const float* diag, const float* tw, const float* vw, const float* vwt, int b, int n, int M, int r0,
                    float margin, int32_t* rowcnt, int32_t* colcnt, float* loss_part, hipStream_t s) {
  if ((n & 3) || (ld & 3) || ((uintptr_t)S & 15)) return MMT_ERR_ALIGN;
  if (M <= 8)
    hipLaunchKernelGGL((ls_sweep_kernel<16, 8, MODE>), dim3(ls_col_blocks(n), (b + 15) / 16), dim3(256), 0, s, S, ld, diag, tw, vw, vwt, b, n,
                       M, r0, margin, rowcnt, colcnt, loss_part);
  else
    hipLaunchKernelGGL((ls_sweep_kernel<8, 16, MODE>), dim3(ls_col_blocks(n), (b + 7) / 8), dim3(256), 0, s, S, ld, diag, tw, vw, vwt, b, n,
                       M, r0, margin, rowcnt, colcnt, loss_part);
  return (int)hipGetLastError();
}

extern "C" int mmt_ls_finish(float* S, int64_t ld, const float* tw, const float* vw, int b, int n, int M, void* stream) {
  if (!S || !tw || !vw || b <= 0 || n <= 0 || M <= 0 || M > LS_MAXM) return MMT_ERR_ARG;
  return ls_sweep<1>(S, ld, nullptr, tw, vw, nullptr, b, n, M, 0, 0.f, nullptr, nullptr, nullptr, (hipStream_t)stream);
}

extern "C" int mmt_ls_diag(const float* S, int64_t ld, const float* tw, const float* vw, int b, int n, int M, int r0,
                           float* diag_local, void* stream) {
  if (!S || !tw || !vw || !diag_local || b <= 0 || M <= 0 || M > LS_MAXM || r0 < 0 || r0 + b > n) return MMT_ERR_ARG;
  hipLaunchKernelGGL(ls_diag_kernel, dim3((b + 255) / 256), dim3(256), 0, (hipStream_t)stream, S, ld, tw, vw, b, M, r0, diag_local);
  return (int)hipGetLastError();
}

// vw_t (nullable): the video weights transposed, [M, n], 16-byte aligned -- the layout the sweeps read fastest.
// rowcnt and colcnt must be zero on entry (they accumulate); loss_part [b, mmt_ls_col_blocks(n)] holds the UN-normalised
// hinge sums of the local rows per column block.  finish = 1: S holds the raw numerators of the similarity GEMM on entry
// and the similarities on return (tw / vw / M needed) -- the division and pass 1 in ONE sweep over the block.
extern "C" int mmt_ls_counts_ex(float* S, int64_t ld, const float* diag, const float* tw, const float* vw, const float* vw_t,
                                int M, int finish,
                                int b, int n, int r0, float margin, int32_t* rowcnt, int32_t* colcnt, float* loss_part,
                                void* stream) {
  if (!S || !diag || !rowcnt || !colcnt || !loss_part || b <= 0 || n <= 1 || r0 < 0 || r0 + b > n) return MMT_ERR_ARG;
  if (finish && (!tw || !vw || M <= 0 || M > LS_MAXM)) return MMT_ERR_ARG;
  if (finish == 2) return ls_sweep<7>(S, ld, diag, tw, vw, vw_t, b, n, M, r0, margin, rowcnt, colcnt, loss_part, (hipStream_t)stream);
  if (finish) return ls_sweep<3>(S, ld, diag, tw, vw, vw_t, b, n, M, r0, margin, rowcnt, colcnt, loss_part, (hipStream_t)stream);
  return ls_sweep<2>(S, ld, diag, nullptr, nullptr, nullptr, b, n, 0, r0, margin, rowcnt, colcnt, loss_part, (hipStream_t)stream);
}
extern "C" int mmt_ls_counts(const float* S, int64_t ld, const float* diag, int b, int n, int r0, float margin,
                             int32_t* rowcnt, int32_t* colcnt, float* loss_part, void* stream) {
  return mmt_ls_counts_ex((float*)S, ld, diag, nullptr, nullptr, nullptr, 0, 0, b, n, r0, margin, rowcnt, colcnt, loss_part, stream);
}

// gs_part: [b, mmt_ls_col_blocks(n), M] partial sums (the caller adds the column blocks up, in order)
extern "C" int mmt_ls_grad_ex(const float* S, int64_t ld, const float* diag, const float* tw, const float* vw,
                              const float* vw_t, const int32_t* rowcnt, const int32_t* colcnt_total, int b, int n, int M, int r0, float margin,
                              float inv_norm, void* G16, int64_t ldg, float* gs_part, int raw, void* stream) {
  if (!S || !diag || !tw || !vw || !rowcnt || !colcnt_total || !G16 || !gs_part || b <= 0 || n <= 1 || M <= 0 || M > LS_MAXM)
    return MMT_ERR_ARG;
  if ((n & 3) || (ld & 3) || (ldg & 3) || ((uintptr_t)S & 15) || ((uintptr_t)G16 & 7)) return MMT_ERR_ALIGN;
#define LS_GRAD(TRR, MMM, RW)                                                                                                  \
  hipLaunchKernelGGL((ls_grad2_kernel<TRR, MMM, RW>), dim3(ls_col_blocks(n), (b + TRR - 1) / TRR), dim3(256), 0, (hipStream_t)stream, \
                     S, ld, diag, tw, vw, vw_t, rowcnt, colcnt_total, b, n, M, r0, margin, inv_norm, (bf16_t*)G16, ldg, gs_part)
  // (TR x MM gs accumulators per thread: 8 rows x 8 experts, or 4 x 16, keep two blocks per CU without spills)
  if (M <= 8) { if (raw) LS_GRAD(8, 8, true); else LS_GRAD(8, 8, false); }
  else { if (raw) LS_GRAD(4, 16, true); else LS_GRAD(4, 16, false); }
#undef LS_GRAD
  return (int)hipGetLastError();
}

// dst[c][r] = src[r][c] for bf16 matrices (rows, cols multiples of 128): the backward GEMMs of the row block are NT GEMMs
// on the 256x256 eight-phase kernel (gemm3.hip), which wants both operands K-contiguous -- V'^T for P = G' V', and G'^T,
// T'^T for Q = G'^T T'.  128 x 128 tiles; a thread transposes 2 x 8 patches in registers (dword = two rows of one column),
// so LDS sees 4-byte accesses only; global reads are 128-byte, writes 256-byte row segments.  HBM-bound: 2 bytes read +
// 2 written per element (torch's .t().contiguous() ran at 1.0 TB/s on the 0.94 GB V' matrix: 1.87 ms).
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const bf16_t* __restrict__ src, int64_t lds_, bf16_t* __restrict__ dst,
                                                             int64_t ldd) {
  __shared__ uint32_t tile[128][65];
  const int t = threadIdx.x;
  const int64_t r0 = (int64_t)blockIdx.y * 128, c0 = (int64_t)blockIdx.x * 128;
  u32x4 a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int pid = i * 256 + t, cg = (pid & 7) + 8 * ((pid >> 6) & 1), p = ((pid >> 3) & 7) + 8 * (pid >> 7);
    const bf16_t* s0 = src + (r0 + 2 * p) * lds_ + c0 + 8 * cg;
    a[i] = *(const u32x4*)s0;
    b[i] = *(const u32x4*)(s0 + lds_);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int pid = i * 256 + t, cg = (pid & 7) + 8 * ((pid >> 6) & 1), p = ((pid >> 3) & 7) + 8 * (pid >> 7);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      tile[8 * cg + 2 * k][p] = (a[i][k] & 0xffffu) | (b[i][k] << 16);
      tile[8 * cg + 2 * k + 1][p] = (a[i][k] >> 16) | (b[i][k] & 0xffff0000u);
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int id = i * 256 + t, q = id & 15, c = id >> 4;
    const u32x4 v = {tile[c][4 * q], tile[c][4 * q + 1], tile[c][4 * q + 2], tile[c][4 * q + 3]};
    *(u32x4*)(dst + (c0 + c) * ldd + r0 + 8 * q) = v;
  }
}

extern "C" int mmt_transpose_bf16(const void* src, int64_t ld_src, int rows, int cols, void* dst, int64_t ld_dst, void* stream) {
  if (!src || !dst || rows <= 0 || cols <= 0 || (rows & 127) || (cols & 127) || ld_src < cols || ld_dst < rows) return MMT_ERR_ARG;
  if ((ld_src & 7) || (ld_dst & 7) || ((uintptr_t)src & 15) || ((uintptr_t)dst & 15)) return MMT_ERR_ALIGN;
  hipLaunchKernelGGL(transpose_bf16_kernel, dim3(cols / 128, rows / 128), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src,
                     ld_src, (bf16_t*)dst, ld_dst);
  return (int)hipGetLastError();
}

extern "C" int mmt_ls_grad(const float* S, int64_t ld, const float* diag, const float* tw, const float* vw,
                           const int32_t* rowcnt, const int32_t* colcnt_total, int b, int n, int M, int r0, float margin,
                           float inv_norm, void* G16, int64_t ldg, float* gs_part, void* stream) {
  return mmt_ls_grad_ex(S, ld, diag, tw, vw, nullptr, rowcnt, colcnt_total, b, n, M, r0, margin, inv_norm, G16, ldg, gs_part, 0, stream);
}

extern "C" int mmt_ls_unfold(const float* P, int64_t ldp, const float* x, const float* w, const float* gsub, int R, int M,
                             int d, float* dx, float* dw, void* stream) {
  if (!P || !x || !w || R <= 0 || M <= 0 || d <= 0 || (d & 3) || d > 1024) return MMT_ERR_ARG;
  int grid = (R * M + 3) / 4;
  if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(ls_unfold_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, P, ldp, x, w, gsub, R, M, d, dx, dw);
  return (int)hipGetLastError();
}
