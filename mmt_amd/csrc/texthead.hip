// Text heads of CENet (gfx950, fp32): per-expert GatedEmbeddingUnit + text MoE weights,
//   model/model.py:683-702 (GatedEmbeddingUnit), :736-750 (ContextGating + BatchNorm1d), :262-283,618 (MoE).
// The reference issues ~50 ATen ops per expert; here the M experts are batched:
//   sgemm_batched : C_b = [beta*C_b +] A_b . B_b^T(strided) [+ bias_b]   generic-stride fp32 GEMM (tiny N = B*C rows)
//   bn_stats      : per (expert, column) batch mean / rstd (+ running-stat update, momentum 0.1)
//   gate_norm     : e = F.normalize(y * sigmoid(BN(x1)))                  written in (B, M, C, d) layout
//   moe           : softmax_m(text . w_m + b_m)
// and the matching backward kernels.  Activations are [N, M, d] (row n contiguous over experts) so that
// the text gradient is ONE GEMM over K = M*d.  Arithmetic is fp32 like the reference (the work is ~0.3 GFLOP).
#include <stdlib.h>
#include "mmt_common.h"
#include "../../include/mmt_hip.h"

typedef __attribute__((ext_vector_type(16))) float f32x16;

// Generic-stride batched fp32 GEMM on the exact-fp32 matrix cores (v_mfma_f32_32x32x2_f32: an fp32 fma chain, so
// the arithmetic stays fp32 like the reference).  The text heads have only N = B*C (~32) rows, so a 32x32 output
// tile per WAVE is the natural unit: each weight element is streamed from HBM exactly once.
//   ksplit == SG_WAVES : the 16 waves of a block split the contraction of ONE tile and reduce through LDS (deep K:
//                        the 64-cycle fp32 MFMA chain per wave stays short, 16x more waves stream the weights);
//   ksplit == 1        : the 16 waves own 16 neighbouring tiles (K = N rows for the weight gradients).
#define SG_WAVES 16
__global__ __launch_bounds__(64 * SG_WAVES) void sgemm_batched_kernel(MmtSgemm g, int ksplit) {
  __shared__ float red[SG_WAVES - 1][16][64];
  const int b = blockIdx.z;
  const float* __restrict__ A = g.A[b];
  const float* __restrict__ B = g.B[b];
  float* __restrict__ C = g.C[b];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
  int tile_n = blockIdx.x, kbeg = 0, kend = g.K;
  if (ksplit > 1) {
    const int per = (((g.K + SG_WAVES - 1) / SG_WAVES) + 7) & ~7;
    kbeg = wave * per;
    kend = min(g.K, kbeg + per);
  } else {
    tile_n = blockIdx.x * SG_WAVES + wave;
  }
  const int i0 = blockIdx.y * 32, j0 = tile_n * 32;
  const int i = i0 + l31, j = j0 + l31;
  const bool iok = i < g.M, jok = j < g.N;
  const float* ap = A + (int64_t)min(i, g.M - 1) * g.sai;
  const float* bp = B + (int64_t)min(j, g.N - 1) * g.sbj;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  // 8 contraction values per iteration = 4 MFMAs; lane half h feeds k = kb + 4h + u to MFMA u.  A K-contiguous
  // operand is fetched as ONE 16-byte load per lane (4-byte row-strided gathers are TA-bound: 32 lines / load).
  const bool avec = g.sak == 1 && !(g.sai & 3) && !((uintptr_t)A & 15);
  const bool bvec = g.sbk == 1 && !(g.sbj & 3) && !((uintptr_t)B & 15);
  for (int kb = kbeg; kb < kend; kb += 8) {  // wave-uniform bounds
    const int k0 = kb + 4 * h;
    f32x4 av = {0.f, 0.f, 0.f, 0.f}, bv = {0.f, 0.f, 0.f, 0.f};
    if (iok) {
      if (avec && k0 + 3 < kend) av = *(const f32x4*)(ap + k0);
      else {
#pragma unroll
        for (int u = 0; u < 4; ++u) if (k0 + u < kend) av[u] = ap[(int64_t)(k0 + u) * g.sak];
      }
    }
    if (jok) {
      if (bvec && k0 + 3 < kend) bv = *(const f32x4*)(bp + k0);
      else {
#pragma unroll
        for (int u = 0; u < 4; ++u) if (k0 + u < kend) bv[u] = bp[(int64_t)(k0 + u) * g.sbk];
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u], acc, 0, 0, 0);
  }
  if (ksplit > 1) {
    if (wave > 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) red[wave - 1][r][lane] = acc[r];
    }
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int w = 0; w < SG_WAVES - 1; ++w)  // fixed order => deterministic
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] += red[w][r][lane];
  }
  if (!jok) return;
  const float bias = g.bias[b] ? g.bias[b][j] : 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {  // lane holds column j, rows (r&3) + 8*(r>>2) + 4*h
    const int row = i0 + (r & 3) + 8 * (r >> 2) + 4 * h;
    if (row >= g.M) continue;
    float* dst = C + (int64_t)row * g.ldc + j;
    const float v = acc[r] + bias;
    *dst = g.beta != 0.f ? *dst * g.beta + v : v;
  }
}

extern "C" int mmt_sgemm_batched(const MmtSgemm* g, void* stream) {
  if (!g || g->batch <= 0 || g->batch > MMT_MAX_EXPERTS || g->M <= 0 || g->N <= 0 || g->K <= 0) return MMT_ERR_ARG;
  for (int b = 0; b < g->batch; ++b)
    if (!g->A[b] || !g->B[b] || !g->C[b]) return MMT_ERR_ARG;
  const int ksplit = g->K >= 128 ? SG_WAVES : 1;
  const int tiles_n = (g->N + 31) / 32;
  hipLaunchKernelGGL(sgemm_batched_kernel,
                     dim3(ksplit > 1 ? tiles_n : (tiles_n + SG_WAVES - 1) / SG_WAVES, (g->M + 31) / 32, g->batch),
                     dim3(64 * SG_WAVES), 0, (hipStream_t)stream, *g, ksplit);
  return (int)hipGetLastError();
}

// ---- BatchNorm statistics over the N rows: x1 [N, M, d] -----------------------------------------------
__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ x1, int N, int M, int d, float eps,
                                                       float momentum, MmtTextHeads h, float* __restrict__ mean_out,
                                                       float* __restrict__ rstd_out) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= M * d) return;
  const int m = idx / d, c = idx % d;
  float s = 0.f;
  for (int n = 0; n < N; ++n) s += x1[((int64_t)n * M + m) * d + c];
  const float mean = s / N;
  float v = 0.f;
  for (int n = 0; n < N; ++n) { const float t = x1[((int64_t)n * M + m) * d + c] - mean; v += t * t; }
  const float var = v / N;
  mean_out[idx] = mean;
  rstd_out[idx] = 1.0f / sqrtf(var + eps);
  if (h.running_mean[m]) {
    h.running_mean[m][c] = (1.f - momentum) * h.running_mean[m][c] + momentum * mean;
    h.running_var[m][c] = (1.f - momentum) * h.running_var[m][c] + momentum * (N > 1 ? v / (N - 1) : var);
  }
}

// eval mode: mean/rstd from the running statistics
__global__ void bn_running_kernel(int M, int d, float eps, MmtTextHeads h, float* __restrict__ mean_out,
                                  float* __restrict__ rstd_out) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= M * d) return;
  const int m = idx / d, c = idx % d;
  mean_out[idx] = h.running_mean[m][c];
  rstd_out[idx] = 1.0f / sqrtf(h.running_var[m][c] + eps);
}

// one wave per (n, m): z = gamma*(x1-mean)*rstd+beta (use_bn) else x1; o = y*sigmoid(z); e = o/max(|o|,1e-12)
// BWD: from de -> dyg (gate-path gradient wrt y) and dz
template <bool BWD>
__global__ __launch_bounds__(256) void gate_norm_kernel(const float* __restrict__ y, const float* __restrict__ x1,
                                                        const float* __restrict__ mean, const float* __restrict__ rstd,
                                                        MmtTextHeads h, int N, int M, int d, int C, int use_bn,
                                                        float* __restrict__ e_out, const float* __restrict__ de,
                                                        float* __restrict__ dyg, float* __restrict__ dz) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nch = (d + 255) >> 8;
  for (int w = blockIdx.x * 4 + wave; w < N * M; w += gridDim.x * 4) {
    const int n = w / M, m = w % M;
    const int64_t src = ((int64_t)n * M + m) * d;
    const int bb = n / C, cc = n % C;
    const int64_t dst = (((int64_t)bb * M + m) * C + cc) * d;  // (B, M, C, d)
    f32x4 o[4], sg[4], yy[4];
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c)
      if (c < nch) {
        const int col = c * 256 + lane * 4;
        if (col < d) {
          yy[c] = *(const f32x4*)(y + src + col);
          f32x4 z = *(const f32x4*)(x1 + src + col);
          if (use_bn) {
            const f32x4 mu = *(const f32x4*)(mean + m * d + col), rs = *(const f32x4*)(rstd + m * d + col);
            const f32x4 ga = *(const f32x4*)(h.bn_gamma[m] + col), be = *(const f32x4*)(h.bn_beta[m] + col);
            z = (z - mu) * rs * ga + be;
          }
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            sg[c][k] = 1.0f / (1.0f + __expf(-z[k]));
            o[c][k] = yy[c][k] * sg[c][k];
            ss += o[c][k] * o[c][k];
          }
        }
      }
    const float nrm = sqrtf(wave_sum(ss));
    const float inv = 1.0f / fmaxf(nrm, 1e-12f);
    if constexpr (!BWD) {
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (c < nch) {
          const int col = c * 256 + lane * 4;
          if (col < d) *(f32x4*)(e_out + dst + col) = o[c] * inv;
        }
    } else {
      f32x4 g[4];
      float dot = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (c < nch) {
          const int col = c * 256 + lane * 4;
          if (col < d) {
            g[c] = *(const f32x4*)(de + dst + col);
            dot += g[c][0] * o[c][0] + g[c][1] * o[c][1] + g[c][2] * o[c][2] + g[c][3] * o[c][3];
          }
        }
      const float proj = nrm > 1e-12f ? wave_sum(dot) * inv * inv : 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (c < nch) {
          const int col = c * 256 + lane * 4;
          if (col < d) {
            f32x4 dy_, dz_;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const float dox = (g[c][k] - o[c][k] * proj) * inv;  // grad wrt o
              dy_[k] = dox * sg[c][k];
              dz_[k] = dox * yy[c][k] * sg[c][k] * (1.0f - sg[c][k]);
            }
            *(f32x4*)(dyg + src + col) = dy_;
            *(f32x4*)(dz + src + col) = dz_;
          }
        }
    }
  }
}

// BatchNorm backward per (m, column): dz [N,M,d] -> dx1 (in place over dz), dgamma, dbeta, db2 = colsum(dx1)
__global__ __launch_bounds__(256) void bn_bwd_kernel(const float* __restrict__ x1, float* __restrict__ dz,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     MmtTextHeads h, int N, int M, int d, int use_bn, int training) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= M * d) return;
  const int m = idx / d, c = idx % d;
  if (!use_bn) {  // dx1 = dz; only the bias gradient of cg.fc is needed
    float s = 0.f;
    for (int n = 0; n < N; ++n) s += dz[((int64_t)n * M + m) * d + c];
    if (h.g_b2[m]) h.g_b2[m][c] = s;
    return;
  }
  const float mu = mean[idx], rs = rstd[idx], ga = h.bn_gamma[m][c];
  float s1 = 0.f, s2 = 0.f;
  for (int n = 0; n < N; ++n) {
    const int64_t p = ((int64_t)n * M + m) * d + c;
    const float xh = (x1[p] - mu) * rs, g = dz[p];
    s1 += g;
    s2 += g * xh;
  }
  if (h.g_bn_gamma[m]) h.g_bn_gamma[m][c] = s2;
  if (h.g_bn_beta[m]) h.g_bn_beta[m][c] = s1;
  float sb = 0.f;
  for (int n = 0; n < N; ++n) {
    const int64_t p = ((int64_t)n * M + m) * d + c;
    float dx;
    if (training) {
      const float xh = (x1[p] - mu) * rs;
      dx = ga * rs * (dz[p] - s1 / N - xh * s2 / N);
    } else {
      dx = ga * rs * dz[p];
    }
    dz[p] = dx;
    sb += dx;
  }
  if (h.g_b2[m]) h.g_b2[m][c] = sb;
}

// column sums of dy [N, M*d] -> g_b1[m][c]
__global__ void colsum_f32_kernel(const float* __restrict__ dy, int N, int M, int d, MmtTextHeads h) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= M * d) return;
  const int m = idx / d, c = idx % d;
  if (!h.g_b1[m]) return;
  float s = 0.f;
  for (int n = 0; n < N; ++n) s += dy[(int64_t)n * M * d + idx];
  h.g_b1[m][c] = s;
}

// MoE: one block per row n.  fwd: tw[n][m] = softmax_m(text[n].w_m + b_m).
__global__ __launch_bounds__(256) void moe_fwd_kernel(const float* __restrict__ text, int K, int M, MmtTextHeads h,
                                                      float* __restrict__ tw) {
  __shared__ float logit[MMT_MAX_EXPERTS];
  const int n = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int m = wave; m < M; m += 4) {
    float s = 0.f;
    for (int k = lane; k < K; k += 64) s += text[(int64_t)n * K + k] * h.moe_w[m][k];
    s = wave_sum(s);
    if (lane == 0) logit[m] = s + h.moe_b[m][0];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float mx = -INFINITY, sum = 0.f;
    for (int m = 0; m < M; ++m) mx = fmaxf(mx, logit[m]);
    for (int m = 0; m < M; ++m) sum += __expf(logit[m] - mx);
    float l1 = 0.f;
    for (int m = 0; m < M; ++m) { logit[m] = __expf(logit[m] - mx) / sum; l1 += fabsf(logit[m]); }
    for (int m = 0; m < M; ++m) tw[(int64_t)n * M + m] = logit[m] / fmaxf(l1, 1e-12f);  // F.normalize(p=1), model.py:618
  }
}

// bwd part 1 (block per row n): dlogit[n][m] = tw (dtw - sum_j tw_j dtw_j); dtext[n] += sum_m dlogit[n][m] w_m
__global__ __launch_bounds__(256) void moe_bwd_row_kernel(const float* __restrict__ tw, const float* __restrict__ dtw, int K,
                                                          int M, MmtTextHeads h, float* __restrict__ dlogit,
                                                          float* __restrict__ dtext, int accumulate) {
  __shared__ float dl[MMT_MAX_EXPERTS];
  const int n = blockIdx.x;
  if (threadIdx.x == 0) {
    float dot = 0.f;
    for (int m = 0; m < M; ++m) dot += tw[(int64_t)n * M + m] * dtw[(int64_t)n * M + m];
    for (int m = 0; m < M; ++m) {
      dl[m] = tw[(int64_t)n * M + m] * (dtw[(int64_t)n * M + m] - dot);
      dlogit[(int64_t)n * M + m] = dl[m];
    }
  }
  __syncthreads();
  if (dtext)
    for (int k = threadIdx.x; k < K; k += 256) {
      float s = 0.f;
      for (int m = 0; m < M; ++m) s += dl[m] * h.moe_w[m][k];
      dtext[(int64_t)n * K + k] = accumulate ? dtext[(int64_t)n * K + k] + s : s;
    }
}
// bwd part 2 (block per expert m): g_moe_w[m][k] = sum_n dlogit[n][m] text[n][k]; g_moe_b[m] = sum_n dlogit[n][m]
__global__ __launch_bounds__(256) void moe_bwd_w_kernel(const float* __restrict__ text, const float* __restrict__ dlogit,
                                                        int N, int K, int M, MmtTextHeads h) {
  const int m = blockIdx.x;
  const int k = blockIdx.y * 256 + threadIdx.x;
  if (k < K) {
    float s = 0.f;
    for (int n = 0; n < N; ++n) s += dlogit[(int64_t)n * M + m] * text[(int64_t)n * K + k];
    if (h.g_moe_w[m]) h.g_moe_w[m][k] = s;
  }
  if (blockIdx.y == 0 && threadIdx.x == 0 && h.g_moe_b[m]) {
    float s = 0.f;
    for (int n = 0; n < N; ++n) s += dlogit[(int64_t)n * M + m];
    h.g_moe_b[m][0] = s;
  }
}

// ------------------------------------------------------------------------------------------------
static int check_heads(const MmtTextHeads* h, const float* text, int N, int M, int d, int K) {
  if (!h || !text || N <= 0 || M <= 0 || M > MMT_MAX_EXPERTS || d % 4 || d > 1024 || K <= 0) return MMT_ERR_ARG;
  for (int m = 0; m < M; ++m)
    if (!h->w1[m] || !h->b1[m] || !h->w2[m] || !h->b2[m]) return MMT_ERR_ARG;
  return 0;
}

#define TRY(x) do { int rc_ = (x); if (rc_) return rc_; } while (0)

// ws layout (floats): y [N*M*d] | x1 [N*M*d] | mean [M*d] | rstd [M*d] | dyg [N*M*d] | dz [N*M*d] | dlogit [N*M] |
//                     (small-batch path, texthead2.hip) sigmoid gate [N*M*d] | un-normalised output [N*M*d] | row partials
extern "C" int64_t mmt_text_heads_workspace_floats(int N, int M, int d) {
  return 6LL * N * M * d + 2LL * M * d + (((int64_t)N * M + 63) & ~63LL) + (int64_t)N * M * ((d + 31) / 32) + 64;
}

// texthead2.hip: N <= 32 rows in 3 + 3 launches
int mmt_text_heads_fwd_small(const MmtTextHeads* h, const float* text, const float* text_moe, int N, int C, int M, int d,
                             int K, int use_bn, int training, float* ws, float* text_embds, float* text_weights,
                             const MmtTextHeadsOpts* opts, hipStream_t s);
int mmt_text_heads_bwd_small(const MmtTextHeads* h, const float* text, const float* text_moe, int N, int C, int M, int d,
                             int K, int use_bn, int training, float* ws, const float* dtext_embds, const float* text_weights,
                             const float* dtext_weights, float* dtext_moe, const MmtTextHeadsOpts* opts, hipStream_t s);
static bool small_path(int N, int M, int d, int K) {
  return mmt_text_heads_fast(N, M, d, K);  // (honours the lab switch MMT_TEXT_HEADS_V1: the one-kernel-per-op path)
}
static bool fused_dropout(const MmtTextHeadsOpts* o, const float* text_moe) { return o && o->moe_drop_thr16 && !text_moe; }

extern "C" int mmt_text_heads_fwd(const MmtTextHeads* h, const float* text, const float* text_moe, int N, int C, int M,
                                  int d, int K, int use_bn, int training, float* ws, float* text_embds,
                                  float* text_weights, const MmtTextHeadsOpts* opts, void* stream) {
  TRY(check_heads(h, text, N, M, d, K));
  if (!ws || !text_embds || C <= 0 || N % C) return MMT_ERR_ARG;
  if (text_weights)
    for (int m = 0; m < M; ++m)
      if (!h->moe_w[m] || !h->moe_b[m]) return MMT_ERR_ARG;
  if (small_path(N, M, d, K))
    return mmt_text_heads_fwd_small(h, text, text_moe, N, C, M, d, K, use_bn, training, ws, text_embds, text_weights, opts,
                                    (hipStream_t)stream);
  if (fused_dropout(opts, text_moe) || (opts && (opts->num_batches_tracked || opts->video_front))) return MMT_ERR_ARG;  // small-batch path only
  const int64_t nmd = (int64_t)N * M * d;
  float *y = ws, *x1 = ws + nmd, *mean = ws + 2 * nmd, *rstd = mean + (int64_t)M * d;
  hipStream_t s = (hipStream_t)stream;
  MmtSgemm g = {};
  g.batch = M; g.M = N; g.N = d; g.K = K; g.sai = K; g.sak = 1; g.sbj = K; g.sbk = 1; g.ldc = (int64_t)M * d;
  for (int m = 0; m < M; ++m) { g.A[m] = text; g.B[m] = h->w1[m]; g.C[m] = y + (int64_t)m * d; g.bias[m] = h->b1[m]; }
  TRY(mmt_sgemm_batched(&g, stream));                       // y = fc(text)                 model.py:698
  g.K = d; g.sai = (int64_t)M * d; g.sbj = d;
  for (int m = 0; m < M; ++m) { g.A[m] = y + (int64_t)m * d; g.B[m] = h->w2[m]; g.C[m] = x1 + (int64_t)m * d; g.bias[m] = h->b2[m]; }
  TRY(mmt_sgemm_batched(&g, stream));                       // x1 = cg.fc(y)                model.py:745
  if (use_bn) {
    const int blocks = (M * d + 255) / 256;
    if (training)
      hipLaunchKernelGGL(bn_stats_kernel, dim3(blocks), dim3(256), 0, s, x1, N, M, d, 1e-5f, 0.1f, *h, mean, rstd);
    else
      hipLaunchKernelGGL(bn_running_kernel, dim3(blocks), dim3(256), 0, s, M, d, 1e-5f, *h, mean, rstd);
  }
  int gb = (N * M + 3) / 4;
  if (gb > 2048) gb = 2048;
  hipLaunchKernelGGL(gate_norm_kernel<false>, dim3(gb), dim3(256), 0, s, y, x1, mean, rstd, *h, N, M, d, C, use_bn,
                     text_embds, nullptr, nullptr, nullptr);
  if (text_weights) {
    for (int m = 0; m < M; ++m)
      if (!h->moe_w[m] || !h->moe_b[m]) return MMT_ERR_ARG;
    hipLaunchKernelGGL(moe_fwd_kernel, dim3(N), dim3(256), 0, s, text_moe ? text_moe : text, K, M, *h, text_weights);
  }
  return (int)hipGetLastError();
}

// w1_all: the M fc.weight matrices contiguous as [M*d, K] (flat layout) for the single text-gradient GEMM.
extern "C" int mmt_text_heads_bwd(const MmtTextHeads* h, const float* text, const float* text_moe, const float* w1_all,
                                  int N, int C, int M, int d, int K, int use_bn, int training, float* ws,
                                  const float* dtext_embds, const float* text_weights, const float* dtext_weights,
                                  float* dtext, float* dtext_moe, const MmtTextHeadsOpts* opts, void* stream) {
  TRY(check_heads(h, text, N, M, d, K));
  if (!ws || !dtext_embds || C <= 0 || N % C) return MMT_ERR_ARG;
  const int64_t nmd = (int64_t)N * M * d;
  float *y = ws, *x1 = ws + nmd, *mean = ws + 2 * nmd, *rstd = mean + (int64_t)M * d;
  float *dyg = rstd + (int64_t)M * d, *dz = dyg + nmd, *dlogit = dz + nmd;
  hipStream_t s = (hipStream_t)stream;
  if (small_path(N, M, d, K)) {
    // the masked MoE-input gradient needs its own buffer when the dropout is applied on the fly
    if (fused_dropout(opts, text_moe) && dtext && !dtext_moe) return MMT_ERR_ARG;
    if (dtext && !w1_all) return MMT_ERR_ARG;
    TRY(mmt_text_heads_bwd_small(h, text, text_moe, N, C, M, d, K, use_bn, training, ws, dtext_embds, text_weights,
                                 dtext_weights, dtext_moe, opts, s));
    if (dtext) {  // dtext = dy_all [N, M*d] . W1_all [M*d, K]   (dyg holds dy after th_bwd3)
      MmtSgemm t = {};
      t.batch = 1; t.M = N; t.N = K; t.K = M * d; t.sai = (int64_t)M * d; t.sak = 1; t.sbj = 1; t.sbk = K; t.ldc = K;
      t.A[0] = dyg; t.B[0] = w1_all; t.C[0] = dtext;
      TRY(mmt_sgemm_batched(&t, stream));
      if (text_weights && dtext_weights && !dtext_moe)  // no separate MoE input: its gradient joins dtext
        hipLaunchKernelGGL(moe_bwd_row_kernel, dim3(N), dim3(256), 0, s, text_weights, dtext_weights, K, M, *h, dlogit, dtext, 1);
    }
    return (int)hipGetLastError();
  }
  if (fused_dropout(opts, text_moe)) return MMT_ERR_ARG;  // small-batch path only
  int gb = (N * M + 3) / 4;
  if (gb > 2048) gb = 2048;
  hipLaunchKernelGGL(gate_norm_kernel<true>, dim3(gb), dim3(256), 0, s, y, x1, mean, rstd, *h, N, M, d, C, use_bn, nullptr,
                     dtext_embds, dyg, dz);
  hipLaunchKernelGGL(bn_bwd_kernel, dim3((M * d + 255) / 256), dim3(256), 0, s, x1, dz, mean, rstd, *h, N, M, d, use_bn,
                     training);  // dz now holds dx1
  MmtSgemm g = {};
  // dy = dyg + dx1 . W2
  g.batch = M; g.M = N; g.N = d; g.K = d; g.sai = (int64_t)M * d; g.sak = 1; g.sbj = 1; g.sbk = d;
  g.ldc = (int64_t)M * d; g.beta = 1.f;
  for (int m = 0; m < M; ++m) { g.A[m] = dz + (int64_t)m * d; g.B[m] = h->w2[m]; g.C[m] = dyg + (int64_t)m * d; g.bias[m] = nullptr; }
  TRY(mmt_sgemm_batched(&g, stream));
  // g_w2[m] = dx1[m]^T . y[m]
  g.M = d; g.N = d; g.K = N; g.sai = 1; g.sak = (int64_t)M * d; g.sbj = 1; g.sbk = (int64_t)M * d; g.ldc = d; g.beta = 0.f;
  bool any = false;
  for (int m = 0; m < M; ++m) { g.A[m] = dz + (int64_t)m * d; g.B[m] = y + (int64_t)m * d; g.C[m] = h->g_w2[m]; any |= h->g_w2[m] != nullptr; }
  if (any) TRY(mmt_sgemm_batched(&g, stream));
  // g_w1[m] = dy[m]^T . text
  g.M = d; g.N = K; g.K = N; g.sai = 1; g.sak = (int64_t)M * d; g.sbj = 1; g.sbk = K; g.ldc = K;
  any = false;
  for (int m = 0; m < M; ++m) { g.A[m] = dyg + (int64_t)m * d; g.B[m] = text; g.C[m] = h->g_w1[m]; any |= h->g_w1[m] != nullptr; }
  if (any) TRY(mmt_sgemm_batched(&g, stream));
  hipLaunchKernelGGL(colsum_f32_kernel, dim3((M * d + 255) / 256), dim3(256), 0, s, dyg, N, M, d, *h);
  if (dtext) {  // dtext = dy_all [N, M*d] . W1_all [M*d, K]
    if (!w1_all) return MMT_ERR_ARG;
    MmtSgemm t = {};
    t.batch = 1; t.M = N; t.N = K; t.K = M * d; t.sai = (int64_t)M * d; t.sak = 1; t.sbj = 1; t.sbk = K; t.ldc = K;
    t.A[0] = dyg; t.B[0] = w1_all; t.C[0] = dtext;
    TRY(mmt_sgemm_batched(&t, stream));
  }
  if (text_weights && dtext_weights) {
    // the MoE branch may read a dropped-out copy of text (moe_txt_dropout, model.py:274): its input gradient then
    // goes to dtext_moe (written); otherwise it is accumulated into dtext.
    hipLaunchKernelGGL(moe_bwd_row_kernel, dim3(N), dim3(256), 0, s, text_weights, dtext_weights, K, M, *h, dlogit,
                       dtext_moe ? dtext_moe : dtext, dtext_moe ? 0 : 1);
    hipLaunchKernelGGL(moe_bwd_w_kernel, dim3(M, (K + 255) / 256), dim3(256), 0, s, text_moe ? text_moe : text, dlogit, N, K, M, *h);
  }
  return (int)hipGetLastError();
}
