// Text heads of CENet for SMALL batches (N = B*C <= 32 caption rows: every training configuration of the reference),
// gfx950, fp32.  Same arithmetic as texthead.hip (model/model.py:683-702 GatedEmbeddingUnit, :736-750 ContextGating +
// BatchNorm1d, :262-283,618 text MoE), re-cut so that a training step spends 3 launches forward and 3 backward on them
// instead of 6 + 8: under HIP-graph replay every dependent launch costs ~5 us however little it does, and the old
// kernels were serial latency chains on a handful of blocks (bn_stats: 14 blocks walking 32 strided rows twice).
//
//   th_fwd1 : y = fc(text)                         one block per (expert, 32 output columns), 16 waves split K;
//             + MoE weights of row n (extra blocks), moe_txt_dropout applied on the fly (no dropped copy of text)
//   th_fwd2 : x1 = cg.fc(y) + BatchNorm statistics (the 32 rows of a column live in ONE block) + running stats
//             + gate o = y * sigmoid(BN(x1)) + per-row partial sums of o^2
//   th_fwd3 : e = o / max(|o|, 1e-12) in (B, M, C, d) layout
//   th_bwd1 : normalise + gate backward per row (-> dyg, dz); MoE softmax backward (-> dlogit, masked d text_moe)
//   th_bwd2 : BatchNorm backward per column block (-> dx1, g_bn_*, g_b2) + g_w2 = dx1^T . y
//   th_bwd3 : dy = dyg + dx1 . W2 (16 waves split the contraction), g_b1, g_w1 = dy^T . text; MoE weight gradients
// The GEMMs stay on the exact-fp32 matrix cores (v_mfma_f32_32x32x2_f32 = an fp32 fma chain).  A wave's operands for
// its whole K slice are loaded before the first MFMA, so a block's critical path is ONE memory round trip.
#include <stdlib.h>
#include "mmt_common.h"
#include "../../include/mmt_hip.h"
#include "video_front.h"

typedef __attribute__((ext_vector_type(16))) float f32x16;

#define TH_W 16               // waves per block
#define TH_T (64 * TH_W)
#define TH_TMAX 8             // f32x4 operand groups per lane and K slice (K <= 16 * 64)

struct ThArgs {
  MmtTextHeads h;
  const float* text; const float* text_moe;
  float* ws;
  float* text_embds; float* text_weights;
  const float* de; const float* tw; const float* dtw;
  float* dtext_moe;
  int N, C, M, d, K, use_bn, training;
  uint32_t drop_key, thr16; float drop_scale;
  const uint32_t* seed_dev; uint32_t* key_dev; long long* nbt;
};

struct ThWs { float *y, *x1, *mean, *rstd, *dyg, *dz, *dlogit, *sg, *o, *part; };
__host__ __device__ inline ThWs th_layout(float* ws, int N, int M, int d) {
  const int64_t nmd = (int64_t)N * M * d, md = (int64_t)M * d;
  ThWs w;
  w.y = ws; w.x1 = ws + nmd; w.mean = ws + 2 * nmd; w.rstd = w.mean + md; w.dyg = w.rstd + md; w.dz = w.dyg + nmd;
  w.dlogit = w.dz + nmd; w.sg = w.dlogit + (((int64_t)N * M + 63) & ~63LL); w.o = w.sg + nmd; w.part = w.o + nmd;
  return w;
}

// LDS of the GEMM blocks: split-K partials [16 waves][16 regs][64 lanes] + three 32x33 tiles + column vectors
struct ThSmem {
  float red[TH_W][16][64];
  float tile[3][32][33];
  float colv[4][32];
};

// partial 32x32 tile of this wave: rows = operand A rows (i = lane & 31), cols = operand B rows (j = lane & 31), both
// K-contiguous; k in [kbeg, kend).  All loads first, then the MFMA chain.
__device__ __forceinline__ f32x16 mm_nt_slice(const float* __restrict__ ap, bool iok, const float* __restrict__ bp, bool jok,
                                              int kbeg, int kend, int h) {
  f32x4 av[TH_TMAX], bv[TH_TMAX];
#pragma unroll
  for (int t = 0; t < TH_TMAX; ++t) {
    const int k0 = kbeg + 8 * t + 4 * h;
    av[t] = bv[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (k0 + 3 < kend) {
      if (iok) av[t] = *(const f32x4*)(ap + k0);
      if (jok) bv[t] = *(const f32x4*)(bp + k0);
    } else if (k0 < kend) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (k0 + u < kend) {
          if (iok) av[t][u] = ap[k0 + u];
          if (jok) bv[t][u] = bp[k0 + u];
        }
    }
  }
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
  for (int t = 0; t < TH_TMAX; ++t) {
    if (kbeg + 8 * t < kend) {  // wave-uniform
#pragma unroll
      for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t][u], bv[t][u], acc, 0, 0, 0);
    }
  }
  return acc;
}

// Sum the 16 waves' partial tiles: wave w reduces accumulator register w of every wave, i.e. it ends up with the tile
// rows n = (w & 3) + 8 * (w >> 2) + 4 * (lane >> 5) at column lane & 31.  Fixed order => deterministic.
__device__ __forceinline__ float splitk_reduce(ThSmem& sm, const f32x16& acc, int wave, int lane) {
#pragma unroll
  for (int r = 0; r < 16; ++r) sm.red[wave][r][lane] = acc[r];
  __syncthreads();
  float v = 0.f;
#pragma unroll
  for (int w = 0; w < TH_W; ++w) v += sm.red[w][wave][lane];
  return v;
}
__device__ __forceinline__ int tile_row(int wave, int lane) { return (wave & 3) + 8 * (wave >> 2) + 4 * (lane >> 5); }

// Column sums of a 32x33 LDS tile over its first N rows, without a serial loop: thread t takes element
// (row t & 31, column t >> 5), so a column's 32 rows sit in the 32 lanes of one half-wave (5 xor-shuffles).  Every lane of
// the half-wave ends up with the sum of column t >> 5.
__device__ __forceinline__ float half_wave_sum(float v) {
#pragma unroll
  for (int s = 16; s > 0; s >>= 1) v += __shfl_xor(v, s, 64);
  return v;
}

__device__ __forceinline__ int k_slice(int K) { return (((K + TH_W - 1) / TH_W) + 7) & ~7; }

// element e of the text row as the MoE branch sees it (model.py:274: dropout in front of the MoE logits)
__device__ __forceinline__ f32x4 moe_input4(const ThArgs& a, unsigned key, int n, int k4) {
  if (a.text_moe) return *(const f32x4*)(a.text_moe + (int64_t)n * a.K + k4);
  f32x4 v = *(const f32x4*)(a.text + (int64_t)n * a.K + k4);
  if (a.thr16) {
    bool kp[4];
    keep4(key, (unsigned long long)n * (unsigned)a.K + (unsigned)k4, a.thr16, kp);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = kp[e] ? v[e] * a.drop_scale : 0.f;
  }
  return v;
}

// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void th_fwd1_block(const ThArgs& a, const int bid, ThSmem& sm) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
  const int ct = a.d / 32, gemm_blocks = a.M * ct;
  const ThWs w = th_layout(a.ws, a.N, a.M, a.d);
  if (bid < gemm_blocks) {
    const int m = bid / ct, j0 = (bid % ct) * 32;
    const int per = k_slice(a.K), kbeg = wave * per, kend = min(a.K, kbeg + per);
    const f32x16 acc = mm_nt_slice(a.text + (int64_t)min(l31, a.N - 1) * a.K, l31 < a.N,
                                   a.h.w1[m] + (int64_t)(j0 + l31) * a.K, true, kbeg, kend, h);
    const float v = splitk_reduce(sm, acc, wave, lane);
    const int n = tile_row(wave, lane);
    if (n < a.N) w.y[((int64_t)n * a.M + m) * a.d + j0 + l31] = v + a.h.b1[m][j0 + l31];
    return;
  }
  // ---- MoE weights of row n: logits (one wave per expert), softmax, L1 normalise (model.py:262-283, 618) ----
  const int n = bid - gemm_blocks;
  unsigned key = 0;
  if (!a.text_moe && a.thr16) {
    key = eff_key(a.drop_key, a.seed_dev);
    if (n == 0 && tid == 0 && a.key_dev) *a.key_dev = key;
  }
  float* logit = sm.colv[0];
  for (int m = wave; m < a.M; m += TH_W) {
    float s = 0.f;
    for (int k4 = lane * 4; k4 < a.K; k4 += 256) {
      const f32x4 x = moe_input4(a, key, n, k4), ww = *(const f32x4*)(a.h.moe_w[m] + k4);
      s += x[0] * ww[0] + x[1] * ww[1] + x[2] * ww[2] + x[3] * ww[3];
    }
    s = wave_sum(s);
    if (lane == 0) logit[m] = s + a.h.moe_b[m][0];
  }
  __syncthreads();
  if (tid == 0) {
    float mx = -INFINITY, sum = 0.f, l1 = 0.f;
    for (int m = 0; m < a.M; ++m) mx = fmaxf(mx, logit[m]);
    for (int m = 0; m < a.M; ++m) sum += __expf(logit[m] - mx);
    for (int m = 0; m < a.M; ++m) { logit[m] = __expf(logit[m] - mx) / sum; l1 += fabsf(logit[m]); }
    for (int m = 0; m < a.M; ++m) a.text_weights[(int64_t)n * a.M + m] = logit[m] / fmaxf(l1, 1e-12f);
  }
}

__global__ __launch_bounds__(TH_T) void th_fwd1_kernel(ThArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char th_smem_raw[];
  th_fwd1_block(a, (int)blockIdx.x, *(ThSmem*)th_smem_raw);
}

struct CastSplit { int32_t begin[MMT_MAX_EXPERTS + 1]; };  // cast blocks of expert e: [begin[e], begin[e + 1]), ~ its width
static_assert(sizeof(ThArgs) + sizeof(VideoPlanArgs) + 64 <= 4096 && sizeof(ThArgs) + sizeof(VideoCastArgs) + sizeof(CastSplit) + 64 <= 4096,
              "kernel arguments: 4 KiB");
// The same launch carrying the video side's token plan as B extra blocks (4 of their 16 waves work, video_front.h).
__global__ __launch_bounds__(TH_T) void th_fwd1_plan_kernel(ThArgs a, VideoPlanArgs p, int th_blocks) {
  extern __shared__ __attribute__((aligned(16))) unsigned char th_smem_raw[];
  if ((int)blockIdx.x < th_blocks) {
    th_fwd1_block(a, (int)blockIdx.x, *(ThSmem*)th_smem_raw);
    return;
  }
  if (threadIdx.x >= 256) return;
  video_plan_block(p, (int)blockIdx.x - th_blocks, (int)threadIdx.x, (float*)th_smem_raw);
}

__device__ __forceinline__ void th_fwd2_block(const ThArgs& a, const int bid, ThSmem& sm) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
  const int ct = a.d / 32;
  const int m = bid / ct, jt = bid % ct, j0 = jt * 32;
  const ThWs w = th_layout(a.ws, a.N, a.M, a.d);
  if (bid == 0 && tid < a.M && a.nbt && a.use_bn && a.training) a.nbt[tid] += 1;  // num_batches_tracked
  const int per = k_slice(a.d), kbeg = wave * per, kend = min(a.d, kbeg + per);
  const f32x16 acc = mm_nt_slice(w.y + ((int64_t)min(l31, a.N - 1) * a.M + m) * a.d, l31 < a.N,
                                 a.h.w2[m] + (int64_t)(j0 + l31) * a.d, true, kbeg, kend, h);
  float v = splitk_reduce(sm, acc, wave, lane);
  {
    const int n = tile_row(wave, lane);
    v += a.h.b2[m][j0 + l31];
    sm.tile[0][n][l31] = n < a.N ? v : 0.f;
    if (n < a.N) w.x1[((int64_t)n * a.M + m) * a.d + j0 + l31] = v;
  }
  __syncthreads();
  if (a.use_bn) {  // BatchNorm1d statistics of column j0 + c over the N rows (model.py:744-747): thread (row r, column c)
    const int r = tid & 31, c = tid >> 5, col = j0 + c;
    float mean, rstd;
    if (a.training) {
      const float x = r < a.N ? sm.tile[0][r][c] : 0.f;
      mean = half_wave_sum(x) / a.N;
      const float dev = r < a.N ? x - mean : 0.f;
      const float q = half_wave_sum(dev * dev);
      const float var = q / a.N;
      rstd = 1.0f / sqrtf(var + 1e-5f);
      if (r == 0 && a.h.running_mean[m]) {
        a.h.running_mean[m][col] = 0.9f * a.h.running_mean[m][col] + 0.1f * mean;
        a.h.running_var[m][col] = 0.9f * a.h.running_var[m][col] + 0.1f * (a.N > 1 ? q / (a.N - 1) : var);
      }
    } else {
      mean = a.h.running_mean[m][col];
      rstd = 1.0f / sqrtf(a.h.running_var[m][col] + 1e-5f);
    }
    if (r == 0) {
      w.mean[(int64_t)m * a.d + col] = mean;
      w.rstd[(int64_t)m * a.d + col] = rstd;
      sm.colv[0][c] = mean;
      sm.colv[1][c] = rstd;
    }
  }
  __syncthreads();
  // gate: thread (n = tid / 32, j = tid % 32)
  const int n = tid >> 5, j = tid & 31, col = j0 + j;
  float o = 0.f;
  if (n < a.N) {
    float z = sm.tile[0][n][j];
    if (a.use_bn) z = (z - sm.colv[0][j]) * sm.colv[1][j] * a.h.bn_gamma[m][col] + a.h.bn_beta[m][col];
    const float sg = 1.0f / (1.0f + __expf(-z));
    const int64_t p = ((int64_t)n * a.M + m) * a.d + col;
    o = w.y[p] * sg;
    w.sg[p] = sg;
    w.o[p] = o;
  }
  float sq = o * o;
#pragma unroll
  for (int s = 16; s > 0; s >>= 1) sq += __shfl_xor(sq, s, 64);
  if (j == 0 && n < a.N) w.part[((int64_t)n * a.M + m) * ct + jt] = sq;
}

__global__ __launch_bounds__(TH_T) void th_fwd2_kernel(ThArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char th_smem_raw[];
  th_fwd2_block(a, (int)blockIdx.x, *(ThSmem*)th_smem_raw);
}

// The same launch carrying the video side's feature cast (cast_bx blocks per expert) and the per-step bump of the
// dropout seed (the text heads read the seed in their FIRST launch, the encoder after this one).
__global__ __launch_bounds__(TH_T) void th_fwd2_cast_kernel(ThArgs a, VideoCastArgs c, int th_blocks, CastSplit cs, int M,
                                                            uint32_t* seed_bump) {
  extern __shared__ __attribute__((aligned(16))) unsigned char th_smem_raw[];
  if (blockIdx.x == 0 && threadIdx.x == 0 && seed_bump) *seed_bump += 1u;
  if ((int)blockIdx.x < th_blocks) {
    th_fwd2_block(a, (int)blockIdx.x, *(ThSmem*)th_smem_raw);
    return;
  }
  const int r = (int)blockIdx.x - th_blocks;
  int ex = 0;
#pragma unroll 1
  for (int q = 1; q < M; ++q)
    if (r >= cs.begin[q]) ex = q;
  video_cast_block(c, ex, r - cs.begin[ex], cs.begin[ex + 1] - cs.begin[ex], (int)threadIdx.x, TH_T, 0, 2);  // first half
}

// one wave per (n, m): e = o / max(|o|, 1e-12) -> (B, M, C, d)
__device__ __forceinline__ void th_fwd3_block(const ThArgs& a, const int bid) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ct = a.d / 32;
  const ThWs w = th_layout(a.ws, a.N, a.M, a.d);
  const int r = bid * 4 + wave;
  if (r >= a.N * a.M) return;
  const int n = r / a.M, m = r % a.M;
  const float ss = wave_sum(lane < ct ? w.part[(int64_t)r * ct + lane] : 0.f);
  const float inv = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
  const int64_t src = (int64_t)r * a.d, dst = (((int64_t)(n / a.C) * a.M + m) * a.C + n % a.C) * a.d;
  for (int c = lane * 4; c < a.d; c += 256) *(f32x4*)(a.text_embds + dst + c) = *(const f32x4*)(w.o + src + c) * inv;
}

__global__ __launch_bounds__(256) void th_fwd3_kernel(ThArgs a) { th_fwd3_block(a, (int)blockIdx.x); }

// The third launch carrying the second half of the feature cast: 256-thread blocks without LDS, i.e. as many per CU as the
// stand-alone cast had (the blocks riding in the second launch inherit its 78 KiB of LDS: two per CU).
__global__ __launch_bounds__(256) void th_fwd3_cast_kernel(ThArgs a, VideoCastArgs c, int th_blocks, CastSplit cs, int M) {
  if ((int)blockIdx.x < th_blocks) {
    th_fwd3_block(a, (int)blockIdx.x);
    return;
  }
  const int r = (int)blockIdx.x - th_blocks;
  int ex = 0;
#pragma unroll 1
  for (int q = 1; q < M; ++q)
    if (r >= cs.begin[q]) ex = q;
  video_cast_block(c, ex, r - cs.begin[ex], cs.begin[ex + 1] - cs.begin[ex], (int)threadIdx.x, 256, 1, 2);  // second half
}

// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void th_bwd1_kernel(ThArgs a) {
  __shared__ float dl[MMT_MAX_EXPERTS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const ThWs w = th_layout(a.ws, a.N, a.M, a.d);
  const int row_blocks = (a.N * a.M + 3) / 4;
  if ((int)blockIdx.x < row_blocks) {
    const int r = blockIdx.x * 4 + wave;
    if (r >= a.N * a.M) return;
    const int n = r / a.M, m = r % a.M;
    const int64_t src = (int64_t)r * a.d, dst = (((int64_t)(n / a.C) * a.M + m) * a.C + n % a.C) * a.d;
    f32x4 o[4], g[4];
    float ss = 0.f, dot = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int col = c * 256 + lane * 4;
      o[c] = g[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (col < a.d) {
        o[c] = *(const f32x4*)(w.o + src + col);
        g[c] = *(const f32x4*)(a.de + dst + col);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) { ss += o[c][k] * o[c][k]; dot += g[c][k] * o[c][k]; }
    }
    const float nrm = sqrtf(wave_sum(ss));
    const float inv = 1.0f / fmaxf(nrm, 1e-12f);
    const float proj = nrm > 1e-12f ? wave_sum(dot) * inv * inv : 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int col = c * 256 + lane * 4;
      if (col < a.d) {
        const f32x4 sg = *(const f32x4*)(w.sg + src + col), yy = *(const f32x4*)(w.y + src + col);
        f32x4 dy_, dz_;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float dox = (g[c][k] - o[c][k] * proj) * inv;
          dy_[k] = dox * sg[k];
          dz_[k] = dox * yy[k] * sg[k] * (1.0f - sg[k]);
        }
        *(f32x4*)(w.dyg + src + col) = dy_;
        *(f32x4*)(w.dz + src + col) = dz_;
      }
    }
    return;
  }
  // ---- MoE softmax backward of row n: dlogit = tw (dtw - <tw, dtw>); d text_moe = mask * sum_m dlogit_m w_m ----
  const int n = blockIdx.x - row_blocks;
  if (wave == 0) {
    const float t = lane < a.M ? a.tw[(int64_t)n * a.M + lane] : 0.f, dv = lane < a.M ? a.dtw[(int64_t)n * a.M + lane] : 0.f;
    const float dot = wave_sum(t * dv);
    if (lane < a.M) {
      dl[lane] = t * (dv - dot);
      w.dlogit[(int64_t)n * a.M + lane] = dl[lane];
    }
  }
  __syncthreads();
  if (!a.dtext_moe) return;
  const unsigned key = (!a.text_moe && a.thr16 && a.key_dev) ? *a.key_dev : 0u;
  for (int k4 = tid * 4; k4 < a.K; k4 += 1024) {
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int m = 0; m < a.M; ++m) s += *(const f32x4*)(a.h.moe_w[m] + k4) * dl[m];
    if (!a.text_moe && a.thr16) {
      bool kp[4];
      keep4(key, (unsigned long long)n * (unsigned)a.K + (unsigned)k4, a.thr16, kp);
#pragma unroll
      for (int e = 0; e < 4; ++e) s[e] = kp[e] ? s[e] * a.drop_scale : 0.f;
    }
    *(f32x4*)(a.dtext_moe + (int64_t)n * a.K + k4) = s;
  }
}

// weight-gradient tile: C[i][c] = sum_{n < N} T[n][i] * G[n * ldg + c0 + (lane & 31)], T = a 32x33 LDS tile
__device__ __forceinline__ f32x16 mm_tn_tile(const float (*T)[33], const float* __restrict__ G, int64_t ldg, int N, int l31,
                                             int h) {
  float b[16];
#pragma unroll
  for (int kk = 0; kk < 16; ++kk) {
    const int k = 2 * kk + h;
    b[kk] = k < N ? G[(int64_t)k * ldg + l31] : 0.f;
  }
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
  for (int kk = 0; kk < 16; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(T[2 * kk + h][l31], b[kk], acc, 0, 0, 0);
  return acc;
}
__device__ __forceinline__ void store_tile(const f32x16& acc, float* __restrict__ out, int64_t ldo, int l31, int h) {
#pragma unroll
  for (int r = 0; r < 16; ++r) out[(int64_t)((r & 3) + 8 * (r >> 2) + 4 * h) * ldo + l31] = acc[r];
}

__global__ __launch_bounds__(TH_T) void th_bwd2_kernel(ThArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char th_smem_raw[];
  ThSmem& sm = *(ThSmem*)th_smem_raw;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
  const int ct = a.d / 32;
  const int m = blockIdx.x / ct, j0 = (blockIdx.x % ct) * 32;
  const ThWs w = th_layout(a.ws, a.N, a.M, a.d);
  const int n = tid >> 5, j = tid & 31, col = j0 + j;
  const int64_t p = ((int64_t)n * a.M + m) * a.d + col;
  const bool live = n < a.N;
  const float dzv = live ? w.dz[p] : 0.f;
  float dx = dzv;
  if (a.use_bn) {
    const float mu = w.mean[(int64_t)m * a.d + col], rs = w.rstd[(int64_t)m * a.d + col], ga = a.h.bn_gamma[m][col];
    const float xh = live ? (w.x1[p] - mu) * rs : 0.f;
    sm.tile[0][n][j] = dzv;
    sm.tile[1][n][j] = dzv * xh;
    __syncthreads();
    {  // column sums over the rows: thread (row r, column c) -- rows >= N hold zeros
      const int r = tid & 31, c = tid >> 5;
      const float s1 = half_wave_sum(sm.tile[0][r][c]), s2 = half_wave_sum(sm.tile[1][r][c]);
      if (r == 0) {
        sm.colv[0][c] = s1;
        sm.colv[1][c] = s2;
        if (a.h.g_bn_beta[m]) a.h.g_bn_beta[m][j0 + c] = s1;
        if (a.h.g_bn_gamma[m]) a.h.g_bn_gamma[m][j0 + c] = s2;
      }
    }
    __syncthreads();
    if (a.training) dx = ga * rs * (dzv - sm.colv[0][j] / a.N - xh * sm.colv[1][j] / a.N);
    else dx = ga * rs * dzv;
    if (!live) dx = 0.f;
  }
  if (live) w.dz[p] = dx;  // dz now holds dx1 (gradient wrt the cg.fc output)
  sm.tile[2][n][j] = dx;
  __syncthreads();
  {
    const float sb = half_wave_sum(sm.tile[2][tid & 31][tid >> 5]);
    if ((tid & 31) == 0 && a.h.g_b2[m]) a.h.g_b2[m][j0 + (tid >> 5)] = sb;
  }
  if (a.h.g_w2[m]) {  // g_w2[m][j0 + i][c] = sum_n dx1[n][j0 + i] * y[n][m][c]
    for (int c_t = wave; c_t < ct; c_t += TH_W) {
      const f32x16 acc = mm_tn_tile(sm.tile[2], w.y + (int64_t)m * a.d + c_t * 32, (int64_t)a.M * a.d, a.N, l31, h);
      store_tile(acc, a.h.g_w2[m] + (int64_t)j0 * a.d + c_t * 32, a.d, l31, h);
    }
  }
}

__global__ __launch_bounds__(TH_T) void th_bwd3_kernel(ThArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char th_smem_raw[];
  ThSmem& sm = *(ThSmem*)th_smem_raw;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
  const int ct = a.d / 32, gemm_blocks = a.M * ct;
  const ThWs w = th_layout(a.ws, a.N, a.M, a.d);
  if ((int)blockIdx.x >= gemm_blocks) {
    // ---- MoE weight gradients: g_moe_w[m][k] = sum_n dlogit[n][m] * text_moe[n][k]; g_moe_b[m] = sum_n dlogit[n][m] ----
    const int kb_n = (a.K + 4 * TH_T - 1) / (4 * TH_T);
    const int q = blockIdx.x - gemm_blocks, m = q / kb_n, k4 = ((q % kb_n) * TH_T + tid) * 4;
    const unsigned key = (!a.text_moe && a.thr16 && a.key_dev) ? *a.key_dev : 0u;
    if (k4 < a.K && a.h.g_moe_w[m]) {
      f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
      for (int n = 0; n < a.N; ++n) s += moe_input4(a, key, n, k4) * w.dlogit[(int64_t)n * a.M + m];
      *(f32x4*)(a.h.g_moe_w[m] + k4) = s;
    }
    if (q % kb_n == 0 && tid == 0 && a.h.g_moe_b[m]) {
      float s = 0.f;
      for (int n = 0; n < a.N; ++n) s += w.dlogit[(int64_t)n * a.M + m];
      a.h.g_moe_b[m][0] = s;
    }
    return;
  }
  const int m = blockIdx.x / ct, j0 = (blockIdx.x % ct) * 32;
  // dy[n][j0 + j] = dyg[n][j0 + j] + sum_c dx1[n][c] * W2[m][c][j0 + j]   (contraction over W2's ROW index)
  const int per = k_slice(a.d), kbeg = wave * per, kend = min(a.d, kbeg + per);
  {
    const float* ap = w.dz + ((int64_t)min(l31, a.N - 1) * a.M + m) * a.d;
    const bool iok = l31 < a.N;
    const float* bp = a.h.w2[m] + j0 + l31;
    f32x4 av[TH_TMAX], bv[TH_TMAX];
#pragma unroll
    for (int t = 0; t < TH_TMAX; ++t) {
      const int k0 = kbeg + 8 * t + 4 * h;
      av[t] = bv[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (k0 + 3 < kend) {  // d % 32 == 0: slices are whole groups
        if (iok) av[t] = *(const f32x4*)(ap + k0);
#pragma unroll
        for (int u = 0; u < 4; ++u) bv[t][u] = bp[(int64_t)(k0 + u) * a.d];
      }
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int t = 0; t < TH_TMAX; ++t) {
      if (kbeg + 8 * t < kend) {
#pragma unroll
        for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t][u], bv[t][u], acc, 0, 0, 0);
      }
    }
    float v = splitk_reduce(sm, acc, wave, lane);
    const int n = tile_row(wave, lane);
    if (n < a.N) {
      const int64_t p = ((int64_t)n * a.M + m) * a.d + j0 + l31;
      v += w.dyg[p];
      w.dyg[p] = v;  // dyg now holds dy (gradient wrt the fc output), read by the text-gradient GEMM
    } else {
      v = 0.f;
    }
    sm.tile[2][n][l31] = v;
  }
  __syncthreads();
  {
    const float sb = half_wave_sum(sm.tile[2][tid & 31][tid >> 5]);
    if ((tid & 31) == 0 && a.h.g_b1[m]) a.h.g_b1[m][j0 + (tid >> 5)] = sb;
  }
  if (a.h.g_w1[m]) {  // g_w1[m][j0 + i][k] = sum_n dy[n][j0 + i] * text[n][k]
    const int kt_n = a.K / 32;
    for (int kt = wave; kt < kt_n; kt += TH_W) {
      const f32x16 acc = mm_tn_tile(sm.tile[2], a.text + kt * 32, a.K, a.N, l31, h);
      store_tile(acc, a.h.g_w1[m] + (int64_t)j0 * a.K + kt * 32, a.K, l31, h);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// The predicate the HOST side asks before it relies on what only these kernels do (on-the-fly MoE dropout, in-kernel
// num_batches_tracked): it therefore includes the lab switch MMT_TEXT_HEADS_V1 (one kernel per op), which the dispatcher in
// texthead.hip obeys -- host and device must agree on the path.
extern "C" int mmt_text_heads_fast(int N, int M, int d, int K) {
  static int off = -1;
  if (off < 0) off = getenv("MMT_TEXT_HEADS_V1") ? 1 : 0;
  return !off && N >= 1 && N <= 32 && M >= 1 && M <= MMT_MAX_EXPERTS && d % 32 == 0 && d >= 32 && d <= 1024 && K % 32 == 0 &&
         K >= 32 && K <= 1024;
}

static int th_configure() {
  static bool done = false;
  if (done) return 0;
  const void* fns[] = {(const void*)th_fwd1_kernel, (const void*)th_fwd2_kernel, (const void*)th_bwd2_kernel,
                       (const void*)th_bwd3_kernel, (const void*)th_fwd1_plan_kernel, (const void*)th_fwd2_cast_kernel};
  for (const void* f : fns) {
    hipError_t rc = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(ThSmem));
    if (rc != hipSuccess) return (int)rc;
  }
  done = true;
  return 0;
}

static void th_fill(ThArgs& a, const MmtTextHeads* h, const float* text, const float* text_moe, int N, int C, int M, int d,
                    int K, int use_bn, int training, float* ws, const MmtTextHeadsOpts* o) {
  a = {};
  a.h = *h;
  a.text = text; a.text_moe = text_moe; a.ws = ws;
  a.N = N; a.C = C; a.M = M; a.d = d; a.K = K; a.use_bn = use_bn; a.training = training;
  if (o) {
    a.drop_key = o->moe_drop_key; a.thr16 = o->moe_drop_thr16; a.drop_scale = o->moe_drop_scale;
    a.seed_dev = o->seed_dev; a.key_dev = o->key_dev; a.nbt = (long long*)o->num_batches_tracked;
  }
}

int mmt_text_heads_fwd_small(const MmtTextHeads* h, const float* text, const float* text_moe, int N, int C, int M, int d,
                             int K, int use_bn, int training, float* ws, float* text_embds, float* text_weights,
                             const MmtTextHeadsOpts* opts, hipStream_t s) {
  if (int rc = th_configure()) return rc;
  ThArgs a;
  th_fill(a, h, text, text_moe, N, C, M, d, K, use_bn, training, ws, opts);
  a.text_embds = text_embds; a.text_weights = text_weights;
  if (!text_moe && a.thr16 && (!a.seed_dev || !a.key_dev)) return MMT_ERR_ARG;
  const int gemm_blocks = M * (d / 32);
  const int blocks1 = gemm_blocks + (text_weights ? N : 0);
  const MmtVideoFront* vf = opts ? opts->video_front : nullptr;
  if (vf) {
    // the plan rides in launch 1 (its seed bump moves to launch 2: the MoE blocks of launch 1 read the seed), the cast in 2
    VideoPlanArgs p;
    if (int e = video_plan_args(p, vf->experts, vf->M, vf->B, vf->T, vf->pack, vf->max_pos, vf->counts, vf->cu_seqlens,
                                vf->n_rows_dev, vf->slot, vf->row_index, vf->type_ids, vf->pos_ids, vf->mask_bias, vf->agg_row,
                                nullptr, vf->src))
      return e;
    if ((size_t)vf->M * vf->T * sizeof(float) > sizeof(ThSmem)) return MMT_ERR_ARG;
    VideoCastArgs c = {};
    CastSplit cs = {};
    int cast_blocks = 0;
    if (vf->do_cast) {
      if (int e = video_cast_args(c, vf->experts, vf->M, vf->B, vf->T, vf->src)) return e;
      // ~400 cast blocks (with the text heads' blocks: two 1024-thread blocks per CU, all carrying the text heads' 78 KiB
      // of dynamic LDS), shared out by input width: rgb / scene (2048 / 2208 channels) get 16x the blocks of audio (128)
      int64_t wsum = 0;
      for (int i = 0; i < vf->M; ++i) wsum += vf->experts[i].Dpad;
      for (int i = 0; i < vf->M; ++i) {
        cs.begin[i] = cast_blocks;
        const int64_t share = (400 * (int64_t)vf->experts[i].Dpad + wsum - 1) / wsum;
        cast_blocks += (int)(share < 1 ? 1 : share);
      }
      cs.begin[vf->M] = cast_blocks;
    }
    hipLaunchKernelGGL(th_fwd1_plan_kernel, dim3(blocks1 + vf->B), dim3(TH_T), sizeof(ThSmem), s, a, p, blocks1);
    hipLaunchKernelGGL(th_fwd2_cast_kernel, dim3(gemm_blocks + cast_blocks), dim3(TH_T), sizeof(ThSmem), s, a, c, gemm_blocks,
                       cs, vf->M, vf->seed_bump);
    if (cast_blocks) {  // second half of the cast: 4x the blocks (a quarter of the threads each)
      CastSplit c3 = cs;
      for (int i = 0; i <= vf->M; ++i) c3.begin[i] = cs.begin[i] * 4;
      hipLaunchKernelGGL(th_fwd3_cast_kernel, dim3((N * M + 3) / 4 + 4 * cast_blocks), dim3(256), 0, s, a, c, (N * M + 3) / 4, c3,
                         vf->M);
      return (int)hipGetLastError();
    }
  } else {
    hipLaunchKernelGGL(th_fwd1_kernel, dim3(blocks1), dim3(TH_T), sizeof(ThSmem), s, a);
    hipLaunchKernelGGL(th_fwd2_kernel, dim3(gemm_blocks), dim3(TH_T), sizeof(ThSmem), s, a);
  }
  hipLaunchKernelGGL(th_fwd3_kernel, dim3((N * M + 3) / 4), dim3(256), 0, s, a);
  return (int)hipGetLastError();
}

int mmt_text_heads_bwd_small(const MmtTextHeads* h, const float* text, const float* text_moe, int N, int C, int M, int d,
                             int K, int use_bn, int training, float* ws, const float* dtext_embds, const float* text_weights,
                             const float* dtext_weights, float* dtext_moe, const MmtTextHeadsOpts* opts, hipStream_t s) {
  if (int rc = th_configure()) return rc;
  ThArgs a;
  th_fill(a, h, text, text_moe, N, C, M, d, K, use_bn, training, ws, opts);
  a.de = dtext_embds; a.tw = text_weights; a.dtw = dtext_weights; a.dtext_moe = dtext_moe;
  const bool moe = text_weights && dtext_weights;
  const int gemm_blocks = M * (d / 32);
  hipLaunchKernelGGL(th_bwd1_kernel, dim3((N * M + 3) / 4 + (moe ? N : 0)), dim3(256), 0, s, a);
  hipLaunchKernelGGL(th_bwd2_kernel, dim3(gemm_blocks), dim3(TH_T), sizeof(ThSmem), s, a);
  const int kb_n = (K + 4 * TH_T - 1) / (4 * TH_T);
  hipLaunchKernelGGL(th_bwd3_kernel, dim3(gemm_blocks + (moe ? M * kb_n : 0)), dim3(TH_T), sizeof(ThSmem), s, a);
  return (int)hipGetLastError();
}
