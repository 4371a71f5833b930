// Expert read-out, gated text-video similarity and the max-margin ranking loss (gfx950).
//
//   readout  : vid_embds[b][m] = F.normalize(last_hidden[agg_row[b][m]])     model.py:583-587,621-625
//   sims     : sims[t][v] = sum_m w[t][v][m] <T[t][m], V[v][m]>,  w = tw*vw / sum_m(tw*vw) (0 -> 1e-5)
//              model.py:789-837 (sharded_cross_view_inner_product), fp32
//   loss     : MaxMarginRankingLoss (loss.py:38-65) in closed form + its gradient matrix; InfoNCE (:68-81)
//
// These are tiny (n = batch of pairs), latency-bound fp32 kernels: one block per text row / video column,
// wave-level dot products, fixed summation orders (deterministic).
#include "mmt_common.h"
#include "../../include/mmt_hip.h"

#define MAXM MMT_MAX_EXPERTS

// ---- read-out --------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void readout_fwd_kernel(const float* __restrict__ last, const int32_t* __restrict__ agg_row,
                                                          int BM, int d, float* __restrict__ out, float* __restrict__ inv_norm) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = blockIdx.x * 4 + wave; i < BM; i += gridDim.x * 4) {
    const float* src = last + (int64_t)agg_row[i] * d;
    float ss = 0.f;
    for (int c = lane * 4; c < d; c += 256) {
      const f32x4 v = *(const f32x4*)(src + c);
      ss += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    const float inv = 1.0f / fmaxf(sqrtf(wave_sum(ss)), 1e-12f);
    for (int c = lane * 4; c < d; c += 256) *(f32x4*)(out + (int64_t)i * d + c) = *(const f32x4*)(src + c) * inv;
    if (lane == 0) inv_norm[i] = inv;
  }
}

// dlast[agg_row] = (g - yhat (yhat.g)) * inv ; every other row of dlast must already be zero
__global__ __launch_bounds__(256) void readout_bwd_kernel(const float* __restrict__ emb, const float* __restrict__ inv_norm,
                                                          const float* __restrict__ demb, const int32_t* __restrict__ agg_row,
                                                          int BM, int d, float* __restrict__ dlast) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = blockIdx.x * 4 + wave; i < BM; i += gridDim.x * 4) {
    float dot = 0.f;
    for (int c = lane * 4; c < d; c += 256) {
      const f32x4 y = *(const f32x4*)(emb + (int64_t)i * d + c), g = *(const f32x4*)(demb + (int64_t)i * d + c);
      dot += y[0] * g[0] + y[1] * g[1] + y[2] * g[2] + y[3] * g[3];
    }
    dot = wave_sum(dot);
    const float inv = inv_norm[i];
    float* dst = dlast + (int64_t)agg_row[i] * d;
    for (int c = lane * 4; c < d; c += 256) {
      const f32x4 y = *(const f32x4*)(emb + (int64_t)i * d + c), g = *(const f32x4*)(demb + (int64_t)i * d + c);
      *(f32x4*)(dst + c) = (g - y * dot) * inv;
    }
  }
}

// ---- similarity ------------------------------------------------------------------------------------
// txt [NT][M][d], vid [NV][M][d], tw [NT][M], vw [NV][M]; sims [NT][NV]; dots [NT][NV][M] saved for backward.
__global__ __launch_bounds__(256) void sims_fwd_kernel(const float* __restrict__ txt, const float* __restrict__ vid,
                                                       const float* __restrict__ tw, const float* __restrict__ vw, int NT,
                                                       int NV, int M, int d, float* __restrict__ sims,
                                                       float* __restrict__ dots) {
  extern __shared__ __attribute__((aligned(16))) float ts[];  // [M][d] text row
  const int t = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x * 4; i < M * d; i += 1024) *(f32x4*)(ts + i) = *(const f32x4*)(txt + (int64_t)t * M * d + i);
  __syncthreads();
  for (int v = blockIdx.y * 4 + wave; v < NV; v += gridDim.y * 4) {
    float nrm = 0.f;
    for (int m = 0; m < M; ++m) nrm += tw[t * M + m] * vw[v * M + m];
    if (nrm == 0.f) nrm = 1e-5f;  // model.py:816
    float s = 0.f;
    for (int m = 0; m < M; ++m) {
      float acc = 0.f;
      for (int c = lane * 4; c < d; c += 256) {
        const f32x4 x = *(const f32x4*)(ts + m * d + c), y = *(const f32x4*)(vid + ((int64_t)v * M + m) * d + c);
        acc += x[0] * y[0] + x[1] * y[1] + x[2] * y[2] + x[3] * y[3];
      }
      const float dm = wave_sum(acc);
      s += tw[t * M + m] * vw[v * M + m] / nrm * dm;
      if (lane == 0) dots[((int64_t)t * NV + v) * M + m] = dm;
    }
    if (lane == 0) sims[(int64_t)t * NV + v] = s;
  }
}

// The same for small batches with every (video, expert) pair of a text row spread over 16 waves: the per-expert loop of
// sims_fwd_kernel is a chain of M dependent wave reductions per wave (10.8 us at n = 32); here all of a wave's dot
// products are independent (6 us).  LDS: text row [M][d] + dots row [NV][M].
#define SF_WAVES 16
#define SF_VCHUNK 8  // videos per block
__global__ __launch_bounds__(64 * SF_WAVES) void sims_fwd_small_kernel(const float* __restrict__ txt, const float* __restrict__ vid,
                                                                       const float* __restrict__ tw, const float* __restrict__ vw,
                                                                       int NT, int NV, int M, int d, float* __restrict__ sims,
                                                                       float* __restrict__ dots) {
  extern __shared__ __attribute__((aligned(16))) float ts[];  // [M][d] text row | [SF_VCHUNK][M] dots of this block
  float* drow = ts + M * d;
  const int t = blockIdx.x, v0 = blockIdx.y * SF_VCHUNK, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nv = min(SF_VCHUNK, NV - v0);
  for (int i = threadIdx.x * 4; i < M * d; i += 256 * SF_WAVES) *(f32x4*)(ts + i) = *(const f32x4*)(txt + (int64_t)t * M * d + i);
  __syncthreads();
  const int npair = nv * M;
  for (int p = wave; p < npair; p += 2 * SF_WAVES) {  // pair p = (video v0 + p / M, expert p % M); two pairs per trip
    const int p1 = p + SF_WAVES;
    const bool two = p1 < npair;
    const int m0 = p % M, m1 = two ? p1 % M : 0;
    const float* y0 = vid + ((int64_t)v0 * M + p) * d;
    const float* y1 = vid + ((int64_t)v0 * M + (two ? p1 : p)) * d;
    float a0 = 0.f, a1 = 0.f;
    for (int c = lane * 4; c < d; c += 256) {
      const f32x4 u0 = *(const f32x4*)(y0 + c), u1 = *(const f32x4*)(y1 + c);
      const f32x4 x0 = *(const f32x4*)(ts + m0 * d + c), x1 = *(const f32x4*)(ts + m1 * d + c);
      a0 += x0[0] * u0[0] + x0[1] * u0[1] + x0[2] * u0[2] + x0[3] * u0[3];
      a1 += x1[0] * u1[0] + x1[1] * u1[1] + x1[2] * u1[2] + x1[3] * u1[3];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a0 += __shfl_xor(a0, o, 64); a1 += __shfl_xor(a1, o, 64); }
    if (lane == 0) {
      drow[p] = a0;
      dots[((int64_t)t * NV + v0) * M + p] = a0;
      if (two) { drow[p1] = a1; dots[((int64_t)t * NV + v0) * M + p1] = a1; }
    }
  }
  __syncthreads();
  if ((int)threadIdx.x < nv) {
    const int v = v0 + threadIdx.x;
    float nrm = 0.f, sacc = 0.f;
    for (int m = 0; m < M; ++m) nrm += tw[t * M + m] * vw[v * M + m];
    if (nrm == 0.f) nrm = 1e-5f;  // model.py:816
    for (int m = 0; m < M; ++m) sacc += tw[t * M + m] * vw[v * M + m] / nrm * drow[threadIdx.x * M + m];
    sims[(int64_t)t * NV + v] = sacc;
  }
}

// ---- loss + d loss / d sims + similarity backward (+ read-out backward) in ONE launch, n <= 64 ------------------------
// Every block loads the whole n x n similarity matrix (<= 16 KB) and derives what it needs of the loss gradient
// G = d loss / d sims itself: max-margin (loss.py:38-65) needs s_tv, s_tt, s_vv per entry and the active-hinge counts of
// row/column k for the diagonal; InfoNCE (loss.py:68-81) the row and column logsumexps.  Block (0, 0, 0) also writes the
// loss.  Then the similarity backward of sims_bwd_kernel with 16 waves over the other side's rows, and for the video
// side optionally the backward of the read-out normalisation (readout_bwd_kernel) straight into the engine's dlast.
#define SB_WAVES 8  // 118 VGPRs: two 8-wave blocks per CU, so the n * M * 2 = 448 blocks of config B are resident at once
struct SimLossArgs {
  const float *txt, *vid, *tw, *vw, *sims, *dots;
  int n, M, d, kind, fix_norm;
  float margin;
  float *loss, *dtxt, *dvid, *dtw, *dvw;
  const float* inv_norm; const int32_t* out_rows; float* dlast;
};
__global__ __launch_bounds__(64 * SB_WAVES) void simloss_bwd_small_kernel(SimLossArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sl[];
  const int n = a.n, M = a.M, d = a.d;
  float* S = sl;                         // [n][n]
  float* lse_r = S + n * n;              // [n]  (InfoNCE) / diag counts scratch
  float* lse_c = lse_r + n;              // [n]
  float* gself = lse_c + n;              // [n]  G of this block's row (side 0) or column (side 1)
  // (S + lse_r + lse_c + gself rounded up to whole 16-byte slots: racc is accessed as f32x4 -- for odd n the prefix
  // n*n + 2n + roundup4(n) is itself odd)
  float* racc = sl + ((n * n + 3 * n + 3) & ~3) + 4;  // [SB_WAVES][d]
  float* rw = racc + SB_WAVES * d;       // [SB_WAVES]
  float* twl = rw + SB_WAVES;            // [n][M] text / video mixture weights
  float* vwl = twl + n * M;
  const int side = blockIdx.z, self = blockIdx.x, m = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < n * n; i += 64 * SB_WAVES) S[i] = a.sims[i];
  for (int i = tid; i < n * M; i += 64 * SB_WAVES) { twl[i] = a.tw[i]; vwl[i] = a.vw[i]; }
  __syncthreads();
  const float norm = a.kind == 0 ? (a.fix_norm ? 2.0f * n * (n - 1) : 2.0f * n * n) : (float)n;
  if (a.kind == 1) {  // InfoNCE: logsumexp of every row and column
    for (int k = tid; k < 2 * n; k += 64 * SB_WAVES) {
      const int r = k % n;
      const bool col = k >= n;
      float mx = -INFINITY;
      for (int c = 0; c < n; ++c) mx = fmaxf(mx, col ? S[c * n + r] : S[r * n + c]);
      float sum = 0.f;
      for (int c = 0; c < n; ++c) sum += expf((col ? S[c * n + r] : S[r * n + c]) - mx);
      (col ? lse_c : lse_r)[r] = mx + logf(sum);
    }
    __syncthreads();
  }
  // G entry (t, v)
  auto g_entry = [&](int t, int v) -> float {
    const float stv = S[t * n + v];
    if (a.kind == 1) return (expf(stv - lse_r[t]) + expf(stv - lse_c[v]) - (t == v ? 2.f : 0.f)) / norm;
    if (t != v) return ((a.margin - S[t * n + t] + stv > 0.f ? 1.f : 0.f) + (a.margin - S[v * n + v] + stv > 0.f ? 1.f : 0.f)) / norm;
    int cnt = 0;  // diagonal: active hinges of row t and column t pull s_tt down
    for (int c = 0; c < n; ++c) {
      if (c == t) continue;
      cnt += (a.margin - stv + S[t * n + c] > 0.f) + (a.margin - stv + S[c * n + t] > 0.f);
    }
    return -(float)cnt / norm;
  };
  for (int o = tid; o < n; o += 64 * SB_WAVES) gself[o] = side == 0 ? g_entry(self, o) : g_entry(o, self);
  if (side == 0 && self == 0 && m == 0 && wave == 0) {  // the loss itself (fixed order: deterministic)
    float acc = 0.f;
    if (a.kind == 1) {
      for (int k = lane; k < n; k += 64) acc += (lse_r[k] + lse_c[k] - 2.0f * S[k * n + k]) / norm;
    } else {
      for (int e = lane; e < n * n; e += 64) {
        const int r = e / n, c = e % n;
        if (r == c && a.fix_norm) continue;
        acc += (fmaxf(a.margin - S[r * n + r] + S[e], 0.f) + fmaxf(a.margin - S[c * n + c] + S[e], 0.f)) / norm;
      }
    }
    acc = wave_sum(acc);
    if (lane == 0) *a.loss = acc;
  }
  __syncthreads();
  // ---- similarity backward for (side, self, m) ----
  float* __restrict__ dx = side == 0 ? a.dtxt : a.dvid;
  float* __restrict__ dwt = side == 0 ? a.dtw : a.dvw;
  const float* other = side == 0 ? a.vid : a.txt;
  f32x4 acc[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float dws = 0.f;
  const float* wo = side == 0 ? vwl : twl;
  for (int o0 = wave; o0 < n; o0 += 2 * SB_WAVES) {  // two rows of the other side per trip: their loads fly together
    float coef[2];
    f32x4 rowv[2][4];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int o = o0 + q * SB_WAVES;
      coef[q] = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c) rowv[q][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (o >= n) continue;  // wave-uniform
      const int t = side == 0 ? self : o, v = side == 0 ? o : self;
      const float g = gself[o];
      const float* dm = a.dots + ((int64_t)t * n + v) * M;
      float nrm = 0.f, gw = 0.f;
      for (int j = 0; j < M; ++j) nrm += twl[t * M + j] * vwl[v * M + j];
      const bool zero = nrm == 0.f;
      if (zero) nrm = 1e-5f;
      for (int j = 0; j < M; ++j) gw += g * dm[j] * twl[t * M + j] * vwl[v * M + j] / nrm;
      const float w = twl[t * M + m] * vwl[v * M + m] / nrm;
      const float da = zero ? g * dm[m] / nrm : (g * dm[m] - gw) / nrm;
      dws += da * wo[o * M + m];
      coef[q] = g * w;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int col = c * 256 + lane * 4;
        if (col < d) rowv[q][c] = *(const f32x4*)(other + ((int64_t)o * M + m) * d + col);
      }
    }
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[c] += rowv[q][c] * coef[q];
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int col = c * 256 + lane * 4;
    if (col < d) *(f32x4*)(racc + wave * d + col) = acc[c];
  }
  if (lane == 0) rw[wave] = dws;
  __syncthreads();
  // combine the waves' partial rows: thread i owns columns i and i + 512 (d <= 1024)
  const bool readout = side == 1 && a.dlast != nullptr;
  const int64_t row = (int64_t)self * M + m;
  float part = 0.f;
  float gx[2] = {0.f, 0.f};
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int i = tid + u * 64 * SB_WAVES;
    if (i < d) {
      float sum = 0.f;
#pragma unroll
      for (int w = 0; w < SB_WAVES; ++w) sum += racc[w * d + i];
      if (dx) dx[row * d + i] = sum;
      if (readout) part += sum * a.vid[row * d + i];
      gx[u] = sum;
    }
  }
  if (tid == 0 && dwt) {
    float sum = 0.f;
    for (int w = 0; w < SB_WAVES; ++w) sum += rw[w];
    dwt[self * M + m] = sum;
  }
  if (readout) {  // dlast = (g - y <y, g>) * inv   (y = normalised read-out row: model.py:621-625 backward)
    part = wave_sum(part);
    __syncthreads();  // racc is free again
    if (lane == 0) racc[wave] = part;
    __syncthreads();
    float dot = 0.f;
    for (int w = 0; w < SB_WAVES; ++w) dot += racc[w];
    const int64_t orow = a.out_rows ? a.out_rows[row] : row;
    const float inv = a.inv_norm[row];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int i = tid + u * 64 * SB_WAVES;
      if (i < d) a.dlast[orow * d + i] = (gx[u] - a.vid[row * d + i] * dot) * inv;
    }
  }
}

// side 0: block per text row t -> dtxt[t], dtw[t];  side 1: block per video v -> dvid[v], dvw[v].
__global__ __launch_bounds__(256) void sims_bwd_kernel(const float* __restrict__ txt, const float* __restrict__ vid,
                                                       const float* __restrict__ tw, const float* __restrict__ vw,
                                                       const float* __restrict__ dots, const float* __restrict__ dsims,
                                                       int NT, int NV, int M, int d, float* __restrict__ dtxt,
                                                       float* __restrict__ dtw, float* __restrict__ dvid,
                                                       float* __restrict__ dvw) {
  __shared__ __attribute__((aligned(16))) float racc[4][1024];
  __shared__ float rw[4];
  const int side = blockIdx.z;  // both sides in ONE launch
  const int self = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (self >= (side == 0 ? NT : NV)) return;
  float* __restrict__ dx = side == 0 ? dtxt : dvid;
  float* __restrict__ dwt = side == 0 ? dtw : dvw;
  const int NO = side == 0 ? NV : NT;  // the other side's count
  const float* other = side == 0 ? vid : txt;
  const float* wother = side == 0 ? vw : tw;
  {
    const int m = blockIdx.y;
    f32x4 acc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float dws = 0.f;
    for (int o = wave; o < NO; o += 4) {
      const int t = side == 0 ? self : o, v = side == 0 ? o : self;
      const float g = dsims[(int64_t)t * NV + v];
      const float* dm = dots + ((int64_t)t * NV + v) * M;
      float nrm = 0.f, gw = 0.f;
      for (int j = 0; j < M; ++j) nrm += tw[t * M + j] * vw[v * M + j];
      const bool zero = nrm == 0.f;
      if (zero) nrm = 1e-5f;
      for (int j = 0; j < M; ++j) gw += g * dm[j] * tw[t * M + j] * vw[v * M + j] / nrm;  // sum_j dw_j w_j
      const float w = tw[t * M + m] * vw[v * M + m] / nrm;
      // d sims / d a_m = (dots_m - sum_j w_j dots_j) / nrm   (no normaliser gradient in the 1e-5 branch)
      const float da = zero ? g * dm[m] / nrm : (g * dm[m] - gw) / nrm;
      dws += da * wother[o * M + m];
      const float coef = g * w;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int col = c * 256 + lane * 4;
        if (col < d) acc[c] += *(const f32x4*)(other + ((int64_t)o * M + m) * d + col) * coef;
      }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int col = c * 256 + lane * 4;
      if (col < d) *(f32x4*)(&racc[wave][col]) = acc[c];
    }
    if (lane == 0) rw[wave] = dws;
    __syncthreads();
    for (int i = threadIdx.x; i < d; i += 256)
      dx[((int64_t)self * M + m) * d + i] = racc[0][i] + racc[1][i] + racc[2][i] + racc[3][i];
    if (threadIdx.x == 0) dwt[self * M + m] = rw[0] + rw[1] + rw[2] + rw[3];
  }
}

// ---- similarity for larger batches (the GLOBAL batch of a data-parallel step: n = 32 * ranks) -----------------------
// The row-per-block kernels above do n*M*d multiply-adds per block on the vector ALU: fine at n = 32 (7 us), 88 us
// forward and 515 us backward at n = 256.  From LARGE_N on the expert dot products run on the exact-fp32 matrix cores
// instead (mmt_sgemm_batched, one GEMM per expert): dots laid out [M][NT][NV], and in the backward the coefficient
// matrices H_m = dsims * w_m (written over the dots) feed two more batched GEMMs for dtxt / dvid.
#define LARGE_N 64
// sims[t][v] = sum_m w_m dots_m,  w_m = tw[t][m] vw[v][m] / nrm  (nrm == 0 -> 1e-5, model.py:816)
__global__ __launch_bounds__(256) void sims_combine_kernel(const float* __restrict__ dots, const float* __restrict__ tw,
                                                           const float* __restrict__ vw, int NT, int NV, int M,
                                                           float* __restrict__ sims) {
  const int t = blockIdx.y, v = blockIdx.x * 256 + threadIdx.x;
  if (v >= NV) return;
  const int64_t plane = (int64_t)NT * NV, at = (int64_t)t * NV + v;
  float nrm = 0.f;
  for (int m = 0; m < M; ++m) nrm += tw[t * M + m] * vw[v * M + m];
  if (nrm == 0.f) nrm = 1e-5f;
  float s = 0.f;
  for (int m = 0; m < M; ++m) s += tw[t * M + m] * vw[v * M + m] / nrm * dots[m * plane + at];
  sims[at] = s;
}

// da_m[t][v] = d loss / d (tw[t][m] vw[v][m])  (see sims_bwd_kernel); MODE 0: block per video v -> dvw[v][m] (reads only);
// MODE 1: block per text t -> dtw[t][m], then dots_m[t][v] <- H_m = dsims * w_m in place.
template <int MODE>
__global__ __launch_bounds__(256) void sims_weights_bwd_kernel(float* __restrict__ dots, const float* __restrict__ dsims,
                                                               const float* __restrict__ tw, const float* __restrict__ vw,
                                                               int NT, int NV, int M, float* __restrict__ dw) {
  __shared__ float red[4][MAXM];
  const int self = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int NO = MODE == 1 ? NV : NT;
  const int64_t plane = (int64_t)NT * NV;
  float acc[MAXM];
#pragma unroll
  for (int m = 0; m < MAXM; ++m) acc[m] = 0.f;
  for (int o = threadIdx.x; o < NO; o += 256) {
    const int t = MODE == 1 ? self : o, v = MODE == 1 ? o : self;
    const int64_t at = (int64_t)t * NV + v;
    const float g = dsims[at];
    float a[MAXM], dm[MAXM];
    float nrm = 0.f, gw = 0.f;
#pragma unroll
    for (int m = 0; m < MAXM; ++m)
      if (m < M) { a[m] = tw[t * M + m] * vw[v * M + m]; dm[m] = dots[m * plane + at]; nrm += a[m]; }
    const bool zero = nrm == 0.f;
    if (zero) nrm = 1e-5f;
#pragma unroll
    for (int m = 0; m < MAXM; ++m)
      if (m < M) gw += g * dm[m] * a[m] / nrm;
#pragma unroll
    for (int m = 0; m < MAXM; ++m)
      if (m < M) {
        const float da = zero ? g * dm[m] / nrm : (g * dm[m] - gw) / nrm;
        acc[m] += da * (MODE == 1 ? vw[v * M + m] : tw[t * M + m]);
        if (MODE == 1) dots[m * plane + at] = g * a[m] / nrm;  // H_m
      }
  }
#pragma unroll
  for (int m = 0; m < MAXM; ++m) {
    const float r = wave_sum(acc[m]);
    if (lane == 0) red[wave][m] = r;
  }
  __syncthreads();
  if (threadIdx.x < M) dw[self * M + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] +
                                                    red[3][threadIdx.x];
}

static int sims_fwd_large(const float* txt, const float* vid, const float* tw, const float* vw, int NT, int NV, int M, int d,
                          float* sims, float* dots, void* stream) {
  MmtSgemm g = {};
  g.batch = M; g.M = NT; g.N = NV; g.K = d;
  g.sai = (int64_t)M * d; g.sak = 1; g.sbj = (int64_t)M * d; g.sbk = 1; g.ldc = NV;
  for (int m = 0; m < M; ++m) { g.A[m] = txt + (int64_t)m * d; g.B[m] = vid + (int64_t)m * d; g.C[m] = dots + (int64_t)m * NT * NV; }
  int rc = mmt_sgemm_batched(&g, stream);
  if (rc) return rc;
  hipLaunchKernelGGL(sims_combine_kernel, dim3((NV + 255) / 256, NT), dim3(256), 0, (hipStream_t)stream, dots, tw, vw, NT, NV,
                     M, sims);
  return (int)hipGetLastError();
}

static int sims_bwd_large(const float* txt, const float* vid, const float* tw, const float* vw, float* dots,
                          const float* dsims, int NT, int NV, int M, int d, float* dtxt, float* dvid, float* dtw,
                          float* dvw, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(sims_weights_bwd_kernel<0>, dim3(NV), dim3(256), 0, s, dots, dsims, tw, vw, NT, NV, M, dvw);
  hipLaunchKernelGGL(sims_weights_bwd_kernel<1>, dim3(NT), dim3(256), 0, s, dots, dsims, tw, vw, NT, NV, M, dtw);
  const int64_t plane = (int64_t)NT * NV;
  MmtSgemm g = {};  // dtxt[t][m][:] = sum_v H_m[t][v] vid[v][m][:]
  g.batch = M; g.M = NT; g.N = d; g.K = NV;
  g.sai = NV; g.sak = 1; g.sbj = 1; g.sbk = (int64_t)M * d; g.ldc = (int64_t)M * d;
  for (int m = 0; m < M; ++m) { g.A[m] = dots + m * plane; g.B[m] = vid + (int64_t)m * d; g.C[m] = dtxt + (int64_t)m * d; }
  int rc = mmt_sgemm_batched(&g, stream);
  if (rc) return rc;
  g = {};           // dvid[v][m][:] = sum_t H_m[t][v] txt[t][m][:]
  g.batch = M; g.M = NV; g.N = d; g.K = NT;
  g.sai = 1; g.sak = NV; g.sbj = 1; g.sbk = (int64_t)M * d; g.ldc = (int64_t)M * d;
  for (int m = 0; m < M; ++m) { g.A[m] = dots + m * plane; g.B[m] = txt + (int64_t)m * d; g.C[m] = dvid + (int64_t)m * d; }
  return mmt_sgemm_batched(&g, stream);
}

// ---- losses ------------------------------------------------------------------------------------------
// Block k handles diagonal index k: row k (first hinge direction) and column k (second direction).
// partial[k] = sum_{c!=k} relu(m - s_kk + s_kc) + sum_{r!=k} relu(m - s_kk + s_rk)   (fix_norm: off-diagonal only)
// G[r][c] = d loss / d s_rc  (already divided by the normaliser).
__global__ __launch_bounds__(256) void maxmargin_kernel(const float* __restrict__ s, int n, float margin, int fix_norm,
                                                        float* __restrict__ partial, float* __restrict__ G) {
  __shared__ float redf[4];
  __shared__ int redi[4];
  const int k = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float skk = s[(int64_t)k * n + k];
  const float norm = fix_norm ? 2.0f * n * (n - 1) : 2.0f * n * n;
  float acc = 0.f;
  int cnt = 0;  // active hinges that pull s_kk down
  for (int c = threadIdx.x; c < n; c += 256) {
    const bool diag = c == k;
    if (diag && fix_norm) continue;
    const float h1 = margin - skk + s[(int64_t)k * n + c];                       // row k, column c
    const float h2 = margin - skk + s[(int64_t)c * n + k];                       // row c, column k
    const float h2g = margin - s[(int64_t)c * n + c] + s[(int64_t)k * n + c];    // second hinge of element (k, c)
    acc += fmaxf(h1, 0.f) + fmaxf(h2, 0.f);
    cnt += (h1 > 0.f) + (h2 > 0.f);
    if (!diag) G[(int64_t)k * n + c] = ((h1 > 0.f ? 1.f : 0.f) + (h2g > 0.f ? 1.f : 0.f)) / norm;
  }
  acc = wave_sum(acc);
  for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
  if (lane == 0) { redf[wave] = acc; redi[wave] = cnt; }
  __syncthreads();
  if (threadIdx.x == 0) {
    partial[k] = (redf[0] + redf[1] + redf[2] + redf[3]) / norm;
    const int total = redi[0] + redi[1] + redi[2] + redi[3];
    // diagonal: with fix_norm the (k,k) terms are excluded; without it they contribute relu(margin) with zero
    // gradient (s_kk cancels), but still count in `total` twice -> remove them.
    const int self_terms = fix_norm ? 0 : (margin > 0.f ? 2 : 0);
    G[(int64_t)k * n + k] = -(float)(total - self_terms) / norm;
  }
}

__global__ void sum_partials_kernel(const float* __restrict__ partial, int n, float* __restrict__ out) {
  __shared__ float red[4];
  float acc = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) acc += partial[i];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) *out = red[0] + red[1] + red[2] + red[3];
}

// InfoNCE: loss = CE(rows) + CE(cols), mean over n.  Block k: logsumexp of row k and of column k.
// G = (softmax_row + softmax_col - 2 I) / n is produced by a second pass once both lse vectors exist.
__global__ __launch_bounds__(256) void infonce_lse_kernel(const float* __restrict__ s, int n, float* __restrict__ lse_row,
                                                          float* __restrict__ lse_col, float* __restrict__ partial) {
  __shared__ float red[2][4];
  const int k = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float mr = -INFINITY, mc = -INFINITY;
  for (int c = threadIdx.x; c < n; c += 256) { mr = fmaxf(mr, s[(int64_t)k * n + c]); mc = fmaxf(mc, s[(int64_t)c * n + k]); }
  mr = wave_max(mr); mc = wave_max(mc);
  if (lane == 0) { red[0][wave] = mr; red[1][wave] = mc; }
  __syncthreads();
  mr = fmaxf(fmaxf(red[0][0], red[0][1]), fmaxf(red[0][2], red[0][3]));
  mc = fmaxf(fmaxf(red[1][0], red[1][1]), fmaxf(red[1][2], red[1][3]));
  __syncthreads();
  float sr = 0.f, sc = 0.f;
  for (int c = threadIdx.x; c < n; c += 256) { sr += expf(s[(int64_t)k * n + c] - mr); sc += expf(s[(int64_t)c * n + k] - mc); }
  sr = wave_sum(sr); sc = wave_sum(sc);
  if (lane == 0) { red[0][wave] = sr; red[1][wave] = sc; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float lr = mr + logf(red[0][0] + red[0][1] + red[0][2] + red[0][3]);
    const float lc = mc + logf(red[1][0] + red[1][1] + red[1][2] + red[1][3]);
    lse_row[k] = lr; lse_col[k] = lc;
    partial[k] = (lr + lc - 2.0f * s[(int64_t)k * n + k]) / n;
  }
}
__global__ void infonce_grad_kernel(const float* __restrict__ s, int n, const float* __restrict__ lse_row,
                                    const float* __restrict__ lse_col, float* __restrict__ G) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= (int64_t)n * n) return;
  const int r = (int)(i / n), c = (int)(i % n);
  G[i] = (expf(s[i] - lse_row[r]) + expf(s[i] - lse_col[c]) - (r == c ? 2.f : 0.f)) / n;
}

// ------------------------------------------------------------------------------------------------
extern "C" int mmt_readout_fwd(const float* last_hidden, const int32_t* agg_row, int BM, int d, float* vid_embds,
                               float* inv_norm, void* stream) {
  if (!last_hidden || !agg_row || !vid_embds || !inv_norm || BM <= 0 || d % 4) return MMT_ERR_ARG;
  hipLaunchKernelGGL(readout_fwd_kernel, dim3((BM + 3) / 4), dim3(256), 0, (hipStream_t)stream, last_hidden, agg_row,
                     BM, d, vid_embds, inv_norm);
  return (int)hipGetLastError();
}

extern "C" int mmt_readout_bwd(const float* vid_embds, const float* inv_norm, const float* dvid_embds,
                               const int32_t* agg_row, int BM, int d, float* dlast_hidden, void* stream) {
  if (!vid_embds || !inv_norm || !dvid_embds || !agg_row || !dlast_hidden || BM <= 0 || d % 4) return MMT_ERR_ARG;
  hipLaunchKernelGGL(readout_bwd_kernel, dim3((BM + 3) / 4), dim3(256), 0, (hipStream_t)stream, vid_embds, inv_norm,
                     dvid_embds, agg_row, BM, d, dlast_hidden);
  return (int)hipGetLastError();
}

extern "C" int mmt_sims_fwd(const float* txt, const float* vid, const float* tw, const float* vw, int NT, int NV, int M,
                            int d, float* sims, float* dots, void* stream) {
  if (!txt || !vid || !tw || !vw || !sims || !dots || NT <= 0 || NV <= 0 || M <= 0 || M > MAXM || d % 4 || d > 1024)
    return MMT_ERR_ARG;
  if (NT >= LARGE_N || NV >= LARGE_N) return sims_fwd_large(txt, vid, tw, vw, NT, NV, M, d, sims, dots, stream);
  const size_t lds = (size_t)M * d * sizeof(float);
  if (lds > 64 * 1024) return MMT_ERR_ARG;
  if (lds + (size_t)SF_VCHUNK * M * sizeof(float) <= 64 * 1024) {
    hipLaunchKernelGGL(sims_fwd_small_kernel, dim3(NT, (NV + SF_VCHUNK - 1) / SF_VCHUNK), dim3(64 * SF_WAVES),
                       lds + (size_t)SF_VCHUNK * M * sizeof(float), (hipStream_t)stream, txt, vid, tw, vw, NT, NV, M, d, sims,
                       dots);
    return (int)hipGetLastError();
  }
  int gy = (NV + 3) / 4;
  if (gy > 16) gy = 16;
  hipLaunchKernelGGL(sims_fwd_kernel, dim3(NT, gy), dim3(256), lds, (hipStream_t)stream, txt, vid, tw, vw, NT, NV, M, d,
                     sims, dots);
  return (int)hipGetLastError();
}

extern "C" int mmt_simloss_small_max_n(void) { return LARGE_N - 1; }

extern "C" int mmt_simloss_bwd_small(const float* txt, const float* vid, const float* tw, const float* vw, const float* sims,
                                     const float* dots, int n, int M, int d, int kind, float margin, int fix_norm,
                                     float* loss, float* dtxt, float* dvid, float* dtw, float* dvw, const float* inv_norm,
                                     const int32_t* out_rows, float* dlast, void* stream) {
  if (!txt || !vid || !tw || !vw || !sims || !dots || !loss || n <= 1 || n >= LARGE_N || M <= 0 || M > MAXM || d % 4 ||
      d > 1024 || (kind != 0 && kind != 1))
    return MMT_ERR_ARG;
  if (dlast && !inv_norm) return MMT_ERR_ARG;
  SimLossArgs a = {txt, vid, tw, vw, sims, dots, n, M, d, kind, fix_norm, margin, loss, dtxt, dvid, dtw, dvw, inv_norm, out_rows,
                   dlast};
  const size_t lds = ((size_t)(((n * n + 3 * n + 3) & ~3) + 4) + (size_t)SB_WAVES * d + SB_WAVES + 2 * (size_t)n * M) * sizeof(float);
  static bool configured = false;
  if (!configured) {
    hipError_t rc = hipFuncSetAttribute((const void*)simloss_bwd_small_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)((63 * 63 + 2 * 63 + 64 + SB_WAVES * 1024 + SB_WAVES + 2 * 63 * MAXM) * sizeof(float)));
    if (rc != hipSuccess) return (int)rc;
    configured = true;
  }
  hipLaunchKernelGGL(simloss_bwd_small_kernel, dim3(n, M, 2), dim3(64 * SB_WAVES), lds, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}

extern "C" int mmt_sims_bwd(const float* txt, const float* vid, const float* tw, const float* vw, float* dots,
                            const float* dsims, int NT, int NV, int M, int d, float* dtxt, float* dvid, float* dtw,
                            float* dvw, void* stream) {
  if (!txt || !vid || !tw || !vw || !dots || !dsims || !dtxt || !dvid || !dtw || !dvw) return MMT_ERR_ARG;
  if (NT <= 0 || NV <= 0 || M <= 0 || M > MAXM || d % 4 || d > 1024) return MMT_ERR_ARG;
  if (NT >= LARGE_N || NV >= LARGE_N)
    return sims_bwd_large(txt, vid, tw, vw, dots, dsims, NT, NV, M, d, dtxt, dvid, dtw, dvw, stream);
  hipLaunchKernelGGL(sims_bwd_kernel, dim3(NT > NV ? NT : NV, M, 2), dim3(256), 0, (hipStream_t)stream, txt, vid, tw, vw, dots,
                     dsims, NT, NV, M, d, dtxt, dtw, dvid, dvw);
  return (int)hipGetLastError();
}

extern "C" int mmt_maxmargin(const float* sims, int n, float margin, int fix_norm, float* partial, float* loss,
                             float* grad, void* stream) {
  if (!sims || !partial || !loss || !grad || n <= 1) return MMT_ERR_ARG;
  hipLaunchKernelGGL(maxmargin_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, sims, n, margin, fix_norm, partial,
                     grad);
  hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partial, n, loss);
  return (int)hipGetLastError();
}

extern "C" int mmt_infonce(const float* sims, int n, float* scratch, float* loss, float* grad, void* stream) {
  if (!sims || !scratch || !loss || !grad || n <= 0) return MMT_ERR_ARG;  // scratch: 3n floats
  hipLaunchKernelGGL(infonce_lse_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, sims, n, scratch, scratch + n,
                     scratch + 2 * n);
  hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, scratch + 2 * n, n, loss);
  const int64_t nn = (int64_t)n * n;
  hipLaunchKernelGGL(infonce_grad_kernel, dim3((unsigned)((nn + 255) / 256)), dim3(256), 0, (hipStream_t)stream, sims, n,
                     scratch, scratch + n, grad);
  return (int)hipGetLastError();
}
