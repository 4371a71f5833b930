// LayerNorm / embedding kernels of the video-BERT (gfx950).  All are HBM/Infinity-Cache bound:
// one wave per row, 16-byte vector accesses, fp32 statistics (eps = 1e-12 is below bf16 resolution).
//
//   ln_fwd       : h = LN(z)                         bert.py:188,236 (z = pre-LN sum from the GEMM epilogue)
//   embed_ln_fwd : h = dropout(LN(feat + type_emb[t] + pos_emb[p]))          bert.py:87-105
//   ln_bwd       : dz (+ dy = dropout'(dz) in bf16 for the GEMMs) and per-block partial column sums
//                  of dgamma, dbeta, dbias
//   col_reduce   : sums the per-block partials
//   table_grad   : embedding-table gradients (deterministic segmented sums, no atomics)
#include "mmt_common.h"
#include "attn_sched.h"
#include "../../include/mmt_hip.h"

#define MAXC 4  // d <= 1024: up to 4 float4 chunks per lane

template <bool EMBED>
__global__ __launch_bounds__(256) void ln_fwd_kernel(
    const float* __restrict__ z_in, const int32_t* __restrict__ type_ids, const int32_t* __restrict__ pos_ids,
    const float* __restrict__ type_emb, const float* __restrict__ pos_emb, float* __restrict__ z_save,
    const float* __restrict__ gamma, const float* __restrict__ beta, float eps, float* __restrict__ h32,
    bf16_t* __restrict__ h16, float* __restrict__ mean_out, float* __restrict__ rstd_out, int rows, int d,
    const int32_t* __restrict__ n_rows_dev, const int32_t* __restrict__ row_index, uint32_t drop_key_in,
    uint32_t thr16, float drop_scale, const uint32_t* __restrict__ seed_dev, const int32_t* __restrict__ dst_rows,
    int ln_blocks, AttnSched sched) {
  // rider (EMBED launches of the encoder engine): ONE extra block builds the attention backward's block order of this batch
  // (attn_sched.h) -- independent of everything here, needed a forward pass later, and not worth a graph node of its own
  if (EMBED && (int)blockIdx.x >= ln_blocks) {
    if (sched.work) attn_schedule_block(sched);
    return;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nrows = n_rows_dev ? min(*n_rows_dev, rows) : rows;
  const int nch = d >> 8;
  const unsigned drop_key = eff_key(drop_key_in, seed_dev);
  for (int row = blockIdx.x * 4 + wave; row < nrows; row += ln_blocks * 4) {
    const int64_t hrow = dst_rows ? dst_rows[row] : row;  // where the fp32 output row goes (scatter to token rows)
    f32x4 x[MAXC];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      if (c < nch) {
        const int col = c * 256 + lane * 4;
        x[c] = *(const f32x4*)(z_in + (int64_t)row * d + col);
        if constexpr (EMBED) {
          x[c] += *(const f32x4*)(type_emb + (int64_t)type_ids[row] * d + col);
          if (pos_ids) x[c] += *(const f32x4*)(pos_emb + (int64_t)pos_ids[row] * d + col);
          if (z_save) *(f32x4*)(z_save + (int64_t)row * d + col) = x[c];
        }
        s += x[c][0] + x[c][1] + x[c][2] + x[c][3];
      }
    }
    const float mean = wave_sum(s) / (float)d;
    float v = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
      if (c < nch) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { const float t = x[c][k] - mean; v += t * t; }
      }
    const float rstd = 1.0f / sqrtf(wave_sum(v) / (float)d + eps);
    if (lane == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
    const int orow = row_index ? row_index[row] : row;
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
      if (c < nch) {
        const int col = c * 256 + lane * 4;
        const f32x4 g = *(const f32x4*)(gamma + col), b = *(const f32x4*)(beta + col);
        f32x4 y;
#pragma unroll
        for (int k = 0; k < 4; ++k) y[k] = (x[c][k] - mean) * rstd * g[k] + b[k];
        if constexpr (EMBED) {
          if (thr16) {
            bool kp[4];
            keep4(drop_key, (unsigned long long)orow * (unsigned)d + (unsigned)col, thr16, kp);
#pragma unroll
            for (int k = 0; k < 4; ++k) y[k] = kp[k] ? y[k] * drop_scale : 0.f;
          }
        }
        if (h32) *(f32x4*)(h32 + hrow * d + col) = y;
        u32x2 o = {pack_bf2(y[0], y[1]), pack_bf2(y[2], y[3])};
        if (h16) *(u32x2*)(h16 + (int64_t)row * d + col) = o;
      }
  }
}

// Split-K GEMM epilogue + LayerNorm in one pass (the compact last layer: a few hundred rows, where every extra launch
// is ~5 us of pure latency): z = sum_s slab[s][row] + bias -> dropout -> + residual (optionally gathered through
// res_rows) -> saved; h = LN(z).  One wave per row.  orow (RNG coordinate) = rowidx[row] if given, else derived as
// row_index[res_rows[row]] and written to rowidx_out for the kernels that follow.
struct SplitkLnArgs {
  const float* slabs; int splits; int64_t slab_stride;
  const float *bias, *res; const int32_t *res_rows, *rowidx, *row_index; int32_t* rowidx_out;
  uint32_t drop_key, thr16; float drop_scale; const uint32_t* seed_dev;
  float* z_out; const float *gamma, *beta; float eps;
  float* h32; bf16_t* h16; float *mean, *rstd;
  int rows, d;
  const int32_t* n_rows_dev;  // (nullable) live rows of a packed batch
};
__global__ __launch_bounds__(256) void splitk_ln_fwd_kernel(SplitkLnArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + wave;
  if (row >= a.rows || (a.n_rows_dev && row >= *a.n_rows_dev)) return;
  const int d = a.d, nch = d >> 8;
  const int64_t rrow = a.res_rows ? a.res_rows[row] : row;
  int orow;
  if (a.rowidx) orow = a.rowidx[row];
  else {
    orow = a.row_index ? a.row_index[rrow] : (int)rrow;
    if (a.rowidx_out && lane == 0) a.rowidx_out[row] = orow;
  }
  const unsigned dkey = eff_key(a.drop_key, a.seed_dev);
  f32x4 x[MAXC];
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < MAXC; ++c)
    if (c < nch) {
      const int col = c * 256 + lane * 4;
      const float* p0 = a.slabs + (int64_t)row * d + col;
      f32x4 part[16];  // every slab load of the chunk in flight together (splits <= 16), fixed summation order
#pragma unroll
      for (int sp = 0; sp < 16; ++sp)
        part[sp] = sp < a.splits ? *(const f32x4*)(p0 + (int64_t)sp * a.slab_stride) : (f32x4){0.f, 0.f, 0.f, 0.f};
      const f32x4 resv = *(const f32x4*)(a.res + rrow * d + col);
      f32x4 v = part[0];
#pragma unroll
      for (int sp = 1; sp < 16; ++sp) v += part[sp];
      v += *(const f32x4*)(a.bias + col);
      if (a.thr16) {
        bool k[4];
        keep4(dkey, (unsigned long long)orow * (unsigned)d + (unsigned)col, a.thr16, k);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = k[e] ? v[e] * a.drop_scale : 0.f;
      }
      v += resv;
      *(f32x4*)(a.z_out + (int64_t)row * d + col) = v;
      x[c] = v;
      s += v[0] + v[1] + v[2] + v[3];
    }
  const float mean = wave_sum(s) / (float)d;
  float q = 0.f;
#pragma unroll
  for (int c = 0; c < MAXC; ++c)
    if (c < nch) {
#pragma unroll
      for (int k = 0; k < 4; ++k) { const float t = x[c][k] - mean; q += t * t; }
    }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)d + a.eps);
  if (lane == 0) { a.mean[row] = mean; a.rstd[row] = rstd; }
#pragma unroll
  for (int c = 0; c < MAXC; ++c)
    if (c < nch) {
      const int col = c * 256 + lane * 4;
      const f32x4 g = *(const f32x4*)(a.gamma + col), b = *(const f32x4*)(a.beta + col);
      f32x4 y;
#pragma unroll
      for (int k = 0; k < 4; ++k) y[k] = (x[c][k] - mean) * rstd * g[k] + b[k];
      if (a.h32) *(f32x4*)(a.h32 + (int64_t)row * d + col) = y;
      if (a.h16) *(u32x2*)(a.h16 + (int64_t)row * d + col) = (u32x2){pack_bf2(y[0], y[1]), pack_bf2(y[2], y[3])};
    }
}

extern "C" int mmt_splitk_ln_fwd_ex(const float* slabs, int splits, int64_t slab_stride, const float* bias, const float* res,
                                 const int32_t* res_rows, const int32_t* rowidx, const int32_t* row_index, int32_t* rowidx_out,
                                 uint32_t drop_key, uint32_t thr16, float drop_scale, const uint32_t* seed_dev, float* z_out,
                                 const float* gamma, const float* beta, float eps, float* h32, void* h16, float* mean,
                                 float* rstd, int rows, int d, const int32_t* n_rows_dev, void* stream) {
  if (!slabs || splits <= 0 || splits > 16 || !bias || !res || !z_out || !gamma || !beta || !mean || !rstd || rows <= 0) return MMT_ERR_ARG;
  if (d % 256 || d > MAXC * 256) return MMT_ERR_ARG;
  SplitkLnArgs a = {slabs, splits, slab_stride, bias, res, res_rows, rowidx, row_index, rowidx_out, drop_key, thr16, drop_scale,
                    seed_dev, z_out, gamma, beta, eps, h32, (bf16_t*)h16, mean, rstd, rows, d, n_rows_dev};
  hipLaunchKernelGGL(splitk_ln_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}
extern "C" int mmt_splitk_ln_fwd(const float* slabs, int splits, int64_t slab_stride, const float* bias, const float* res,
                                 const int32_t* res_rows, const int32_t* rowidx, const int32_t* row_index, int32_t* rowidx_out,
                                 uint32_t drop_key, uint32_t thr16, float drop_scale, const uint32_t* seed_dev, float* z_out,
                                 const float* gamma, const float* beta, float eps, float* h32, void* h16, float* mean,
                                 float* rstd, int rows, int d, void* stream) {
  return mmt_splitk_ln_fwd_ex(slabs, splits, slab_stride, bias, res, res_rows, rowidx, row_index, rowidx_out, drop_key, thr16,
                              drop_scale, seed_dev, z_out, gamma, beta, eps, h32, h16, mean, rstd, rows, d, nullptr, stream);
}

// DROP: 0 none, 1 dropout applied BEFORE the LN input (dy = mask*dz*scale), 2 dropout applied AFTER the
// LN output (incoming dout is masked first; embeddings).
// 8 waves per block, RPW rows per wave with every load of the wave's rows issued up front (the previous version -- 4 waves
// walking 4 rows each with a one-row prefetch -- ran at 2.3 TB/s with 3.5 waves per CU: latency-bound); the column
// partials of the block's waves are combined through LDS with 16-byte accesses.
#define LNB_WAVES 8
template <int DROP, int RPW, int NCH>
__global__ __launch_bounds__(64 * LNB_WAVES) void ln_bwd_kernel(
    const float* __restrict__ dout, const float* __restrict__ z, const float* __restrict__ mean_in,
    const float* __restrict__ rstd_in, const float* __restrict__ gamma, float* __restrict__ dz_out,
    bf16_t* __restrict__ dy_out, float* __restrict__ partials, int rows,
    const int32_t* __restrict__ n_rows_dev, const int32_t* __restrict__ row_index, uint32_t drop_key_in,
    uint32_t thr16, float drop_scale, const uint32_t* __restrict__ seed_dev, int slab_splits, int64_t slab_stride,
    const float* slab_res) {
  // slab_splits > 0: `dout` is the first of slab_splits split-K partial slabs and the incoming gradient is their sum
  // (+ slab_res): the epilogue of the input-gradient GEMM that precedes this LayerNorm backward, folded in
  extern __shared__ __attribute__((aligned(16))) float red_raw[];  // [LNB_WAVES][3][d]
  const unsigned drop_key = eff_key(drop_key_in, seed_dev);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nrows = n_rows_dev ? min(*n_rows_dev, rows) : rows;
  constexpr int d = NCH * 256;
  f32x4 dg[NCH], db[NCH], dbias[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) dg[c] = db[c] = dbias[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int r_begin = blockIdx.x * (LNB_WAVES * RPW);
  f32x4 go[RPW][NCH], zz[RPW][NCH];
  float mean[RPW], rstd[RPW];
  int orow[RPW];
  bool live[RPW];
#pragma unroll
  for (int q = 0; q < RPW; ++q) {  // all loads of the wave's rows in flight together
    const int row = r_begin + q * LNB_WAVES + wave;
    live[q] = row < nrows;
    const int rr = live[q] ? row : 0;
    mean[q] = mean_in[rr]; rstd[q] = rstd_in[rr];
    orow[q] = row_index ? row_index[rr] : rr;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
      {
        const int col = c * 256 + lane * 4;
        go[q][c] = *(const f32x4*)(dout + (int64_t)rr * d + col);
        zz[q][c] = *(const f32x4*)(z + (int64_t)rr * d + col);
        if (slab_splits > 0) {  // block-uniform; every slab load in flight together (splits <= 16), fixed order
          f32x4 part[15];
#pragma unroll
          for (int sp = 1; sp < 16; ++sp)
            part[sp - 1] = sp < slab_splits ? *(const f32x4*)(dout + (int64_t)sp * slab_stride + (int64_t)rr * d + col)
                                            : (f32x4){0.f, 0.f, 0.f, 0.f};
          f32x4 rv = {0.f, 0.f, 0.f, 0.f};
          if (slab_res) rv = *(const f32x4*)(slab_res + (int64_t)rr * d + col);
#pragma unroll
          for (int sp = 0; sp < 15; ++sp) go[q][c] += part[sp];
          go[q][c] += rv;
        }
      }
  }
  f32x4 gm[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c)
    gm[c] = *(const f32x4*)(gamma + c * 256 + lane * 4);
  f32x4 xh[RPW][NCH], g[RPW][NCH];
  float s1[RPW], s2[RPW];
#pragma unroll
  for (int q = 0; q < RPW; ++q) {
    s1[q] = s2[q] = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
      {
        const int col = c * 256 + lane * 4;
        if constexpr (DROP == 2) {
          if (thr16) {
            bool kp[4];
            keep4(drop_key, (unsigned long long)orow[q] * (unsigned)d + (unsigned)col, thr16, kp);
#pragma unroll
            for (int k = 0; k < 4; ++k) go[q][c][k] = kp[k] ? go[q][c][k] * drop_scale : 0.f;
          }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          xh[q][c][k] = (zz[q][c][k] - mean[q]) * rstd[q];
          g[q][c][k] = go[q][c][k] * gm[c][k];
          s1[q] += g[q][c][k];
          s2[q] += g[q][c][k] * xh[q][c][k];
        }
        if (live[q]) {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            dg[c][k] += go[q][c][k] * xh[q][c][k];
            db[c][k] += go[q][c][k];
          }
        }
      }
  }
  // the 2*RPW wave reductions are independent chains: interleaved by the compiler
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
    for (int q = 0; q < RPW; ++q) {
      s1[q] += __shfl_xor(s1[q], o, 64);
      s2[q] += __shfl_xor(s2[q], o, 64);
    }
  }
#pragma unroll
  for (int q = 0; q < RPW; ++q) {
    if (!live[q]) continue;  // wave-uniform
    const int row = r_begin + q * LNB_WAVES + wave;
    const float m1 = s1[q] / (float)d, m2 = s2[q] / (float)d;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
      {
        const int col = c * 256 + lane * 4;
        f32x4 dzv;
#pragma unroll
        for (int k = 0; k < 4; ++k) dzv[k] = rstd[q] * (g[q][c][k] - m1 - xh[q][c][k] * m2);
        if (dz_out) *(f32x4*)(dz_out + (int64_t)row * d + col) = dzv;
        if (dy_out) {
          f32x4 dyv = dzv;
          if constexpr (DROP == 1) {
            if (thr16) {
              bool kp[4];
              keep4(drop_key, (unsigned long long)orow[q] * (unsigned)d + (unsigned)col, thr16, kp);
#pragma unroll
              for (int k = 0; k < 4; ++k) dyv[k] = kp[k] ? dyv[k] * drop_scale : 0.f;
            }
          }
          u32x2 o = {pack_bf2(dyv[0], dyv[1]), pack_bf2(dyv[2], dyv[3])};
          *(u32x2*)(dy_out + (int64_t)row * d + col) = o;
          // the bias gradient sums exactly what the weight-gradient GEMM will read (bf16-rounded)
          dbias[c][0] += bf2f((bf16_t)(o[0] & 0xffff)); dbias[c][1] += bf2f((bf16_t)(o[0] >> 16));
          dbias[c][2] += bf2f((bf16_t)(o[1] & 0xffff)); dbias[c][3] += bf2f((bf16_t)(o[1] >> 16));
        }
      }
  }
  // cross-wave reduction of the column partials, one [3][d] record per block (fixed order => deterministic)
#pragma unroll
  for (int c = 0; c < NCH; ++c)
    {
      const int col = c * 256 + lane * 4;
      *(f32x4*)(red_raw + (wave * 3 + 0) * d + col) = dg[c];
      *(f32x4*)(red_raw + (wave * 3 + 1) * d + col) = db[c];
      *(f32x4*)(red_raw + (wave * 3 + 2) * d + col) = dbias[c];
    }
  __syncthreads();
  for (int e = threadIdx.x; e < 3 * d / 4; e += 64 * LNB_WAVES) {
    f32x4 acc = *(const f32x4*)(red_raw + e * 4);
#pragma unroll
    for (int w = 1; w < LNB_WAVES; ++w) acc += *(const f32x4*)(red_raw + w * 3 * d + e * 4);
    *(f32x4*)(partials + (int64_t)blockIdx.x * 3 * d + e * 4) = acc;
  }
}

// out[j][c] (+)= sum_b partials[b][j][c]   (j < nvec).  Block = 64 columns x 4 row-groups; each thread sums
// a strided quarter of the blocks, then a fixed-order LDS combine => deterministic.
struct ColOuts { float* p[4]; };
#define CR_RG 16
__global__ __launch_bounds__(64 * CR_RG) void col_reduce_kernel(const float* __restrict__ partials, int nblocks, int nvec,
                                                                int d, ColOuts outs, int accumulate) {
  __shared__ float red[CR_RG][64];
  const int chunks = (d + 63) / 64;
  const int j = blockIdx.x / chunks, c = (blockIdx.x % chunks) * 64 + (threadIdx.x & 63);
  const int rg = threadIdx.x >> 6;
  float* out = outs.p[j];
  float s = 0.f;
  if (out && c < d)
    for (int b = rg; b < nblocks; b += CR_RG) s += partials[((int64_t)b * nvec + j) * d + c];
  red[rg][threadIdx.x & 63] = s;
  __syncthreads();
  if (rg == 0 && out && c < d) {
    const int t = threadIdx.x;
    float v = 0.f;
#pragma unroll
    for (int g = 0; g < CR_RG; ++g) v += red[g][t];
    out[c] = accumulate ? out[c] + v : v;
  }
}

// The same reduction for up to MMT_COLRED_MAX independent jobs in one launch (blockIdx.y = job): the backward pass
// leaves every site's partials in its own buffer and sums them all at the end instead of paying one ~6 us
// launch-latency-bound kernel per LayerNorm.
struct ColJobs { MmtColReduceJob job[MMT_COLRED_MAX]; int begin[MMT_COLRED_MAX]; int count; };
__global__ __launch_bounds__(64 * CR_RG) void col_reduce_multi_kernel(ColJobs jobs) {
  __shared__ float red[CR_RG][64];
  int p = 0;
#pragma unroll 1
  for (int q = 1; q < jobs.count; ++q)
    if ((int)blockIdx.x >= jobs.begin[q]) p = q;
  const MmtColReduceJob& jb = jobs.job[p];
  const int bid = (int)blockIdx.x - jobs.begin[p];
  const int d = jb.d, nvec = jb.nvec, nblocks = jb.nblocks;
  const int chunks = (d + 63) / 64;
  const int j = bid / chunks, c = (bid % chunks) * 64 + (threadIdx.x & 63);
  const int rg = threadIdx.x >> 6;
  float* out = jb.out[j];
  const float* __restrict__ partials = jb.partials;
  float s = 0.f;
  if (out && c < d) {
    float s1 = 0.f, s2 = 0.f, s3 = 0.f;  // independent chains: the strided loads are in flight together
    int b = rg;
    for (; b + 3 * CR_RG < nblocks; b += 4 * CR_RG) {
      s += partials[((int64_t)b * nvec + j) * d + c];
      s1 += partials[((int64_t)(b + CR_RG) * nvec + j) * d + c];
      s2 += partials[((int64_t)(b + 2 * CR_RG) * nvec + j) * d + c];
      s3 += partials[((int64_t)(b + 3 * CR_RG) * nvec + j) * d + c];
    }
    for (; b < nblocks; b += CR_RG) s += partials[((int64_t)b * nvec + j) * d + c];
    s = (s + s1) + (s2 + s3);
  }
  red[rg][threadIdx.x & 63] = s;
  __syncthreads();
  if (rg == 0 && out && c < d) {
    const int t = threadIdx.x;
    float v = 0.f;
#pragma unroll
    for (int g = 0; g < CR_RG; ++g) v += red[g][t];
    out[c] = v;
  }
}

// partial[chunk][v][c] = sum over the chunk's rows with ids[row] == v of g[row][c].  grid = (vocab, d/256, chunks);
// the chunks are then summed by col_reduce_kernel (fixed order => deterministic, no atomics).
#define TABLE_CHUNKS 64
__global__ __launch_bounds__(256) void table_grad_kernel(
    const float* __restrict__ g, const int32_t* __restrict__ ids, int rows, int d, int vocab,
    const int32_t* __restrict__ n_rows_dev, float* __restrict__ partial) {
  const int v = blockIdx.x, col = blockIdx.y * 256 + threadIdx.x, chunk = blockIdx.z;
  const int nrows = n_rows_dev ? min(*n_rows_dev, rows) : rows;
  const int per = (nrows + TABLE_CHUNKS - 1) / TABLE_CHUNKS;
  const int r_begin = chunk * per, r_end = min(nrows, r_begin + per);
  __shared__ int32_t sid[256];
  float s = 0.f;
  for (int r0 = r_begin; r0 < r_end; r0 += 256) {
    const int r = r0 + threadIdx.x;
    sid[threadIdx.x] = r < r_end ? ids[r] : -1;
    __syncthreads();
    const int lim = min(256, r_end - r0);
    for (int k = 0; k < lim; ++k)
      if (sid[k] == v) s += g[(int64_t)(r0 + k) * d + col];
    __syncthreads();
  }
  partial[((int64_t)chunk * vocab + v) * d + col] = s;
}

// The same partial sums on the exact-fp32 matrix cores: partial[chunk] = onehot(ids)^T . g over the chunk's rows
// (v_mfma_f32_32x32x2_f32 is an fp32 fma chain, products 1 * g: exact row-order sums).  A = onehot[v][r] built in
// registers, B = g[r][c0 + lane & 31] (128-byte coalesced).  grid (d / 128, TABLE_CHUNKS, tables): 4 waves per block, one
// 32-column tile each; up to two tables (token types + temporal positions) in one launch.  The scan kernel above walks
// its chunk row by row per vocabulary entry (16.9 + 11.8 us for the two tables of config B); this one needs ~4 us.
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
struct TablePair { const int32_t* ids[2]; float* partial[2]; int vocab[2]; };
#define TG_ROWS 64  // rows per burst: [64][128] fp32 = 32 KiB of LDS, one memory round trip
// NU = vocabulary tiles of 32 per table: 2 (tables of <= 64 rows: MSRVTT's 32 temporal positions, 19 token types) or 4
// (<= 128: ActivityNet's 102 positions -- r03 sent those to the row-scanning kernel below, 666 us per step at S = 708).
template <int NU>
__global__ __launch_bounds__(256) void table_grad_mfma_kernel(const float* __restrict__ g, TablePair tp, int rows, int d,
                                                              const int32_t* __restrict__ n_rows_dev) {
  __shared__ __attribute__((aligned(16))) float gt[TG_ROWS][128 + 4];
  __shared__ int32_t idt[2][TG_ROWS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
  const int cb = blockIdx.x * 128, chunk = blockIdx.y;
  const int nrows = n_rows_dev ? min(*n_rows_dev, rows) : rows;
  const int per = (nrows + TABLE_CHUNKS - 1) / TABLE_CHUNKS;
  const int r_begin = chunk * per, r_end = min(nrows, r_begin + per);
  const int ntab = tp.ids[1] ? 2 : 1;
  f32x16_t acc[2][NU];  // [table][vocabulary tile of 32]
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int u = 0; u < NU; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;
  for (int r0 = r_begin; r0 < r_end; r0 += TG_ROWS) {
    // burst: 64 rows x 128 columns of g (512-byte row segments, 16 B per lane) + the ids of both tables
#pragma unroll
    for (int i = 0; i < TG_ROWS * 32 / 256; ++i) {
      const int e = i * 256 + tid, rr = e >> 5, c4 = (e & 31) * 4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (r0 + rr < r_end) v = *(const f32x4*)(g + (int64_t)(r0 + rr) * d + cb + c4);
      *(f32x4*)(&gt[rr][c4]) = v;
    }
    if (tid < 2 * TG_ROWS) {
      const int t = tid / TG_ROWS, rr = tid % TG_ROWS;
      idt[t][rr] = (t < ntab && r0 + rr < r_end) ? tp.ids[t][r0 + rr] : -1;
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      if (t >= ntab) continue;
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        if (u * 32 >= tp.vocab[t]) continue;
        const int myv = u * 32 + l31;
#pragma unroll 8
        for (int kk = 0; kk < TG_ROWS / 2; ++kk) {
          const int k = 2 * kk + h;
          acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(idt[t][k] == myv ? 1.0f : 0.0f, gt[k][wave * 32 + l31], acc[t][u], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    if (t >= ntab) continue;
#pragma unroll
    for (int u = 0; u < NU; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) {  // lane holds column cb + 32 wave + l31, vocabulary rows (r & 3) + 8 (r >> 2) + 4 h
        const int v = u * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (v < tp.vocab[t]) tp.partial[t][((int64_t)chunk * tp.vocab[t] + v) * d + cb + wave * 32 + l31] = acc[t][u][r];
      }
  }
}

// ------------------------------------------------------------------------------------------------
static inline int ln_grid(int rows) {
  int g = (rows + 3) / 4;
  return g < 1 ? 1 : (g > 4096 ? 4096 : g);
}

extern "C" int mmt_ln_fwd(const float* z, const float* gamma, const float* beta, float eps, float* h32,
                          void* h16, float* mean, float* rstd, int rows, int d, const int32_t* n_rows_dev,
                          void* stream) {
  if (!z || !gamma || !beta || !h16 || !mean || !rstd || rows <= 0) return MMT_ERR_ARG;
  if (d % 256 || d > MAXC * 256) return MMT_ERR_ARG;
  hipLaunchKernelGGL(ln_fwd_kernel<false>, dim3(ln_grid(rows)), dim3(256), 0, (hipStream_t)stream, z,
                     nullptr, nullptr, nullptr, nullptr, nullptr, gamma, beta, eps, h32, (bf16_t*)h16, mean,
                     rstd, rows, d, n_rows_dev, nullptr, 0u, 0u, 1.0f, nullptr, nullptr, ln_grid(rows), AttnSched{});
  return (int)hipGetLastError();
}

// h = LN(z) for a compact set of rows; the fp32 result row i is written to h32[dst_rows[i]] (h16 stays compact, nullable)
extern "C" int mmt_ln_fwd_scatter(const float* z, const float* gamma, const float* beta, float eps, float* h32,
                                  const int32_t* dst_rows, void* h16, float* mean, float* rstd, int rows, int d,
                                  void* stream) {
  if (!z || !gamma || !beta || !h32 || !dst_rows || !mean || !rstd || rows <= 0) return MMT_ERR_ARG;
  if (d % 256 || d > MAXC * 256) return MMT_ERR_ARG;
  hipLaunchKernelGGL(ln_fwd_kernel<false>, dim3(ln_grid(rows)), dim3(256), 0, (hipStream_t)stream, z,
                     nullptr, nullptr, nullptr, nullptr, nullptr, gamma, beta, eps, h32, (bf16_t*)h16, mean,
                     rstd, rows, d, nullptr, nullptr, 0u, 0u, 1.0f, nullptr, dst_rows, ln_grid(rows), AttnSched{});
  return (int)hipGetLastError();
}

// dst[i] = src[rows[i]] (gather) / dst[rows[i]] = src[i] (scatter), fp32 rows of d floats; idx_out[i] = idx_in ?
// idx_in[rows[i]] : rows[i] (gather only, nullable) carries the RNG row coordinates along.
__global__ __launch_bounds__(256) void row_move_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                       const int32_t* __restrict__ rows, int n, int d, int scatter,
                                                       const int32_t* __restrict__ idx_in, int32_t* __restrict__ idx_out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = blockIdx.x * 4 + wave; i < n; i += gridDim.x * 4) {
    const int64_t r = rows[i];
    const float* s = scatter ? src + (int64_t)i * d : src + r * d;
    float* t = scatter ? dst + r * d : dst + (int64_t)i * d;
    if (scatter == 2) {
      for (int c = lane * 4; c < d; c += 256) *(f32x4*)(t + c) += *(const f32x4*)(s + c);
    } else {
      for (int c = lane * 4; c < d; c += 256) *(f32x4*)(t + c) = *(const f32x4*)(s + c);
    }
    if (!scatter && idx_out && lane == 0) idx_out[i] = idx_in ? idx_in[r] : (int32_t)r;
  }
}

extern "C" int mmt_rows_gather(const float* src, const int32_t* rows, int n, int d, float* dst, const int32_t* idx_in,
                               int32_t* idx_out, void* stream) {
  if (!src || !rows || !dst || n <= 0 || d <= 0 || (d & 3)) return MMT_ERR_ARG;
  hipLaunchKernelGGL(row_move_kernel, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, src, dst, rows, n, d, 0, idx_in,
                     idx_out);
  return (int)hipGetLastError();
}
extern "C" int mmt_rows_scatter(const float* src, const int32_t* rows, int n, int d, float* dst, int accumulate,
                                void* stream) {
  if (!src || !rows || !dst || n <= 0 || d <= 0 || (d & 3)) return MMT_ERR_ARG;
  hipLaunchKernelGGL(row_move_kernel, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, src, dst, rows, n, d,
                     accumulate ? 2 : 1, nullptr, nullptr);
  return (int)hipGetLastError();
}

// Table gradient for LARGE vocabularies with few distinct ids in use (BERT-base position table: 512 rows, 30 used):
// one block per table row v sums the token rows with ids[r] == v in row order and writes (zero if none) -- no partial
// buffers proportional to vocab * d.  d <= 1024.
__global__ __launch_bounds__(256) void table_grad_direct_kernel(const float* __restrict__ g, const int32_t* __restrict__ ids,
                                                                int rows, int d, const int32_t* __restrict__ n_rows_dev,
                                                                float* __restrict__ out) {
  const int v = blockIdx.x, c = threadIdx.x * 4;
  const int n = n_rows_dev ? min(rows, *n_rows_dev) : rows;
  if (c >= d) return;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int r = 0; r < n; ++r)
    if (ids[r] == v) acc += *(const f32x4*)(g + (int64_t)r * d + c);  // block-uniform branch
  *(f32x4*)(out + (int64_t)v * d + c) = acc;
}

extern "C" int mmt_table_grad_direct(const float* g, const int32_t* ids, int rows, int d, int vocab,
                                     const int32_t* n_rows_dev, float* dtable, void* stream) {
  if (!g || !ids || !dtable || rows <= 0 || vocab <= 0 || d <= 0 || (d & 3) || d > 1024) return MMT_ERR_ARG;
  hipLaunchKernelGGL(table_grad_direct_kernel, dim3(vocab), dim3(256), 0, (hipStream_t)stream, g, ids, rows, d, n_rows_dev,
                     dtable);
  return (int)hipGetLastError();
}

// Word-embedding gradient (the text tower's nn.Embedding backward, deterministic): dtable[id] = sum over the token rows
// i with ids[i] == id of g[i], in row order.  One block per token row; the block of an id's FIRST occurrence does the
// whole sum for that id (rows <= a few thousand: the id scan is cheap), every other block exits.  `padding_idx` rows get
// no gradient (nn.Embedding(padding_idx=...)).  dtable must be zero on entry (untouched ids keep a zero gradient).
__global__ __launch_bounds__(256) void embedding_grad_kernel(const float* __restrict__ g, const int32_t* __restrict__ ids,
                                                             int n, int d, int vocab, int padding_idx,
                                                             const int32_t* __restrict__ n_rows_dev,
                                                             float* __restrict__ dtable) {
  const int i = blockIdx.x;
  if (n_rows_dev) n = min(n, *n_rows_dev);
  if (i >= n) return;
  const int id = ids[i];
  if (id < 0 || id >= vocab || id == padding_idx) return;
  int seen = 0;
  for (int j = threadIdx.x; j < i; j += 256) seen |= (ids[j] == id);
  if (__syncthreads_or(seen)) return;
  for (int c = threadIdx.x * 4; c < d; c += 1024) {
    f32x4 acc = *(const f32x4*)(g + (int64_t)i * d + c);
    for (int j = i + 1; j < n; ++j)
      if (ids[j] == id) acc += *(const f32x4*)(g + (int64_t)j * d + c);
    *(f32x4*)(dtable + (int64_t)id * d + c) = acc;
  }
}

extern "C" int mmt_embedding_grad(const float* g, const int32_t* ids, int n, int d, int vocab, int padding_idx,
                                  const int32_t* n_rows_dev, float* dtable, void* stream) {
  if (!g || !ids || !dtable || n <= 0 || d <= 0 || (d & 3) || vocab <= 0) return MMT_ERR_ARG;
  hipLaunchKernelGGL(embedding_grad_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, g, ids, n, d, vocab, padding_idx,
                     n_rows_dev, dtable);
  return (int)hipGetLastError();
}

extern "C" int mmt_embed_ln_fwd_sched(const float* features, const int32_t* type_ids, const int32_t* pos_ids,
                                const float* type_emb, const float* pos_emb, float* z_save,
                                const float* gamma, const float* beta, float eps, float* h32, void* h16,
                                float* mean, float* rstd, int rows, int d, const int32_t* n_rows_dev,
                                const int32_t* row_index, uint32_t drop_key, uint32_t thr16, float drop_scale,
                                const uint32_t* seed_dev, const int32_t* sched_cu, int sched_B, int sched_H, int sched_tiles,
                                      int32_t* sched_work, void* stream) {
  if (!features || !type_ids || !type_emb || !gamma || !beta || !h16 || !mean || !rstd || rows <= 0)
    return MMT_ERR_ARG;
  if (pos_ids && !pos_emb) return MMT_ERR_ARG;
  if (d % 256 || d > MAXC * 256) return MMT_ERR_ARG;
  const AttnSched sc = {sched_cu, sched_work, sched_B, sched_H, sched_tiles};
  hipLaunchKernelGGL(ln_fwd_kernel<true>, dim3(ln_grid(rows) + (sched_work ? 1 : 0)), dim3(256), 0, (hipStream_t)stream, features,
                     type_ids, pos_ids, type_emb, pos_emb, z_save, gamma, beta, eps, h32, (bf16_t*)h16, mean,
                     rstd, rows, d, n_rows_dev, row_index, drop_key, thr16, drop_scale, seed_dev, nullptr, ln_grid(rows), sc);
  return (int)hipGetLastError();
}
extern "C" int mmt_embed_ln_fwd(const float* features, const int32_t* type_ids, const int32_t* pos_ids,
                                const float* type_emb, const float* pos_emb, float* z_save,
                                const float* gamma, const float* beta, float eps, float* h32, void* h16,
                                float* mean, float* rstd, int rows, int d, const int32_t* n_rows_dev,
                                const int32_t* row_index, uint32_t drop_key, uint32_t thr16, float drop_scale,
                                const uint32_t* seed_dev, void* stream) {
  return mmt_embed_ln_fwd_sched(features, type_ids, pos_ids, type_emb, pos_emb, z_save, gamma, beta, eps, h32, h16, mean, rstd,
                                rows, d, n_rows_dev, row_index, drop_key, thr16, drop_scale, seed_dev, nullptr, 0, 0, 0, nullptr,
                                stream);
}

// rows per block: 16 (two rows per wave) for big inputs; 8 (one row per wave) when there are few rows, so that a
// compact last-layer LayerNorm is not a 14-block launch
extern "C" int mmt_ln_bwd_rows_per_block(int rows) { return rows <= 2048 ? 8 : 16; }

static int ln_bwd_impl(const float* dout, const float* z, const float* mean, const float* rstd,
                          const float* gamma, float* dz, void* dy, float* partials, int rows, int d,
                          int drop_mode, const int32_t* n_rows_dev, const int32_t* row_index,
                          uint32_t drop_key, uint32_t thr16, float drop_scale, const uint32_t* seed_dev,
                          int slab_splits, int64_t slab_stride, const float* slab_res, void* stream) {
  if (!dout || !z || !mean || !rstd || !gamma || !partials || rows <= 0) return MMT_ERR_ARG;
  if (d % 256 || d > MAXC * 256) return MMT_ERR_ARG;
  const int rpb = mmt_ln_bwd_rows_per_block(rows), grid = (rows + rpb - 1) / rpb;
  hipStream_t s = (hipStream_t)stream;
  const size_t lds = (size_t)LNB_WAVES * 3 * d * sizeof(float);
#define LN_BWD_LAUNCH(MODE, RPW, NCH)                                                                             \
  do {                                                                                                            \
    static bool configured = false;                                                                               \
    if (!configured && lds > 64 * 1024) {                                                                         \
      hipError_t rc = hipFuncSetAttribute((const void*)ln_bwd_kernel<MODE, RPW, NCH>,                             \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                  \
      if (rc != hipSuccess) return (int)rc;                                                                       \
      configured = true;                                                                                          \
    }                                                                                                             \
    hipLaunchKernelGGL((ln_bwd_kernel<MODE, RPW, NCH>), dim3(grid), dim3(64 * LNB_WAVES), lds, s, dout, z, mean, rstd, gamma, \
                       dz, (bf16_t*)dy, partials, rows, n_rows_dev, row_index, drop_key, thr16, drop_scale, seed_dev,  \
                       slab_splits, slab_stride, slab_res);                                                       \
  } while (0)
#define LN_BWD_NCH(MODE, RPW)                                    \
  switch (d >> 8) {                                              \
    case 1: LN_BWD_LAUNCH(MODE, RPW, 1); break;                  \
    case 2: LN_BWD_LAUNCH(MODE, RPW, 2); break;                  \
    case 3: LN_BWD_LAUNCH(MODE, RPW, 3); break;                  \
    default: LN_BWD_LAUNCH(MODE, RPW, 4); break;                 \
  }
  if (drop_mode < 0 || drop_mode > 2) return MMT_ERR_ARG;
  if (rpb == 16) {
    if (drop_mode == 0) { LN_BWD_NCH(0, 2) } else if (drop_mode == 1) { LN_BWD_NCH(1, 2) } else { LN_BWD_NCH(2, 2) }
  } else {
    if (drop_mode == 0) { LN_BWD_NCH(0, 1) } else if (drop_mode == 1) { LN_BWD_NCH(1, 1) } else { LN_BWD_NCH(2, 1) }
  }
#undef LN_BWD_NCH
#undef LN_BWD_LAUNCH
  return (int)hipGetLastError();
}

extern "C" int mmt_ln_bwd(const float* dout, const float* z, const float* mean, const float* rstd,
                          const float* gamma, float* dz, void* dy, float* partials, int rows, int d,
                          int drop_mode, const int32_t* n_rows_dev, const int32_t* row_index,
                          uint32_t drop_key, uint32_t thr16, float drop_scale, const uint32_t* seed_dev,
                          void* stream) {
  return ln_bwd_impl(dout, z, mean, rstd, gamma, dz, dy, partials, rows, d, drop_mode, n_rows_dev, row_index, drop_key, thr16,
                     drop_scale, seed_dev, 0, 0, nullptr, stream);
}

// the same with the incoming gradient given as split-K partial slabs (+ an optional residual gradient): dout = sum_s
// slabs[s] + res -- the ADD_F32 epilogue of the input-gradient GEMM in front of this LayerNorm, without its own launch
extern "C" int mmt_ln_bwd_slabs_ex(const float* slabs, int splits, int64_t slab_stride, const float* res, const float* z,
                                const float* mean, const float* rstd, const float* gamma, float* dz, void* dy,
                                float* partials, int rows, int d, int drop_mode, const int32_t* n_rows_dev, const int32_t* row_index,
                                uint32_t drop_key, uint32_t thr16, float drop_scale, const uint32_t* seed_dev, void* stream) {
  if (splits <= 0 || splits > 16) return MMT_ERR_ARG;
  return ln_bwd_impl(slabs, z, mean, rstd, gamma, dz, dy, partials, rows, d, drop_mode, n_rows_dev, row_index, drop_key, thr16,
                     drop_scale, seed_dev, splits, slab_stride, res, stream);
}
extern "C" int mmt_ln_bwd_slabs(const float* slabs, int splits, int64_t slab_stride, const float* res, const float* z,
                                const float* mean, const float* rstd, const float* gamma, float* dz, void* dy,
                                float* partials, int rows, int d, int drop_mode, const int32_t* row_index,
                                uint32_t drop_key, uint32_t thr16, float drop_scale, const uint32_t* seed_dev, void* stream) {
  return mmt_ln_bwd_slabs_ex(slabs, splits, slab_stride, res, z, mean, rstd, gamma, dz, dy, partials, rows, d, drop_mode, nullptr,
                             row_index, drop_key, thr16, drop_scale, seed_dev, stream);
}

extern "C" int mmt_col_reduce(const float* partials, int nblocks, int nvec, int d, float* out0, float* out1,
                              float* out2, float* out3, int accumulate, void* stream) {
  if (!partials || nblocks <= 0 || nvec <= 0 || nvec > 4 || d <= 0) return MMT_ERR_ARG;
  ColOuts outs = {{out0, out1, out2, out3}};
  hipLaunchKernelGGL(col_reduce_kernel, dim3(nvec * ((d + 63) / 64)), dim3(64 * CR_RG), 0, (hipStream_t)stream, partials,
                     nblocks, nvec, d, outs, accumulate);
  return (int)hipGetLastError();
}

extern "C" int mmt_col_reduce_multi(const MmtColReduceJob* jobs, int n, void* stream) {
  if (!jobs || n <= 0) return MMT_ERR_ARG;
  for (int base = 0; base < n; base += MMT_COLRED_MAX) {
    ColJobs tab;
    const int cnt = n - base < MMT_COLRED_MAX ? n - base : MMT_COLRED_MAX;
    tab.count = cnt;
    int gx = 0;
    for (int i = 0; i < cnt; ++i) {
      const MmtColReduceJob& jb = jobs[base + i];
      if (!jb.partials || jb.nblocks <= 0 || jb.nvec <= 0 || jb.nout <= 0 || jb.nout > 4 || jb.nout > jb.nvec || jb.d <= 0)
        return MMT_ERR_ARG;
      tab.job[i] = jb;
      tab.begin[i] = gx;
      gx += jb.nout * ((jb.d + 63) / 64);
    }
    hipLaunchKernelGGL(col_reduce_multi_kernel, dim3(gx), dim3(64 * CR_RG), 0, (hipStream_t)stream, tab);
  }
  return (int)hipGetLastError();
}

// table_grad split in two so that the final sum can join a mmt_col_reduce_multi batch: partials only.
extern "C" int mmt_table_grad_partials(const float* g, const int32_t* ids, int rows, int d, int vocab,
                                       const int32_t* n_rows_dev, float* scratch, void* stream) {
  if (!g || !ids || !scratch || rows <= 0 || vocab <= 0 || d % 256) return MMT_ERR_ARG;
  hipLaunchKernelGGL(table_grad_kernel, dim3(vocab, d / 256, TABLE_CHUNKS), dim3(256), 0, (hipStream_t)stream, g, ids,
                     rows, d, vocab, n_rows_dev, scratch);
  return (int)hipGetLastError();
}
extern "C" int mmt_table_grad_chunks(void) { return TABLE_CHUNKS; }

// both embedding tables of the video BERT (token types, temporal positions; ids1 nullable) in ONE launch on the fp32 MFMA
extern "C" int mmt_table_grad_partials_pair(const float* g, const int32_t* ids0, int vocab0, float* scratch0,
                                            const int32_t* ids1, int vocab1, float* scratch1, int rows, int d,
                                            const int32_t* n_rows_dev, void* stream) {
  if (!g || !ids0 || !scratch0 || rows <= 0 || vocab0 <= 0 || vocab0 > 128 || d % 128) return MMT_ERR_ARG;
  if (ids1 && (!scratch1 || vocab1 <= 0 || vocab1 > 128)) return MMT_ERR_ARG;
  TablePair tp = {{ids0, ids1}, {scratch0, scratch1}, {vocab0, vocab1}};
  if (vocab0 <= 64 && (!ids1 || vocab1 <= 64))
    hipLaunchKernelGGL(table_grad_mfma_kernel<2>, dim3(d / 128, TABLE_CHUNKS), dim3(256), 0, (hipStream_t)stream, g, tp, rows, d,
                       n_rows_dev);
  else
    hipLaunchKernelGGL(table_grad_mfma_kernel<4>, dim3(d / 128, TABLE_CHUNKS), dim3(256), 0, (hipStream_t)stream, g, tp, rows, d,
                       n_rows_dev);
  return (int)hipGetLastError();
}

extern "C" int64_t mmt_table_grad_scratch_floats(int vocab, int d) { return (int64_t)TABLE_CHUNKS * vocab * d; }

extern "C" int mmt_table_grad(const float* g, const int32_t* ids, int rows, int d, int vocab,
                              const int32_t* n_rows_dev, float* scratch, float* dtable, int accumulate,
                              void* stream) {
  if (!g || !ids || !dtable || !scratch || rows <= 0 || vocab <= 0 || d % 256) return MMT_ERR_ARG;
  hipLaunchKernelGGL(table_grad_kernel, dim3(vocab, d / 256, TABLE_CHUNKS), dim3(256), 0, (hipStream_t)stream, g, ids,
                     rows, d, vocab, n_rows_dev, scratch);
  ColOuts outs = {{dtable, nullptr, nullptr, nullptr}};
  const int n = vocab * d;
  hipLaunchKernelGGL(col_reduce_kernel, dim3((n + 63) / 64), dim3(64 * CR_RG), 0, (hipStream_t)stream, scratch, TABLE_CHUNKS, 1,
                     n, outs, accumulate);
  return (int)hipGetLastError();
}
