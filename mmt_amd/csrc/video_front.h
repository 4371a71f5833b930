// Pieces of the video-token pipeline (assemble.hip) that can also ride along in another kernel's launch: the token plan
// and the fp32 -> bf16 cast of the expert features are written as block-level device functions, so that the text heads'
// first two launches (texthead2.hip: 144 and 112 blocks of 1024 threads -- half a chip) can carry them as extra blocks
// instead of the step spending two more dependent launches (13 + 7 us) on them.
#pragma once
#include "mmt_common.h"
#include "../../include/mmt_hip.h"

struct ExpertTable { MmtExpertIO e[MMT_MAX_EXPERTS]; };

__device__ __forceinline__ void decode_slot(int s, int T, int& expert, int& j) {
  expert = (s - 1) / (T + 1);
  j = (s - 1) % (T + 1);  // 0 = AGG, 1..T = FEA t = j-1
}

// grid = B blocks of 256 threads, ONE launch: block b derives the row offset of its sample (and, per expert, the offset
// of its valid feature rows inside the expert's COMPACT source matrix) from the validity flags of the samples in front of
// it (<= B*M*T flags: trivial), then fills the slot map, ids, mask and the source-row maps.
//
// Compact source matrix of expert e (X_e / Y_e / dY_e): rows [0, B) = the max-pooled vector of every sample (the AGG
// token's input, always present: keep_missing_modalities), rows B + i = the VALID feature rows in (sample, time) order.
// Padded feature rows are never projected: the ReduceDim GEMM, its weight gradient and the cast only see live rows
// (src_cnt[e] on the device; ~52 % of B*(T+1) at the synthetic MSRVTT fill).  Without token packing every feature row
// counts as valid (the dense token grid of the reference).
// (compact per-expert tables: the launches that carry these blocks also carry the text heads' arguments, and a kernel's
// argument block is limited to 4 KiB)
struct PlanExperts { const float* ind[MMT_MAX_EXPERTS]; const float* t[MMT_MAX_EXPERTS]; int32_t type_idx[MMT_MAX_EXPERTS]; };
struct CastExperts {
  const float* feat[MMT_MAX_EXPERTS]; const float* maxpool[MMT_MAX_EXPERTS]; void* x[MMT_MAX_EXPERTS];
  int32_t D[MMT_MAX_EXPERTS], Dpad[MMT_MAX_EXPERTS];
};
struct VideoPlanArgs {
  PlanExperts tab;
  int B, M, T, S, pack, max_pos;
  int32_t *counts, *cu, *n_rows, *slot, *row_index, *type_ids, *pos_ids;
  float* mask_bias;
  int32_t* agg_row;
  uint32_t* seed_bump;
  MmtVideoSrc src;
};
// Block b of the plan: the FIRST 256 threads of the block (tid < 256; further waves of a bigger block must have left --
// finished waves do not take part in the workgroup barriers).  ind_s: M * T floats of LDS.
__device__ __forceinline__ void video_plan_block(const VideoPlanArgs& p, const int b, const int tid, float* ind_s) {
  __shared__ int scan[256];
  __shared__ int carry;
  __shared__ int offs[MMT_MAX_EXPERTS], own[MMT_MAX_EXPERTS];  // valid feature rows of expert e: before sample b / in it
  const PlanExperts& tab = p.tab;
  const int B = p.B, M = p.M, T = p.T, S = p.S, pack = p.pack, max_pos = p.max_pos;
  int32_t* __restrict__ counts = p.counts; int32_t* __restrict__ cu = p.cu; int32_t* __restrict__ n_rows = p.n_rows;
  int32_t* __restrict__ slot = p.slot; int32_t* __restrict__ row_index = p.row_index; int32_t* __restrict__ type_ids = p.type_ids;
  int32_t* __restrict__ pos_ids = p.pos_ids; float* __restrict__ mask_bias = p.mask_bias; int32_t* __restrict__ agg_row = p.agg_row;
  uint32_t* __restrict__ seed_bump = p.seed_bump;
  const MmtVideoSrc src = p.src;
  const int lane = tid & 63, wave = tid >> 6;
  if (b == 0 && tid == 0 && seed_bump) *seed_bump += 1u;  // per-step dropout seed (one launch less)
  for (int i = tid; i < M * T; i += 256) ind_s[i] = tab.ind[i / T][(int64_t)b * T + i % T];
  for (int ex = wave; ex < M; ex += 4) {  // one wave per expert
    const float* __restrict__ ind_e = tab.ind[ex];
    int before = 0, mine = 0;
    if (pack) {
      for (int i = lane; i < b * T; i += 64) before += ind_e[i] != 0.f;
      for (int t = lane; t < T; t += 64) mine += ind_e[(int64_t)b * T + t] != 0.f;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) { before += __shfl_xor(before, o, 64); mine += __shfl_xor(mine, o, 64); }
    } else {
      before = b * T;
      mine = T;
    }
    if (lane == 0) { offs[ex] = before; own[ex] = mine; }
  }
  if (tid == 0) carry = 0;
  __syncthreads();
  int base = b * S;
  if (pack) {
    base = b * (1 + M);
    for (int ex = 0; ex < M; ++ex) base += offs[ex];
  }
  if (b == B - 1 && tid < M && src.src_cnt) src.src_cnt[tid] = B + offs[tid] + own[tid];
  for (int s0 = 0; s0 < S; s0 += 256) {
    const int s = s0 + tid;
    int live = 0, expert = 0, j = 0;
    float ind = 1.f;
    if (s < S) {
      if (s == 0) live = 1;
      else {
        decode_slot(s, T, expert, j);
        if (j == 0) live = 1;
        else { ind = ind_s[expert * T + (j - 1)]; live = pack ? (ind != 0.f) : 1; }
      }
    }
    scan[tid] = live;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {  // Hillis-Steele inclusive scan
      const int v = tid >= o ? scan[tid - o] : 0;
      __syncthreads();
      scan[tid] += v;
      __syncthreads();
    }
    const int before = carry + scan[tid] - live;
    if (s < S) {
      const int row = live ? base + before : -1;
      slot[(int64_t)b * S + s] = row;
      if (live) {
        row_index[row] = b * S + s;
        int type = 0, pos = 0, srow = -1;
        float mask = 1.f;
        if (s > 0) {
          type = tab.type_idx[expert];
          const float* ind_e = ind_s + expert * T;
          if (j == 0) {
            float mx = 0.f;  // th.max(features_ind, 1)  model.py:330
            for (int t = 0; t < T; ++t) mx = fmaxf(mx, ind_e[t]);
            mask = mx;
            agg_row[b * M + expert] = row;
            srow = b;  // the max-pooled rows lead the compact source matrix
          } else {
            mask = ind;
            float tv = tab.t[expert][(int64_t)b * T + (j - 1)];
            tv = fminf(fmaxf(tv, 0.f), (float)max_pos);  // clamp_ then .long()  model.py:516-520
            pos = (int)tv;
            int rank = j - 1;
            if (pack) {
              rank = 0;
              for (int t = 0; t < j - 1; ++t) rank += ind_e[t] != 0.f;
            }
            srow = B + offs[expert] + rank;
            if (src.xsrc) src.xsrc[(int64_t)expert * B * T + offs[expert] + rank] = b * T + (j - 1);
          }
        }
        type_ids[row] = type;
        pos_ids[row] = pos;
        mask_bias[row] = (1.0f - mask) * -10000.0f;  // bert.py:395
        if (src.src_row) src.src_row[row] = srow;
      }
    }
    __syncthreads();
    if (tid == 255) carry += scan[255];
    __syncthreads();
  }
  if (tid == 0) {
    counts[b] = carry;
    cu[b] = base;
    if (b == B - 1) { cu[B] = base + carry; *n_rows = base + carry; }
  }
}


// X_e (compact, see video_plan_kernel): row b < B = maxpool[b]; row B + i = features row xsrc[e][i]; bf16, K zero-padded.
// Only the src_cnt[e] live rows are written: the rows behind them are never read as results (the GEMM's tiles past the
// live count exit, the weight gradient zeroes the ragged tail of its last 64-row unit).
struct VideoCastArgs { CastExperts tab; int B, T; MmtVideoSrc src; };
// Block bx of nbx (of nthreads threads each) working on expert ex.
// part / parts: this launch handles the part-th of `parts` equal slices of every expert's elements (the cast can be spread
// over two launches that carry it as a rider).
__device__ __forceinline__ void video_cast_block(const VideoCastArgs& c, const int ex, const int bx, const int nbx,
                                                 const int tid, const int nthreads, const int part = 0, const int parts = 1) {
  struct { const float *feat, *maxpool; void* x; int D, Dpad; } e = {c.tab.feat[ex], c.tab.maxpool[ex], c.tab.x[ex],
                                                                     c.tab.D[ex], c.tab.Dpad[ex]};
  const int B = c.B, T = c.T;
  const MmtVideoSrc src = c.src;
  const int rows = src.src_cnt[ex];
  const int32_t* __restrict__ xs = src.xsrc + (int64_t)ex * B * T;
  const int64_t n_all = (int64_t)rows * (e.Dpad / 4);
  const int64_t lo = n_all * part / parts, n = n_all * (part + 1) / parts;  // elements [lo, n)
  const int64_t stride = (int64_t)nbx * nthreads;
  const bool vec = !(e.D & 3);
  // four elements per pass, their (dependent) row lookups and then their feature loads in flight together: a thread that
  // walks its elements one by one spends a lookup + a load round trip on each
  for (int64_t i0 = lo + bx * (int64_t)nthreads + tid; i0 < n; i0 += 4 * stride) {
    int r[4], cc[4];
    int64_t srow[4];
    bool ok[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t i = i0 + u * stride;
      ok[u] = i < n;
      const int64_t ii = ok[u] ? i : 0;
      r[u] = (int)(ii / (e.Dpad / 4));
      cc[u] = (int)(ii % (e.Dpad / 4)) * 4;
      srow[u] = r[u] < B ? -1 : (int64_t)xs[r[u] - B];
    }
    f32x4 q[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float* srcp = r[u] < B ? e.maxpool + (int64_t)r[u] * e.D : e.feat + srow[u] * e.D;
      q[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (ok[u]) {
        if (vec && cc[u] + 3 < e.D) {
          q[u] = *(const f32x4*)(srcp + cc[u]);
        } else {
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if (cc[u] + k < e.D) q[u][k] = srcp[cc[u] + k];
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (ok[u]) {
        u32x2 o = {pack_bf2(q[u][0], q[u][1]), pack_bf2(q[u][2], q[u][3])};
        *(u32x2*)((bf16_t*)e.x + (int64_t)r[u] * e.Dpad + cc[u]) = o;
      }
  }
}

static inline int make_table(const MmtExpertIO* experts, int M, ExpertTable& tab) {
  if (!experts || M <= 0 || M > MMT_MAX_EXPERTS) return MMT_ERR_ARG;
  for (int i = 0; i < M; ++i) {
    tab.e[i] = experts[i];
    if (experts[i].n_part < 0 || experts[i].n_part > 2) return MMT_ERR_ARG;
    for (int k = 0; k < experts[i].n_part; ++k)
      if (!experts[i].y_part[k]) return MMT_ERR_ARG;
  }
  return 0;
}


static inline int check_src(const MmtVideoSrc* src) {
  return (src && src->src_row && src->src_cnt && src->xsrc) ? 0 : MMT_ERR_ARG;
}


static inline int video_plan_args(VideoPlanArgs& p, const MmtExpertIO* experts, int M, int B, int T, int pack, int max_pos,
                                  int32_t* counts, int32_t* cu_seqlens, int32_t* n_rows_dev, int32_t* slot, int32_t* row_index,
                                  int32_t* type_ids, int32_t* pos_ids, float* mask_bias, int32_t* agg_row, uint32_t* seed_bump,
                                  const MmtVideoSrc* src) {
  if (!experts || M <= 0 || M > MMT_MAX_EXPERTS) return MMT_ERR_ARG;
  if (!counts || !cu_seqlens || !n_rows_dev || !slot || !row_index || !type_ids || !pos_ids || !mask_bias || !agg_row)
    return MMT_ERR_ARG;
  if (int e = check_src(src)) return e;
  if (B <= 0 || T <= 0 || max_pos < 0) return MMT_ERR_ARG;
  for (int i = 0; i < M; ++i)
    if (!experts[i].ind || !experts[i].t || experts[i].type_idx < 0) return MMT_ERR_ARG;
  p.tab = {};
  for (int i = 0; i < M; ++i) { p.tab.ind[i] = experts[i].ind; p.tab.t[i] = experts[i].t; p.tab.type_idx[i] = experts[i].type_idx; }
  p.B = B; p.M = M; p.T = T; p.S = 1 + M * (T + 1); p.pack = pack; p.max_pos = max_pos;
  p.counts = counts; p.cu = cu_seqlens; p.n_rows = n_rows_dev; p.slot = slot; p.row_index = row_index; p.type_ids = type_ids;
  p.pos_ids = pos_ids; p.mask_bias = mask_bias; p.agg_row = agg_row; p.seed_bump = seed_bump; p.src = *src;
  return 0;
}

static inline int video_cast_args(VideoCastArgs& c, const MmtExpertIO* experts, int M, int B, int T, const MmtVideoSrc* src) {
  if (!experts || M <= 0 || M > MMT_MAX_EXPERTS) return MMT_ERR_ARG;
  if (int e = check_src(src)) return e;
  for (int i = 0; i < M; ++i)
    if (!experts[i].feat || !experts[i].maxpool || !experts[i].x || (experts[i].Dpad & 3) ||
        experts[i].rows_pad < B * (T + 1))
      return MMT_ERR_ARG;
  c.tab = {};
  for (int i = 0; i < M; ++i) {
    c.tab.feat[i] = experts[i].feat; c.tab.maxpool[i] = experts[i].maxpool; c.tab.x[i] = experts[i].x;
    c.tab.D[i] = experts[i].D; c.tab.Dpad[i] = experts[i].Dpad;
  }
  c.B = B; c.T = T; c.src = *src;
  return 0;
}
