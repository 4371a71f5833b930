// Video-token pipeline in front of the video-BERT (gfx950):  model/model.py:426-437 (ReduceDim) and
// :485-567 (token assembly), re-designed for the device:
//
//   plan    : per sample, decide which of the S = 1 + M*(1+T) token slots are live and where they go.
//             Dense mode keeps all S slots (row = b*S + s).  Packed mode keeps CLS, every AGG token and
//             the FEA tokens whose features_ind is 1 -- padded FEA tokens are masked as keys in every
//             layer and never read out, so dropping them changes no consumed value
//             or gradient; rows then shrink from B*S to sum_b S_b and every later kernel reads the live
//             count from device memory (no host sync).
//   cast    : expert features fp32 [B,T,D] + maxpool [B,D] -> one bf16 matrix per expert, K zero-padded.
//   (GEMM)  : Y = X . W^T + b per expert (gemm.hip, MMT_EPI_BIAS_F32)
//   scatter : L2-normalise each Y row (F.normalize, eps 1e-12) straight into its token slot.
//   scatter_bwd : gradient of the slot -> gradient of Y (bf16, zero for dropped tokens).
#include "mmt_common.h"
#include "../../include/mmt_hip.h"

struct ExpertTable { MmtExpertIO e[MMT_MAX_EXPERTS]; };

__device__ __forceinline__ void decode_slot(int s, int T, int& expert, int& j) {
  expert = (s - 1) / (T + 1);
  j = (s - 1) % (T + 1);  // 0 = AGG, 1..T = FEA t = j-1
}

// grid = B blocks of 256 threads.  Phase 0 (counts) / phase 1 (fill, after cu has been scanned).
__global__ __launch_bounds__(256) void plan_kernel(ExpertTable tab, int B, int M, int T, int S, int pack, int max_pos,
                                                   int phase, int32_t* __restrict__ counts, int32_t* __restrict__ cu,
                                                   int32_t* __restrict__ n_rows, int32_t* __restrict__ slot, int32_t* __restrict__ row_index,
                                                   int32_t* __restrict__ type_ids, int32_t* __restrict__ pos_ids,
                                                   float* __restrict__ mask_bias, int32_t* __restrict__ agg_row,
                                                   uint32_t* __restrict__ seed_bump) {
  __shared__ int scan[256];
  __shared__ int carry;
  const int b = blockIdx.x, tid = threadIdx.x;
  const bool one_launch = phase == 2;
  int base = 0;
  if (phase == 2 && b == 0 && tid == 0 && seed_bump) *seed_bump += 1u;  // per-step dropout seed (one launch less)
  if (phase == 2) {  // ONE launch: the block counts the live tokens of the samples in front of it itself
    int part = 0;
    if (pack) {  // FEA slots of the samples 0..b-1 (CLS + M AGG tokens are always live): ind is [B, T] per expert
      const int bT = b * T;
#pragma unroll
      for (int ex = 0; ex < MMT_MAX_EXPERTS; ++ex) {  // uniform, unrolled: every expert's flag loads are in flight together
        if (ex < M) {
          const float* __restrict__ ind_e = tab.e[ex].ind;
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int i = tid + 256 * u;
            if (i < bT) part += ind_e[i] != 0.f;
          }
          for (int i = tid + 1024; i < bT; i += 256) part += ind_e[i] != 0.f;
        }
      }
    }
    scan[tid] = part;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (tid < o) scan[tid] += scan[tid + o];
      __syncthreads();
    }
    base = pack ? b * (1 + M) + scan[0] : b * S;
    __syncthreads();
    phase = 1;  // fill below; cu / n_rows are written once this sample's own count is known
  } else if (phase == 1) {  // exclusive prefix of the per-sample counts (phase 0), computed by the block itself: no scan launch
    int part = 0;
    for (int i = tid; i < b; i += 256) part += counts[i];
    scan[tid] = part;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (tid < o) scan[tid] += scan[tid + o];
      __syncthreads();
    }
    base = scan[0];
    __syncthreads();
    if (tid == 0) {
      cu[b] = base;
      if (b == B - 1) { cu[B] = base + counts[b]; *n_rows = base + counts[b]; }
    }
  }
  if (tid == 0) carry = 0;
  __syncthreads();
  for (int s0 = 0; s0 < S; s0 += 256) {
    const int s = s0 + tid;
    int live = 0, expert = 0, j = 0;
    float ind = 1.f;
    if (s < S) {
      if (s == 0) live = 1;
      else {
        decode_slot(s, T, expert, j);
        if (j == 0) live = 1;
        else { ind = tab.e[expert].ind[(int64_t)b * T + (j - 1)]; live = pack ? (ind != 0.f) : 1; }
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
    if (phase == 1 && s < S) {
      const int row = live ? base + before : -1;
      slot[(int64_t)b * S + s] = row;
      if (live) {
        row_index[row] = b * S + s;
        int type = 0, pos = 0;
        float mask = 1.f;
        if (s > 0) {
          type = tab.e[expert].type_idx;
          if (j == 0) {
            float mx = 0.f;  // th.max(features_ind, 1)  model.py:330
            for (int t = 0; t < T; ++t) mx = fmaxf(mx, tab.e[expert].ind[(int64_t)b * T + t]);
            mask = mx;
            agg_row[b * M + expert] = row;
          } else {
            mask = ind;
            float tv = tab.e[expert].t[(int64_t)b * T + (j - 1)];
            tv = fminf(fmaxf(tv, 0.f), (float)max_pos);  // clamp_ then .long()  model.py:516-520
            pos = (int)tv;
          }
        }
        type_ids[row] = type;
        pos_ids[row] = pos;
        mask_bias[row] = (1.0f - mask) * -10000.0f;  // bert.py:395
      }
    }
    __syncthreads();
    if (tid == 255) carry += scan[255];
    __syncthreads();
  }
  if (phase == 0 && tid == 0) counts[b] = carry;
  if (one_launch && tid == 0) {
    counts[b] = carry;
    cu[b] = base;
    if (b == B - 1) { cu[B] = base + carry; *n_rows = base + carry; }
  }
}

__global__ void scan_counts_kernel(const int32_t* __restrict__ counts, int B, int32_t* __restrict__ cu,
                                   int32_t* __restrict__ n_rows) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    int acc = 0;
    for (int b = 0; b < B; ++b) { cu[b] = acc; acc += counts[b]; }
    cu[B] = acc;
    *n_rows = acc;
  }
}

// X_e[r][c]: r < B*T -> features[b = r/T][t = r%T][c];  B*T <= r < B*(T+1) -> maxpool[r - B*T][c]; zero padding.
__global__ __launch_bounds__(256) void cast_kernel(ExpertTable tab, int B, int T) {
  const MmtExpertIO e = tab.e[blockIdx.y];
  const int rows = B * (T + 1);
  const int64_t n = (int64_t)e.rows_pad * (e.Dpad / 4);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / (e.Dpad / 4)), c = (int)(i % (e.Dpad / 4)) * 4;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (r < rows) {
      const float* src = r < B * T ? e.feat + (int64_t)r * e.D : e.maxpool + (int64_t)(r - B * T) * e.D;
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (c + k < e.D) v[k] = src[c + k];
    }
    u32x2 o = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
    *(u32x2*)((bf16_t*)e.x + (int64_t)r * e.Dpad + c) = o;
  }
}

// one wave per source row (expert e, r): normalise Y_e[r] into features[slot]
template <bool BWD>
__global__ __launch_bounds__(256) void scatter_kernel(ExpertTable tab, int B, int M, int T, int S, int d,
                                                      const int32_t* __restrict__ slot, float* __restrict__ feat,
                                                      const float* __restrict__ dfeat) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int per = B * (T + 1);
  const int total = M * per + B;  // + CLS rows
  const int nch = d >> 8;
  for (int w = blockIdx.x * 4 + wave; w < total; w += gridDim.x * 4) {
    if (w >= M * per) {  // CLS token: zero feature (model.py:502-503); no gradient to propagate
      if constexpr (!BWD) {
        const int b = w - M * per;
        const int dst = slot[(int64_t)b * S];
        for (int c = lane * 4; c < d; c += 256) *(f32x4*)(feat + (int64_t)dst * d + c) = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
      continue;
    }
    const int ex = w / per, r = w % per;
    const MmtExpertIO e = tab.e[ex];
    const int b = r < B * T ? r / T : r - B * T;
    const int s = 1 + ex * (T + 1) + (r < B * T ? 1 + r % T : 0);
    const int dst = slot[(int64_t)b * S + s];
    if constexpr (!BWD) {
      if (dst < 0) continue;
    } else {
      if (dst < 0) {
        for (int c = lane * 4; c < d; c += 256) *(u32x2*)((bf16_t*)e.dy + (int64_t)r * d + c) = (u32x2){0u, 0u};
        continue;
      }
    }
    f32x4 y[4];
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c)
      if (c < nch) {
        y[c] = *(const f32x4*)(e.y + (int64_t)r * d + c * 256 + lane * 4);
        ss += y[c][0] * y[c][0] + y[c][1] * y[c][1] + y[c][2] * y[c][2] + y[c][3] * y[c][3];
      }
    const float nrm = sqrtf(wave_sum(ss));
    const float inv = 1.0f / fmaxf(nrm, 1e-12f);
    if constexpr (!BWD) {
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (c < nch) *(f32x4*)(feat + (int64_t)dst * d + c * 256 + lane * 4) = y[c] * inv;
    } else {
      f32x4 g[4];
      float dot = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (c < nch) {
          g[c] = *(const f32x4*)(dfeat + (int64_t)dst * d + c * 256 + lane * 4);
          dot += g[c][0] * y[c][0] + g[c][1] * y[c][1] + g[c][2] * y[c][2] + g[c][3] * y[c][3];
        }
      // d/dy [y / max(|y|, eps)] : (g - yhat (yhat.g)) / |y| above eps, g / eps below
      const float proj = nrm > 1e-12f ? wave_sum(dot) * inv * inv : 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (c < nch) {
          f32x4 dyv = (g[c] - y[c] * proj) * inv;
          u32x2 o = {pack_bf2(dyv[0], dyv[1]), pack_bf2(dyv[2], dyv[3])};
          *(u32x2*)((bf16_t*)e.dy + (int64_t)r * d + c * 256 + lane * 4) = o;
        }
    }
  }
}

// ------------------------------------------------------------------------------------------------
static int make_table(const MmtExpertIO* experts, int M, ExpertTable& tab) {
  if (!experts || M <= 0 || M > MMT_MAX_EXPERTS) return MMT_ERR_ARG;
  for (int i = 0; i < M; ++i) tab.e[i] = experts[i];
  return 0;
}

// ---- text-side token plan ------------------------------------------------------------------------------------------
// The reference pads every caption to max_text_words and runs the text tower on all of them (model/model.py:353-376).
// Only the [CLS] row of the last layer is read (post_agg 'cls', :378-379) and padded tokens are masked as keys in every
// layer, so -- exactly as on the video side -- dropping them changes nothing: keep the tokens with attention_mask != 0,
// in order, sample after sample.  row_index keeps the dense coordinate b*W + t (dropout RNG), cls_rows[b] = first kept
// row of sample b (the caller guarantees attention_mask[:, 0] == 1).
__global__ __launch_bounds__(256) void text_count_kernel(const int64_t* __restrict__ mask, int W, int32_t* __restrict__ counts) {
  __shared__ int red[4];
  const int b = blockIdx.x;
  int c = 0;
  for (int t = threadIdx.x; t < W; t += 256) c += mask[(int64_t)b * W + t] != 0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) counts[b] = red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void text_pack_kernel(const int64_t* __restrict__ ids_in, const int64_t* __restrict__ types_in,
                                                        const int64_t* __restrict__ pos_in, const int64_t* __restrict__ mask,
                                                        int W, const int32_t* __restrict__ cu, int32_t* __restrict__ ids,
                                                        int32_t* __restrict__ types, int32_t* __restrict__ pos,
                                                        int32_t* __restrict__ row_index, int32_t* __restrict__ cls_rows) {
  __shared__ int wave_cnt[4];
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int base = cu[b];
  int carry = 0;
  for (int t0 = 0; t0 < W; t0 += 256) {
    const int t = t0 + threadIdx.x;
    const int64_t src = (int64_t)b * W + t;
    const bool keep = t < W && mask[src] != 0;
    const unsigned long long bal = __ballot(keep);
    const int before = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wave_cnt[wave] = __popcll(bal);
    __syncthreads();
    int woff = 0, total = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) { if (i < wave) woff += wave_cnt[i]; total += wave_cnt[i]; }
    if (keep) {
      const int r = base + carry + woff + before;
      ids[r] = (int32_t)ids_in[src];
      types[r] = types_in ? (int32_t)types_in[src] : 0;
      pos[r] = pos_in ? (int32_t)pos_in[src] : t;
      row_index[r] = (int32_t)src;
    }
    carry += total;
    __syncthreads();
  }
  if (threadIdx.x == 0) cls_rows[b] = base;
}

// input_ids / token_type_ids (nullable: 0) / position_ids (nullable: 0..W-1) / attention_mask: contiguous int64 [B, W].
// Outputs (int32): counts [B], cu_seqlens [B+1], n_rows_dev [1], ids / types / pos / row_index [>= B*W] (only the first
// *n_rows_dev entries are written), cls_rows [B].
extern "C" int mmt_text_plan(const int64_t* input_ids, const int64_t* token_type_ids, const int64_t* position_ids,
                             const int64_t* attention_mask, int B, int W, int32_t* counts, int32_t* cu_seqlens,
                             int32_t* n_rows_dev, int32_t* ids, int32_t* types, int32_t* pos, int32_t* row_index,
                             int32_t* cls_rows, void* stream) {
  if (!input_ids || !attention_mask || !counts || !cu_seqlens || !n_rows_dev || !ids || !types || !pos || !row_index ||
      !cls_rows || B <= 0 || W <= 0)
    return MMT_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(text_count_kernel, dim3(B), dim3(256), 0, s, attention_mask, W, counts);
  hipLaunchKernelGGL(scan_counts_kernel, dim3(1), dim3(64), 0, s, counts, B, cu_seqlens, n_rows_dev);
  hipLaunchKernelGGL(text_pack_kernel, dim3(B), dim3(256), 0, s, input_ids, token_type_ids, position_ids, attention_mask, W,
                     cu_seqlens, ids, types, pos, row_index, cls_rows);
  return (int)hipGetLastError();
}

extern "C" int mmt_video_plan(const MmtExpertIO* experts, int M, int B, int T, int pack, int max_pos,
                              int32_t* counts, int32_t* cu_seqlens, int32_t* n_rows_dev, int32_t* slot,
                              int32_t* row_index, int32_t* type_ids, int32_t* pos_ids, float* mask_bias,
                              int32_t* agg_row, uint32_t* seed_bump, void* stream) {
  ExpertTable tab;
  if (int e = make_table(experts, M, tab)) return e;
  if (!counts || !cu_seqlens || !n_rows_dev || !slot || !row_index || !type_ids || !pos_ids || !mask_bias || !agg_row)
    return MMT_ERR_ARG;
  const int S = 1 + M * (T + 1);
  // one launch: every block counts the live tokens of the samples in front of it itself (<= B*M*T flags, trivial)
  hipLaunchKernelGGL(plan_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, tab, B, M, T, S, pack, max_pos, 2, counts,
                     cu_seqlens, n_rows_dev, slot, row_index, type_ids, pos_ids, mask_bias, agg_row, seed_bump);
  return (int)hipGetLastError();
}

extern "C" int mmt_video_cast(const MmtExpertIO* experts, int M, int B, int T, void* stream) {
  ExpertTable tab;
  if (int e = make_table(experts, M, tab)) return e;
  for (int i = 0; i < M; ++i)
    if (!experts[i].feat || !experts[i].maxpool || !experts[i].x || (experts[i].Dpad & 3) ||
        experts[i].rows_pad < B * (T + 1))
      return MMT_ERR_ARG;
  hipLaunchKernelGGL(cast_kernel, dim3(256, M), dim3(256), 0, (hipStream_t)stream, tab, B, T);
  return (int)hipGetLastError();
}

extern "C" int mmt_video_scatter(const MmtExpertIO* experts, int M, int B, int T, int d, const int32_t* slot,
                                 float* features, void* stream) {
  ExpertTable tab;
  if (int e = make_table(experts, M, tab)) return e;
  if (!slot || !features || d % 256 || d > 1024) return MMT_ERR_ARG;
  const int total = M * B * (T + 1) + B;
  hipLaunchKernelGGL(scatter_kernel<false>, dim3((total + 3) / 4), dim3(256), 0, (hipStream_t)stream, tab, B, M, T,
                     1 + M * (T + 1), d, slot, features, nullptr);
  return (int)hipGetLastError();
}

extern "C" int mmt_video_scatter_bwd(const MmtExpertIO* experts, int M, int B, int T, int d, const int32_t* slot,
                                     const float* dfeatures, void* stream) {
  ExpertTable tab;
  if (int e = make_table(experts, M, tab)) return e;
  if (!slot || !dfeatures || d % 256 || d > 1024) return MMT_ERR_ARG;
  for (int i = 0; i < M; ++i)
    if (!experts[i].dy) return MMT_ERR_ARG;
  const int total = M * B * (T + 1) + B;
  hipLaunchKernelGGL(scatter_kernel<true>, dim3((total + 3) / 4), dim3(256), 0, (hipStream_t)stream, tab, B, M, T,
                     1 + M * (T + 1), d, slot, nullptr, dfeatures);
  return (int)hipGetLastError();
}
