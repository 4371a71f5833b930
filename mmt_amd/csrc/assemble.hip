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
#include "video_front.h"

// (ExpertTable, the plan and the cast live in video_front.h as block-level device functions: the text heads' launches can
// carry them as extra blocks, texthead2.hip)
__global__ __launch_bounds__(256) void video_plan_kernel(VideoPlanArgs p) {
  extern __shared__ float plan_ind_s[];  // [M][T] validity flags of THIS sample
  video_plan_block(p, (int)blockIdx.x, (int)threadIdx.x, plan_ind_s);
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

__global__ __launch_bounds__(256) void cast_kernel(VideoCastArgs c) {
  video_cast_block(c, (int)blockIdx.y, (int)blockIdx.x, (int)gridDim.x, (int)threadIdx.x, 256);
}

// One wave per LIVE token row: CLS -> zero feature (model.py:502-503); otherwise normalise the ReduceDim output row
// Y_e[src_row] (F.normalize, eps 1e-12, model.py:724) into the token's feature row.  BWD: gradient of the token row ->
// gradient of Y_e[src_row] (bf16).  Every live source row belongs to exactly one live token, so dY needs no zero fill.
template <bool BWD>
__global__ __launch_bounds__(256) void scatter_kernel(ExpertTable tab, int M, int T, int S, int d, int rows,
                                                      const int32_t* __restrict__ n_rows_dev,
                                                      const int32_t* __restrict__ row_index, MmtVideoSrc src,
                                                      float* __restrict__ feat, const float* __restrict__ dfeat) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nrows = n_rows_dev ? min(*n_rows_dev, rows) : rows;
  const int nch = d >> 8;
  for (int row = blockIdx.x * 4 + wave; row < nrows; row += gridDim.x * 4) {
    const int s = row_index[row] % S;
    if (s == 0) {
      if constexpr (!BWD)
        for (int c = lane * 4; c < d; c += 256) *(f32x4*)(feat + (int64_t)row * d + c) = (f32x4){0.f, 0.f, 0.f, 0.f};
      continue;
    }
    const MmtExpertIO e = tab.e[(s - 1) / (T + 1)];
    const int64_t r = src.src_row[row];
    f32x4 y[4];
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c)
      if (c < nch) {
        y[c] = *(const f32x4*)(e.y + r * d + c * 256 + lane * 4);
        if (e.n_part > 0) y[c] += *(const f32x4*)(e.y_part[0] + r * d + c * 256 + lane * 4);
        if (e.n_part > 1) y[c] += *(const f32x4*)(e.y_part[1] + r * d + c * 256 + lane * 4);
        ss += y[c][0] * y[c][0] + y[c][1] * y[c][1] + y[c][2] * y[c][2] + y[c][3] * y[c][3];
      }
    const float nrm = sqrtf(wave_sum(ss));
    const float inv = 1.0f / fmaxf(nrm, 1e-12f);
    if constexpr (!BWD) {
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (c < nch) *(f32x4*)(feat + (int64_t)row * d + c * 256 + lane * 4) = y[c] * inv;
    } else {
      f32x4 g[4];
      float dot = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (c < nch) {
          g[c] = *(const f32x4*)(dfeat + (int64_t)row * d + c * 256 + lane * 4);
          dot += g[c][0] * y[c][0] + g[c][1] * y[c][1] + g[c][2] * y[c][2] + g[c][3] * y[c][3];
        }
      // d/dy [y / max(|y|, eps)] : (g - yhat (yhat.g)) / |y| above eps, g / eps below
      const float proj = nrm > 1e-12f ? wave_sum(dot) * inv * inv : 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (c < nch) {
          f32x4 dyv = (g[c] - y[c] * proj) * inv;
          u32x2 o = {pack_bf2(dyv[0], dyv[1]), pack_bf2(dyv[2], dyv[3])};
          *(u32x2*)((bf16_t*)e.dy + r * d + c * 256 + lane * 4) = o;
        }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// ---- text-side token plan ------------------------------------------------------------------------------------------
// The reference pads every caption to max_text_words and runs the text tower on all of them (model/model.py:353-376).
// Only the [CLS] row of the last layer is read (post_agg 'cls', :378-379) and padded tokens are masked as keys in every
// layer, so -- exactly as on the video side -- dropping them changes nothing: keep the tokens with attention_mask != 0,
// in order, sample after sample.  row_index keeps the dense coordinate b*W + t (dropout RNG), cls_rows[b] = first kept
// row of sample b.  Token 0 ([CLS]) of every sample is ALWAYS kept, whatever its mask says (the reference's tokenisation
// always marks it valid, base_dataset.py:63-68 after :336-344): a caption with an all-zero mask then still owns its CLS row
// instead of reading the next sample's.
__global__ __launch_bounds__(256) void text_count_kernel(const int64_t* __restrict__ mask, int W, int32_t* __restrict__ counts) {
  __shared__ int red[4];
  const int b = blockIdx.x;
  int c = 0;
  for (int t = threadIdx.x; t < W; t += 256) c += (mask[(int64_t)b * W + t] != 0) || t == 0;
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
    const bool keep = t < W && (mask[src] != 0 || t == 0);
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
                              int32_t* agg_row, uint32_t* seed_bump, const MmtVideoSrc* src, void* stream) {
  VideoPlanArgs p;
  if (int e = video_plan_args(p, experts, M, B, T, pack, max_pos, counts, cu_seqlens, n_rows_dev, slot, row_index, type_ids,
                              pos_ids, mask_bias, agg_row, seed_bump, src))
    return e;
  hipLaunchKernelGGL(video_plan_kernel, dim3(B), dim3(256), (size_t)M * T * sizeof(float), (hipStream_t)stream, p);
  return (int)hipGetLastError();
}

extern "C" int mmt_video_cast(const MmtExpertIO* experts, int M, int B, int T, const MmtVideoSrc* src, void* stream) {
  VideoCastArgs c;
  if (int e = video_cast_args(c, experts, M, B, T, src)) return e;
  hipLaunchKernelGGL(cast_kernel, dim3(256, M), dim3(256), 0, (hipStream_t)stream, c);
  return (int)hipGetLastError();
}

extern "C" int mmt_video_scatter(const MmtExpertIO* experts, int M, int B, int T, int d, const int32_t* n_rows_dev,
                                 const int32_t* row_index, const MmtVideoSrc* src, float* features, void* stream) {
  ExpertTable tab;
  if (int e = make_table(experts, M, tab)) return e;
  if (int e = check_src(src)) return e;
  if (!row_index || !features || d % 256 || d > 1024) return MMT_ERR_ARG;
  const int S = 1 + M * (T + 1), rows = B * S;
  hipLaunchKernelGGL(scatter_kernel<false>, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, tab, M, T, S, d, rows,
                     n_rows_dev, row_index, *src, features, nullptr);
  return (int)hipGetLastError();
}

extern "C" int mmt_video_scatter_bwd(const MmtExpertIO* experts, int M, int B, int T, int d, const int32_t* n_rows_dev,
                                     const int32_t* row_index, const MmtVideoSrc* src, const float* dfeatures,
                                     void* stream) {
  ExpertTable tab;
  if (int e = make_table(experts, M, tab)) return e;
  if (int e = check_src(src)) return e;
  if (!row_index || !dfeatures || d % 256 || d > 1024) return MMT_ERR_ARG;
  for (int i = 0; i < M; ++i)
    if (!experts[i].dy) return MMT_ERR_ARG;
  const int S = 1 + M * (T + 1), rows = B * S;
  hipLaunchKernelGGL(scatter_kernel<true>, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, tab, M, T, S, d, rows,
                     n_rows_dev, row_index, *src, nullptr, dfeatures);
  return (int)hipGetLastError();
}
