// Evaluation path at scale (reference: trainer/trainer.py:372-447, model/metric.py:26-243).
//
// The reference gathers every embedding on the CPU, builds the N_text x N_video similarity there
// (sharded_cross_view_inner_product, model/model.py:789-837) and ranks it with numpy.  Here the matrix stays in HBM:
//
//   mmt_sims_eval       : sims[t][v] = sum_m tw[t][m] vw[v][m] <T_m[t], V_m[v]> / sum_m tw[t][m] vw[v][m]
//                         as ONE GEMM with K = M*d on the exact-fp32 matrix cores (weights folded into the operands),
//                         followed by the normaliser (zero -> 1e-5, model.py:816).  No [NT][NV][M] tensor.
//   mmt_retrieval_ranks : tie-averaged rank of the ground truth per text query (t2v, metric.py:90-121) and the best
//                         such rank among a video's own captions (v2t, metric.py:153-243), both on device -- only
//                         O(n) floats ever leave the GPU.
#include "mmt_common.h"
#include "../../include/mmt_hip.h"

// out[r][m*d + c] = w[r][m] * x[r][m][c]
__global__ __launch_bounds__(256) void scale_rows_kernel(const float* __restrict__ x, const float* __restrict__ w, int64_t n4,
                                                         int M, int d, float* __restrict__ out) {
  const int d4 = d >> 2;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t rm = i / d4;  // r*M + m
    ((f32x4*)out)[i] = ((const f32x4*)x)[i] * w[rm];
  }
}

// sims[t][v] /= sum_m tw[t][m] vw[v][m]   (0 -> 1e-5)
__global__ __launch_bounds__(256) void sims_normalise_kernel(float* __restrict__ sims, const float* __restrict__ tw,
                                                             const float* __restrict__ vw, int NT, int NV, int M) {
  const int t = blockIdx.y;
  const int v = blockIdx.x * 256 + threadIdx.x;
  if (v >= NV) return;
  float den = 0.f;
  for (int m = 0; m < M; ++m) den += tw[(int64_t)t * M + m] * vw[(int64_t)v * M + m];
  if (den == 0.f) den = 1e-5f;
  sims[(int64_t)t * NV + v] /= den;
}

extern "C" int64_t mmt_sims_eval_workspace_floats(int NT, int NV, int M, int d) { return (int64_t)(NT + NV) * M * d; }

extern "C" int mmt_sims_eval(const float* txt, const float* vid, const float* tw, const float* vw, int NT, int NV, int M,
                             int d, float* ws, float* sims, void* stream) {
  if (!txt || !vid || !tw || !vw || !ws || !sims || NT <= 0 || NV <= 0 || M <= 0 || M > MMT_MAX_EXPERTS || d <= 0 || (d & 3))
    return MMT_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  float* ts = ws;
  float* vs = ws + (int64_t)NT * M * d;
  const int64_t nt4 = (int64_t)NT * M * d / 4, nv4 = (int64_t)NV * M * d / 4;
  hipLaunchKernelGGL(scale_rows_kernel, dim3((int)((nt4 + 255) / 256 < 4096 ? (nt4 + 255) / 256 : 4096)), dim3(256), 0, s,
                     txt, tw, nt4, M, d, ts);
  hipLaunchKernelGGL(scale_rows_kernel, dim3((int)((nv4 + 255) / 256 < 4096 ? (nv4 + 255) / 256 : 4096)), dim3(256), 0, s,
                     vid, vw, nv4, M, d, vs);
  MmtSgemm g = {};
  g.batch = 1; g.M = NT; g.N = NV; g.K = M * d;
  g.sai = (int64_t)M * d; g.sak = 1; g.sbj = (int64_t)M * d; g.sbk = 1; g.ldc = NV;
  g.A[0] = ts; g.B[0] = vs; g.C[0] = sims;
  int rc = mmt_sgemm_batched(&g, stream);
  if (rc) return rc;
  hipLaunchKernelGGL(sims_normalise_kernel, dim3((NV + 255) / 256, NT), dim3(256), 0, s, sims, tw, vw, NT, NV, M);
  return (int)hipGetLastError();
}

// ---- ranks -------------------------------------------------------------------------------------------------
// t2v: one block per text query q (ground truth video q / cpv): rank = #(s[q][:] > gt) + (#(s[q][:] == gt) - 1) / 2.
__global__ __launch_bounds__(256) void t2v_rank_kernel(const float* __restrict__ sims, int NQ, int NV, int cpv,
                                                       float* __restrict__ rank) {
  __shared__ int red[2][4];
  const int q = blockIdx.x;
  const float* row = sims + (int64_t)q * NV;
  const float gt = row[q / cpv];
  int gtr = 0, eq = 0;
  for (int v = threadIdx.x; v < NV; v += 256) {
    const float x = row[v];
    gtr += x > gt;
    eq += x == gt;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { gtr += __shfl_xor(gtr, o, 64); eq += __shfl_xor(eq, o, 64); }
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = gtr; red[1][threadIdx.x >> 6] = eq; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int g = red[0][0] + red[0][1] + red[0][2] + red[0][3], e = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    rank[q] = (float)g + ((float)e - 1.0f) * 0.5f;
  }
}

// v2t: block (64 videos, caption slot k): lane = video i, its caption j = i*cpv + k; counts over all unmasked captions c
// of s[c][i] > s[j][i] (coalesced across the 64 columns).  all_rank[j] = +inf for masked captions.
__global__ __launch_bounds__(256) void v2t_rank_kernel(const float* __restrict__ sims, const uint8_t* __restrict__ qmask, int NC,
                                                       int NV, int cpv, float* __restrict__ all_rank) {
  __shared__ int red[2][4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + lane, k = blockIdx.y;
  const bool ok = i < NV;
  const int j = ok ? i * cpv + k : 0;
  const bool valid = ok && (!qmask || qmask[j]);
  const float gt = valid ? sims[(int64_t)j * NV + i] : 0.f;
  int gtr = 0, eq = 0;
  if (ok) {
    for (int c = wave; c < NC; c += 4) {
      if (qmask && !qmask[c]) continue;  // wave-uniform
      const float x = sims[(int64_t)c * NV + i];
      gtr += x > gt;
      eq += x == gt;
    }
  }
  red[0][wave][lane] = gtr; red[1][wave][lane] = eq;
  __syncthreads();
  if (wave == 0 && ok) {
    const int g = red[0][0][lane] + red[0][1][lane] + red[0][2][lane] + red[0][3][lane];
    const int e = red[1][0][lane] + red[1][1][lane] + red[1][2][lane] + red[1][3][lane];
    all_rank[j] = valid ? (float)g + ((float)e - 1.0f) * 0.5f : INFINITY;
  }
}

__global__ void v2t_best_kernel(const float* __restrict__ all_rank, int NV, int cpv, float* __restrict__ best) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= NV) return;
  float b = INFINITY;
  for (int k = 0; k < cpv; ++k) b = fminf(b, all_rank[(int64_t)i * cpv + k]);
  best[i] = b;
}

// sims [NQ = NV*cpv][NV] fp32 (rows = text queries, caption index fastest within a video); qmask (nullable) uint8 [NQ]:
// 1 = real caption.  t2v_rank [NQ]; v2t_rank [NV]; scratch [NQ] floats.
extern "C" int mmt_retrieval_ranks(const float* sims, const uint8_t* qmask, int NQ, int NV, float* t2v_rank,
                                   float* v2t_rank, float* scratch, void* stream) {
  if (!sims || !t2v_rank || !v2t_rank || !scratch || NQ <= 0 || NV <= 0 || NQ % NV) return MMT_ERR_ARG;
  const int cpv = NQ / NV;
  if (cpv > 65535) return MMT_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(t2v_rank_kernel, dim3(NQ), dim3(256), 0, s, sims, NQ, NV, cpv, t2v_rank);
  hipLaunchKernelGGL(v2t_rank_kernel, dim3((NV + 63) / 64, cpv), dim3(256), 0, s, sims, qmask, NQ, NV, cpv, scratch);
  hipLaunchKernelGGL(v2t_best_kernel, dim3((NV + 255) / 256), dim3(256), 0, s, scratch, NV, cpv, v2t_rank);
  return (int)hipGetLastError();
}
