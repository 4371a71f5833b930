// Memory-bound utility kernels (gfx950): weight packing fp32 -> bf16 (+ transposed copy for the
// input-gradient GEMMs), bf16 column sums (bias gradients), padded slab reduction, fused Adam.
#include "mmt_common.h"
#include "../../include/mmt_hip.h"
#include "adam_unit.h"

// ---- weight packing ----------------------------------------------------------------------------
// One launch packs up to MMT_PACK_MAX matrices: dst[r][c] = bf16(src[r][c]) for c < cols, 0 for
// cols <= c < dst_ld; optional dst_t[c][r] (ld = dst_t_ld, rows padded with zeros up to dst_t_rows).
struct PackTable { MmtPackItem item[MMT_PACK_MAX]; int n; };

__global__ __launch_bounds__(256) void pack_weights_kernel(PackTable tab) {
  __shared__ float tile[64][65];
  const MmtPackItem it = tab.item[blockIdx.y];
  const int tiles_c = (it.dst_ld + 63) / 64, tiles_r = (it.rows + 63) / 64;
  if ((int)blockIdx.x >= tiles_c * tiles_r) return;
  const int tr = blockIdx.x / tiles_c, tc = blockIdx.x % tiles_c;
  const int r0 = tr * 64, c0 = tc * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 64 x 4
  for (int i = ty; i < 64; i += 4) {
    const int r = r0 + i, c = c0 + tx;
    float v = 0.f;
    if (r < it.rows && c < it.cols) v = it.src[(int64_t)r * it.cols + c];
    tile[i][tx] = v;
    if (r < it.rows && c < it.dst_ld) ((bf16_t*)it.dst)[(int64_t)r * it.dst_ld + c] = f2bf(v);
  }
  if (!it.dst_t) return;
  __syncthreads();
  for (int i = ty; i < 64; i += 4) {  // transposed: row = source column, col = source row
    const int c = c0 + i, r = r0 + tx;
    if (c < it.dst_t_rows && r < it.dst_t_ld) ((bf16_t*)it.dst_t)[(int64_t)c * it.dst_t_ld + r] = f2bf(tile[tx][i]);
  }
}

extern "C" int mmt_pack_weights(const MmtPackItem* items, int n, void* stream) {
  if (!items || n <= 0) return MMT_ERR_ARG;
  for (int base = 0; base < n; base += MMT_PACK_MAX) {
    PackTable tab;
    tab.n = n - base < MMT_PACK_MAX ? n - base : MMT_PACK_MAX;
    int max_tiles = 0;
    for (int i = 0; i < tab.n; ++i) {
      const MmtPackItem& it = items[base + i];
      if (!it.src || !it.dst || it.rows <= 0 || it.cols <= 0 || it.dst_ld < it.cols) return MMT_ERR_ARG;
      if (it.dst_t && (it.dst_t_ld < it.rows || it.dst_t_rows < it.cols)) return MMT_ERR_ARG;
      tab.item[i] = it;
      const int t = ((it.dst_ld + 63) / 64) * ((it.rows + 63) / 64);
      max_tiles = t > max_tiles ? t : max_tiles;
    }
    hipLaunchKernelGGL(pack_weights_kernel, dim3(max_tiles, tab.n), dim3(256), 0, (hipStream_t)stream, tab);
  }
  return (int)hipGetLastError();
}

// ---- bf16 column sums: partials[blk][c] = sum over the block's 32 rows -------------------------
__global__ __launch_bounds__(256) void colsum_bf16_kernel(const bf16_t* __restrict__ x, int64_t ld, int rows, int cols,
                                                          const int32_t* __restrict__ n_rows_dev,
                                                          float* __restrict__ partials) {
  const int nrows = n_rows_dev ? min(*n_rows_dev, rows) : rows;
  const int c = (blockIdx.y * 256 + threadIdx.x) * 4;
  if (c >= cols) return;
  const int r0 = blockIdx.x * 32, r1 = min(nrows, r0 + 32);
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  for (int r = r0; r < r1; ++r) {
    const u32x2 v = *(const u32x2*)(x + (int64_t)r * ld + c);
    s[0] += bf2f((bf16_t)(v[0] & 0xffff)); s[1] += bf2f((bf16_t)(v[0] >> 16));
    s[2] += bf2f((bf16_t)(v[1] & 0xffff)); s[3] += bf2f((bf16_t)(v[1] >> 16));
  }
  *(f32x4*)(partials + (int64_t)blockIdx.x * cols + c) = s;
}

extern "C" int mmt_colsum_bf16(const void* x, int64_t ld, int rows, int cols, const int32_t* n_rows_dev,
                               float* partials, void* stream) {
  if (!x || !partials || rows <= 0 || cols <= 0 || (cols & 3) || (ld & 3)) return MMT_ERR_ARG;
  hipLaunchKernelGGL(colsum_bf16_kernel, dim3((rows + 31) / 32, (cols + 1023) / 1024), dim3(256), 0,
                     (hipStream_t)stream, (const bf16_t*)x, ld, rows, cols, n_rows_dev, partials);
  return (int)hipGetLastError();
}

// ---- slab reduction with column un-padding: out[r][c] (+)= sum_s ws[s][r][c], c < cols_out ------
__global__ void reduce_slabs_2d_kernel(const float* __restrict__ ws, int splits, int rows, int cols_ws, int cols_out,
                                       float* __restrict__ out, int accumulate) {
  const int64_t n = (int64_t)rows * cols_out;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / cols_out), c = (int)(i % cols_out);
    float s = 0.f;
    for (int k = 0; k < splits; ++k) s += ws[((int64_t)k * rows + r) * cols_ws + c];
    out[i] = accumulate ? out[i] + s : s;
  }
}

extern "C" int mmt_reduce_slabs_2d(const float* ws, int splits, int rows, int cols_ws, int cols_out, float* out,
                                   int accumulate, void* stream) {
  if (!ws || !out || splits <= 0 || rows <= 0 || cols_out <= 0 || cols_out > cols_ws) return MMT_ERR_ARG;
  const int64_t n = (int64_t)rows * cols_out;
  const int grid = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
  hipLaunchKernelGGL(reduce_slabs_2d_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, ws, splits, rows,
                     cols_ws, cols_out, out, accumulate);
  return (int)hipGetLastError();
}

// ---- fused Adam over a flat fp32 buffer (torch.optim.Adam semantics, train.py:100) --------------
// The update of four elements (adam_update4) and of one 4096-element unit live in adam_unit.h, shared by the plain kernel
// (one span: a data-parallel rank's shard), the fused one (whole buffer + bf16 shadows), the queue kernel and the rider
// blocks of the GEMM launches, so that all of them round identically.
// p, m, v updated in place from g; `step_dev` holds the 1-based step count on the device (graph safe).
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, int64_t n4,
                                                   float lr, float beta1, float beta2, float eps, float weight_decay,
                                                   const int32_t* __restrict__ step_dev,
                                                   const float* __restrict__ lr_dev) {
  if (lr_dev) lr = *lr_dev;  // learning-rate schedules under graph replay: the rate lives on the device
  const float t = (float)*step_dev;
  const float bc1 = 1.0f - powf(beta1, t), bc2 = 1.0f - powf(beta2, t);
  const float step_size = lr / bc1, inv_sqrt_bc2 = 1.0f / sqrtf(bc2);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    f32x4 pv = ((f32x4*)p)[i], gv = ((const f32x4*)g)[i], mv = ((f32x4*)m)[i], vv = ((f32x4*)v)[i];
    adam_update4(pv, gv, mv, vv, beta1, beta2, eps, weight_decay, step_size, inv_sqrt_bc2);
    ((f32x4*)p)[i] = pv; ((f32x4*)m)[i] = mv; ((f32x4*)v)[i] = vv;
  }
}

extern "C" int mmt_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t count,
                             float lr, float beta1, float beta2, float eps, float weight_decay,
                             const int32_t* step_dev, const float* lr_dev, void* stream) {
  if (!params || !grads || !exp_avg || !exp_avg_sq || !step_dev || count <= 0 || (count & 3)) return MMT_ERR_ARG;
  const int64_t n4 = count / 4;
  const int grid = (int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
  hipLaunchKernelGGL(adam_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, params, grads, exp_avg,
                     exp_avg_sq, n4, lr, beta1, beta2, eps, weight_decay, step_dev, lr_dev);
  return (int)hipGetLastError();
}

// ---- fused Adam + bf16 shadow refresh ------------------------------------------------------------------------------
// The GEMMs read bf16 shadows (W, and W^T for the input-gradient GEMMs) of the fp32 master weights.  Re-packing them
// after every optimizer step was a pure re-layout pass over every GEMM weight (pack_weights_kernel: 48 us of a 1.65 ms
// step, 9 % together with Adam).  Here the optimizer writes them itself: the flat buffer is cut into SEGMENTS, each
// either a shadowed matrix [rows, cols] (worked on in 64x64 tiles: Adam update, bf16 row-major store, transposed
// bf16 store through an LDS tile) or a plain span (biases, LayerNorm, embeddings, fp32 text heads).  One launch,
// one 4096-element unit of work per block either way.  The segment table lives in device memory (built once by the
// host side); the first-block prefix of every segment travels by value so a block finds its segment without a
// dependent chain of loads.
struct AdamSegIndex { int32_t n; int32_t begin[MMT_ADAM_SEG_MAX + 1]; };
__global__ __launch_bounds__(256) void adam_fused_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                         float* __restrict__ m, float* __restrict__ v,
                                                         const MmtAdamSeg* __restrict__ segs, AdamSegIndex idx, float lr,
                                                         float beta1, float beta2, float eps, float weight_decay,
                                                         int32_t* __restrict__ step_dev,
                                                         const float* __restrict__ lr_dev, int bump_step) {
  __shared__ float tile[64][65];
  if (lr_dev) lr = *lr_dev;
  // bump_step: step_dev[0] holds the steps taken so far; this launch is step step_dev[0] + 1 and the LAST block to finish
  // (ticket in step_dev[1]) stores the new count -- every block has read the old one by then.  One launch less per step
  // than a separate increment.
  // (a plain, cacheable load: a volatile one is 1.5 M uncached reads of one word -- measured 119 -> 560 us)
  const int t_int = __builtin_amdgcn_readfirstlane(*(const int32_t*)step_dev) + (bump_step ? 1 : 0);
  struct Bump {
    int32_t* s; int t, on;
    __device__ ~Bump() {
      if (!on) return;
      __syncthreads();
      // relaxed: the only ordering needed is "every block's read of s[0] precedes the store", and a block's read has
      // returned (its value fed the update) long before its ticket; a device-scope release here would write back the
      // XCD's L2 once per block (measured: +137 us on a 119 us launch)
      if (threadIdx.x == 0 &&
          __hip_atomic_fetch_add(s + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1) {
        s[1] = 0;
        s[0] = t;
      }
    }
  } bump{step_dev, t_int, bump_step == 1};  // bump_step == 2: this launch is step count + 1 too, but a later launch of
                                           // the same step stores the new count (per-region launches, FlatAdam.step_span)
  const AdamHyper h = adam_hyper(lr, beta1, beta2, eps, weight_decay, t_int);
  int s = 0;
#pragma unroll 1
  for (int q = 1; q < idx.n; ++q)
    if ((int)blockIdx.x >= idx.begin[q]) s = q;
  const MmtAdamSeg seg = segs[s];
  const int lb = (int)blockIdx.x - idx.begin[s];
  if (!adam_unit_update(p, g, m, v, seg, lb, (int)threadIdx.x, tile, h)) return;
  __syncthreads();
  adam_unit_store_t(seg, lb, (int)threadIdx.x, tile);
}

// ---- the optimizer queue (include/mmt_hip.h, "Adam riders") ---------------------------------------------------------
// What the riders of the backward's GEMM launches left of the step: entry k of the queue is block k's; of stage s the first
// min(claimed[s], size[s]) entries were run by riders, their blocks exit at once.  The last block to finish zeroes the queue
// state for the next step and stores the new step count.  EVERY block must have read the claim counters and the step count
// by then -- also the blocks that exit at once, and those may be dispatched late -- so every block takes a ticket; 6 074
// tickets on ONE word, most of them within a few microseconds, serialise (~30 ns each: the first r06 version took 164 us
// for a launch with 37 % of its blocks working), so the tickets form a two-level tree: block k counts on word k % 64 (one
// 256-byte line each), the block that completes a word counts on the top word, the block that completes that one finishes.
__global__ __launch_bounds__(256) void adam_queue_kernel(const MmtAdamQueue* __restrict__ qd) {
  __shared__ float tile[64][65];
  __shared__ int s_last;
  const int k = (int)blockIdx.x;
  const int n_units = qd->n_units, n_stages = qd->n_stages;
  int32_t* __restrict__ st = qd->state;
  int ridden = 0;
  bool mine_taken = false;
#pragma unroll 1
  for (int s = 0; s < n_stages; ++s) {  // (uniform: scalar loads)
    const int lo = qd->stage_begin[s], size = qd->stage_begin[s + 1] - lo;
    const int c = *(const int32_t*)(st + s);
    const int taken = c < size ? c : size;
    ridden += taken;
    if (k >= lo && k - lo < taken) mine_taken = true;
  }
  const int t_int = __builtin_amdgcn_readfirstlane(*(const int32_t*)qd->step_dev) + 1;
  if (!mine_taken) {
    const int si = __builtin_amdgcn_readfirstlane(qd->unit_seg[k]), lb = __builtin_amdgcn_readfirstlane(qd->unit_blk[k]);
    const MmtAdamSeg seg = qd->segs[si];
    const AdamHyper h = adam_hyper(qd->lr_dev ? *qd->lr_dev : qd->lr, qd->beta1, qd->beta2, qd->eps, qd->weight_decay, t_int);
    if (adam_unit_update(qd->p, qd->g, qd->m, qd->v, seg, lb, (int)threadIdx.x, tile, h)) {
      __syncthreads();
      adam_unit_store_t(seg, lb, (int)threadIdx.x, tile);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int j = k & 63, expect = n_units / 64 + (j < (n_units & 63) ? 1 : 0);
    int last = 0;
    if (__hip_atomic_fetch_add(st + MMT_RIDER_SUBTICKET0 + j * 64, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == expect - 1) {
      const int words = n_units < 64 ? n_units : 64;
      last = __hip_atomic_fetch_add(st + MMT_RIDER_TICKET, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == words - 1;
    }
    s_last = last;
  }
  __syncthreads();
  if (!s_last) return;
  // (statistics, never reset: entries the riders ran over all steps, steps finished -- FlatAdam.queue_stats)
  if (threadIdx.x == 0) { st[MMT_RIDER_STAT0] += ridden; st[MMT_RIDER_STAT0 + 1] += 1; }
  for (int i = threadIdx.x; i < MMT_RIDER_STAT0; i += blockDim.x) st[i] = 0;
  for (int i = threadIdx.x; i < 64; i += blockDim.x) st[MMT_RIDER_SUBTICKET0 + i * 64] = 0;
  if (threadIdx.x == 0) qd->step_dev[0] = t_int;
}

extern "C" int mmt_adam_step_queue(const MmtAdamQueue* q_host, const MmtAdamQueue* q_dev, void* stream) {
  if (!q_host || !q_dev || q_host->n_units <= 0 || !q_host->p || !q_host->g || !q_host->m || !q_host->v || !q_host->segs ||
      !q_host->unit_seg || !q_host->unit_blk || !q_host->state || !q_host->step_dev || q_host->n_stages < 0 ||
      q_host->n_stages > MMT_RIDER_STAGES)
    return MMT_ERR_ARG;
  hipLaunchKernelGGL(adam_queue_kernel, dim3(q_host->n_units), dim3(256), 0, (hipStream_t)stream, q_dev);
  return (int)hipGetLastError();
}

// rider blocks with no hosting GEMM (tests, tools/adam_lab.py): drain entries [.., limit)
__global__ __launch_bounds__(512) void adam_rider_probe_kernel(const MmtAdamQueue* __restrict__ qd, int limit) {
  extern __shared__ __attribute__((aligned(16))) unsigned char rider_smem[];
  adam_rider_run<512>(qd, limit, (int)(gridDim.x << 16), 0, (int)blockIdx.x, (int)gridDim.x, rider_smem);
}

extern "C" int mmt_adam_rider_probe(const MmtAdamQueue* q_dev, int limit, int blocks, void* stream) {
  if (!q_dev || limit < 0 || limit > MMT_RIDER_STAGES || blocks <= 0) return MMT_ERR_ARG;
  hipLaunchKernelGGL(adam_rider_probe_kernel, dim3(blocks), dim3(512), adam_rider_lds_bytes<512>(), (hipStream_t)stream, q_dev,
                     limit);
  return (int)hipGetLastError();
}

extern "C" int mmt_adam_fused_blocks(const MmtAdamSeg* seg) {
  if (!seg) return MMT_ERR_ARG;
  if (!seg->dst) return (int)((seg->count + 4095) / 4096);
  return ((seg->rows + 63) / 64) * ((seg->cols + 63) / 64);
}

extern "C" int mmt_adam_step_fused(float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                                   const MmtAdamSeg* segs_host, const MmtAdamSeg* segs_dev, int n_segs, float lr, float beta1,
                                   float beta2, float eps, float weight_decay, int32_t* step_dev, const float* lr_dev,
                                   int bump_step, void* stream) {
  if (!params || !grads || !exp_avg || !exp_avg_sq || !segs_host || !segs_dev || !step_dev || n_segs <= 0 ||
      n_segs > MMT_ADAM_SEG_MAX)
    return MMT_ERR_ARG;
  AdamSegIndex idx;
  idx.n = n_segs;
  int blocks = 0;
  for (int i = 0; i < n_segs; ++i) {
    const MmtAdamSeg& sg = segs_host[i];
    if ((sg.offset & 3) || sg.count <= 0) return MMT_ERR_ARG;
    if (sg.dst) {
      if (sg.rows <= 0 || sg.cols <= 0 || (sg.cols & 3) || sg.dst_ld < sg.cols || (sg.dst_ld & 3) || ((uintptr_t)sg.dst & 7))
        return MMT_ERR_ARG;
      if (sg.dst_t && (sg.dst_t_ld < sg.rows || (sg.dst_t_ld & 7) || ((uintptr_t)sg.dst_t & 15))) return MMT_ERR_ALIGN;
    } else if (sg.count & 3) {
      return MMT_ERR_ARG;
    }
    idx.begin[i] = blocks;
    blocks += mmt_adam_fused_blocks(&sg);
  }
  idx.begin[n_segs] = blocks;
  hipLaunchKernelGGL(adam_fused_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, params, grads, exp_avg, exp_avg_sq,
                     segs_dev, idx, lr, beta1, beta2, eps, weight_decay, step_dev, lr_dev, bump_step);
  return (int)hipGetLastError();
}


// ---- stand-alone dropout (the MoE-logit input of the text heads, model/model.py:274) ------------------------------
// y = keep ? x * scale : 0 with the engine's counter-based RNG, so that a captured training step contains no framework
// RNG at all (a philox-based dropout makes every graph replay refill the generator's seed / offset tensors first).
// The stream key is hash(drop_key, *seed_dev) at FORWARD time; it is written to key_save so that the backward, which
// runs after the encoder has advanced the seed, re-draws the same mask (key_load).
__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n4,
                                                      unsigned drop_key, unsigned thr16, float scale,
                                                      const unsigned* __restrict__ seed_dev, unsigned* __restrict__ key_save,
                                                      const unsigned* __restrict__ key_load) {
  const unsigned key = key_load ? *key_load : eff_key(drop_key, seed_dev);
  if (key_save && blockIdx.x == 0 && threadIdx.x == 0) *key_save = key;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    f32x4 v = ((const f32x4*)x)[i];
    bool k[4];
    keep4(key, (unsigned long long)i * 4ull, thr16, k);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = k[e] ? v[e] * scale : 0.f;
    ((f32x4*)y)[i] = v;
  }
}

extern "C" int mmt_dropout_f32(const float* x, float* y, int64_t n, uint32_t drop_key, uint32_t thr16, float scale,
                               const uint32_t* seed_dev, uint32_t* key_save, const uint32_t* key_load, void* stream) {
  if (!x || !y || n <= 0 || (n & 3) || ((uintptr_t)x & 15) || ((uintptr_t)y & 15)) return MMT_ERR_ARG;
  const int64_t n4 = n / 4;
  const int grid = (int)((n4 + 255) / 256 < 1024 ? (n4 + 255) / 256 : 1024);
  hipLaunchKernelGGL(dropout_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, y, n4, drop_key, thr16, scale, seed_dev,
                     key_save, key_load);
  return (int)hipGetLastError();
}

// ---- measurement hook (tools/dispatch_lab.py): how fast does the hardware dispatch workgroups of a given shape? ----
// Every block touches its dynamic LDS once and spins for `spin` clock ticks.
__global__ void dispatch_probe_kernel(int spin, float* __restrict__ sink) {
  extern __shared__ float probe_lds[];
  probe_lds[threadIdx.x] = (float)blockIdx.x;
  const long long t0 = clock64();
  while (clock64() - t0 < spin) {}
  if (sink && probe_lds[threadIdx.x] < 0.f) sink[0] = 1.f;
}

extern "C" int mmt_debug_dispatch_probe(int blocks, int threads, int lds_bytes, int spin, float* sink, void* stream) {
  if (blocks <= 0 || threads <= 0 || threads > 1024 || lds_bytes < threads * 4 || lds_bytes > 160 * 1024) return MMT_ERR_ARG;
  static int configured = 0;
  if (lds_bytes > configured) {
    hipError_t rc = hipFuncSetAttribute((const void*)dispatch_probe_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        160 * 1024);
    if (rc != hipSuccess) return (int)rc;
    configured = 160 * 1024;
  }
  hipLaunchKernelGGL(dispatch_probe_kernel, dim3(blocks), dim3(threads), lds_bytes, (hipStream_t)stream, spin, sink);
  return (int)hipGetLastError();
}
