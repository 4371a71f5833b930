// One unit of the fused Adam step (torch.optim.Adam semantics, train.py:100 of the reference): a 64x64 tile of a
// weight matrix with bf16 shadows, or 4096 elements of a plain span, handled by 256 threads.  Shared by
//   * adam_fused_kernel / adam_queue_kernel (elementwise.hip): one unit per block, the optimizer as its own launch, and
//   * the RIDER blocks of the GEMM launches (gemm2.hip, adam_rider_run below): blocks of a backward GEMM launch that have
//     no tile of their own run queue entries whose gradients are already final (include/mmt_hip.h, "Adam riders"),
// so that every path rounds alike: every multiply-add is an explicit fma and every product an explicit multiply -- a
// sharded step, a per-region step and a ridden step must reproduce the single-launch step bit for bit
// (tests/test_dp_gpu.py, tests/test_optim_gpu.py).
#pragma once
#include "mmt_common.h"
#include "../../include/mmt_hip.h"

__device__ __forceinline__ void adam_update4(f32x4& pv, const f32x4& gv, f32x4& mv, f32x4& vv, float beta1, float beta2,
                                             float eps, float weight_decay, float step_size, float inv_sqrt_bc2) {
  const float omb1 = 1.0f - beta1, omb2 = 1.0f - beta2;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float gg = __builtin_fmaf(weight_decay, pv[k], gv[k]);
    mv[k] = __builtin_fmaf(beta1, mv[k], __fmul_rn(omb1, gg));
    vv[k] = __builtin_fmaf(beta2, vv[k], __fmul_rn(__fmul_rn(omb2, gg), gg));
    const float denom = __builtin_fmaf(sqrtf(vv[k]), inv_sqrt_bc2, eps);
    pv[k] = __builtin_fmaf(-step_size, __fdiv_rn(mv[k], denom), pv[k]);
  }
}

struct AdamHyper { float beta1, beta2, eps, weight_decay, step_size, inv_sqrt_bc2; };
// t = the 1-based number of THIS step (bias correction)
__device__ __forceinline__ AdamHyper adam_hyper(float lr, float beta1, float beta2, float eps, float weight_decay, int t_int) {
  const float t = (float)t_int;
  const float bc1 = 1.0f - powf(beta1, t), bc2 = 1.0f - powf(beta2, t);
  AdamHyper h;
  h.beta1 = beta1; h.beta2 = beta2; h.eps = eps; h.weight_decay = weight_decay;
  h.step_size = lr / bc1;
  h.inv_sqrt_bc2 = 1.0f / sqrtf(bc2);
  return h;
}

// First half of a unit: loads, update, stores of p / m / v (+ the row-major bf16 shadow), and the fp32 values of a
// matrix tile into `tile` for the transposed shadow.  tid in [0, 256).  Returns true when adam_unit_store_t must follow
// AFTER a barrier over the 256 threads that share `tile`.
__device__ __forceinline__ bool adam_unit_update(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                 float* __restrict__ v, const MmtAdamSeg& seg, int lb, int tid,
                                                 float (*tile)[65], const AdamHyper& h) {
  if (!seg.dst) {  // plain span: 4096 elements per unit, 16 B per lane, four sweeps
    const int64_t base = seg.offset + (int64_t)lb * 4096;
    const int64_t end = seg.offset + seg.count;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t e = base + (int64_t)(i * 256 + tid) * 4;
      if (e < end) {
        f32x4 pv = *(const f32x4*)(p + e), gv = __builtin_nontemporal_load((const f32x4*)(g + e));
        f32x4 mv = __builtin_nontemporal_load((const f32x4*)(m + e)), vv = __builtin_nontemporal_load((const f32x4*)(v + e));
        adam_update4(pv, gv, mv, vv, h.beta1, h.beta2, h.eps, h.weight_decay, h.step_size, h.inv_sqrt_bc2);
        *(f32x4*)(p + e) = pv;
        __builtin_nontemporal_store(mv, (f32x4*)(m + e));
        __builtin_nontemporal_store(vv, (f32x4*)(v + e));
      }
    }
    return false;
  }
  // shadowed matrix: tile (tr, tc) of 64x64; thread (r = tid/16 + 16 i, c = 4 (tid%16))
  const int tiles_c = (seg.cols + 63) >> 6;
  const int r0 = (lb / tiles_c) * 64, c0 = (lb % tiles_c) * 64;
  const int c = c0 + (tid & 15) * 4;
  bf16_t* __restrict__ dst = (bf16_t*)seg.dst;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int rl = (tid >> 4) + 16 * i, r = r0 + rl;
    f32x4 pv = {0.f, 0.f, 0.f, 0.f};
    if (r < seg.rows && c < seg.cols) {  // cols % 4 == 0: a 4-group is inside or outside as a whole
      const int64_t e = seg.offset + (int64_t)r * seg.cols + c;
      pv = *(const f32x4*)(p + e);
      const f32x4 gv = __builtin_nontemporal_load((const f32x4*)(g + e));
      f32x4 mv = __builtin_nontemporal_load((const f32x4*)(m + e)), vv = __builtin_nontemporal_load((const f32x4*)(v + e));
      adam_update4(pv, gv, mv, vv, h.beta1, h.beta2, h.eps, h.weight_decay, h.step_size, h.inv_sqrt_bc2);
      *(f32x4*)(p + e) = pv;
      __builtin_nontemporal_store(mv, (f32x4*)(m + e));
      __builtin_nontemporal_store(vv, (f32x4*)(v + e));
      u32x2 o = {pack_bf2(pv[0], pv[1]), pack_bf2(pv[2], pv[3])};
      *(u32x2*)(dst + (int64_t)r * seg.dst_ld + c) = o;
    }
    if (seg.dst_t) {
      const int cl = (tid & 15) * 4;
      tile[rl][cl] = pv[0]; tile[rl][cl + 1] = pv[1]; tile[rl][cl + 2] = pv[2]; tile[rl][cl + 3] = pv[3];
    }
  }
  return seg.dst_t != nullptr;
}

// Second half: the transposed bf16 shadow of the tile (row = source column, 16 consecutive source rows = 32 B per thread).
__device__ __forceinline__ void adam_unit_store_t(const MmtAdamSeg& seg, int lb, int tid, float (*tile)[65]) {
  const int tiles_c = (seg.cols + 63) >> 6;
  const int r0 = (lb / tiles_c) * 64, c0 = (lb % tiles_c) * 64;
  bf16_t* __restrict__ dst_t = (bf16_t*)seg.dst_t;
  const int tcl = tid >> 2, rq = (tid & 3) * 16;
  const int tcol = c0 + tcl;  // row of dst_t
  if (tcol < seg.cols) {
    unsigned w[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) w[k] = pack_bf2(tile[rq + 2 * k][tcl], tile[rq + 2 * k + 1][tcl]);
    bf16_t* out = dst_t + (int64_t)tcol * seg.dst_t_ld + r0 + rq;
    if (r0 + rq + 15 < seg.rows) {
      *(u32x4*)out = (u32x4){w[0], w[1], w[2], w[3]};
      *(u32x4*)(out + 8) = (u32x4){w[4], w[5], w[6], w[7]};
    } else {
#pragma unroll
      for (int k = 0; k < 16; ++k)
        if (r0 + rq + k < seg.rows) out[k] = (bf16_t)((k & 1) ? (w[k >> 1] >> 16) : (w[k >> 1] & 0xffffu));
    }
  }
}

// ---- riders ------------------------------------------------------------------------------------------------------
// LDS a rider block of NT threads needs at `smem`: a control word line + one fp32 tile per group of 256 threads.
#define MMT_RIDER_TILE_BYTES (64 * 65 * 4)
#define MMT_RIDER_CHUNK 2  // queue entries a rider block claims per pop (one fetch-add per chunk; a claimed chunk is always run)
#define MMT_RIDER_CAP 64   // rider blocks at work per launch, chip-wide, unless MmtEpilogue.rider_cap says otherwise: a pass of one
                           // block moves 256 KB, ~20 block-passes per microsecond is what HBM delivers beside the GEMM
template <int NT>
constexpr int adam_rider_lds_bytes() { return 64 + (NT / 256) * MMT_RIDER_TILE_BYTES; }

// queue state words (MmtAdamQueue.state): [0, STAGES) claim counters, [STAGES] ticket, the hosts' finished-block counters,
// two statistics words (+ padding), the first-level tickets
#define MMT_RIDER_TICKET MMT_RIDER_STAGES
#define MMT_RIDER_SLOT0 (MMT_RIDER_STAGES + 1)
#define MMT_RIDER_STAT0 (MMT_RIDER_STAGES + 1 + MMT_RIDER_SLOTS)
#define MMT_RIDER_SUBTICKET0 (MMT_RIDER_STAT0 + 64)  // 64 first-level ticket words of mmt_adam_step_queue, one 256-byte line each
static_assert(MMT_RIDER_STATE_WORDS == MMT_RIDER_SUBTICKET0 + 64 * 64, "queue state layout (include/mmt_hip.h)");

// The hosting launch reports "my last round of tiles is ending" here: ONE block per launch does -- the first block of the last
// round, thread 0, at the end of its K-loop (blocks of a round run in lock-step: one representative is enough).  r06's first
// versions let EVERY block with a tile report: 232-696 agent-scope atomics on one word, all within a microsecond, are
// serialised at ~30 ns each and the launch cannot end before its last one has -- +3...5 us on every hosting launch whatever
// the riders did.  x index of the reporting block (its y index is 0):
__device__ __forceinline__ int adam_rider_signal_x(int live_tiles, int gy, int slot_word) {
  int resident = (slot_word >> 16) & 0xffff;
  if (resident <= 0) resident = 256;
  const int live_total = live_tiles * gy;
  const int last_round = live_total > 0 ? ((live_total - 1) % resident) + 1 : 0;
  const int first = live_tiles - (last_round + gy - 1) / gy;  // (blocks are numbered x-major within a y row: approximate for gy > 1)
  return first > 0 ? first : 0;
}
__device__ __forceinline__ void adam_rider_host_done(const void* rider, int slot_word) {
  const MmtAdamQueue* q = (const MmtAdamQueue*)rider;
  __hip_atomic_fetch_add(q->state + MMT_RIDER_SLOT0 + (slot_word & 0xffff), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Claim up to MMT_RIDER_CHUNK entries of the first `stages` stages of `q`: ONE fetch-add on the claim counter of the first
// stage that is not known to be exhausted (s0: block-local memory of that; a counter only grows within a step, so an
// exhausted stage stays exhausted).  A compare-and-swap pop serialises: with 512 blocks contending for one word a pop took
// ~1.3 us (first r06 version: 713 us for a 9 us host launch); a fetch-add never retries, and overshooting a stage's size is
// harmless -- mmt_adam_step_queue clamps.
__device__ __forceinline__ bool adam_rider_claim(const MmtAdamQueue* __restrict__ q, int stages, int& s0, int& first, int& n) {
  while (s0 < stages) {
    const int lo = q->stage_begin[s0], size = q->stage_begin[s0 + 1] - lo;
    const int k = size > 0 ? __hip_atomic_fetch_add(q->state + s0, MMT_RIDER_CHUNK, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : size;
    if (k < size) {
      first = lo + k;
      n = size - k < MMT_RIDER_CHUNK ? size - k : MMT_RIDER_CHUNK;
      return true;
    }
    ++s0;
  }
  return false;
}

// A block of NT threads (NT a multiple of 256) with no tile of the hosting launch: claim chunks of queue entries and run
// them, NT / 256 entries at a time (one per group of 256 threads), until the stages that are ready are exhausted or the
// host launch is in its tail.
//   stages: leading stages of the queue whose gradients are final (MmtEpilogue.rider_limit)
//   slot_word: low 16 bits = the launch's finished-block counter, high 16 bits = how many of the launch's blocks are resident
//   at once (chip-wide); live_total = blocks of the launch that compute a tile; rank = this block's number among the
//   launch's tile-less blocks (they are dispatched in this order, after every block with a tile); cap = rider blocks at work.
// Blocks are dispatched in grid order, so the launch's LAST round of tiles is live_total mod resident blocks (or a full
// round) and leaves resident - last_round residency slots free: tile-less blocks of higher rank than that are only dispatched
// when the launch is ending and return without touching memory.  The others work until the first block of the last round
// reports the end of its K-loop -- a pass is ~2 us of streaming, the launch must not wait for it.
template <int NT>
__device__ __forceinline__ void adam_rider_run(const void* rider, int stages, int slot_word, int live_total, int rank, int cap,
                                               unsigned char* smem) {
  static_assert(NT % 256 == 0, "rider groups are 256 threads");
  constexpr int G = NT / 256;
  int resident = (slot_word >> 16) & 0xffff;
  if (resident <= 0) resident = 256;
  const int last_round = live_total > 0 ? ((live_total - 1) % resident) + 1 : 0;
  const int free_slots = resident - last_round;
  // cap: bits 0..11 = rider blocks at work (0: MMT_RIDER_CAP), bits 12..15 = passes a block makes (0: until the host's tail
  // begins; n: exactly n claims, no polling of the host -- a launch of known length cannot then be outlasted by a rider)
  const int max_pass = (cap >> 12) & 15;
  cap &= 0xfff;
  if (cap <= 0) cap = MMT_RIDER_CAP;
  if (rank >= (free_slots < cap ? free_slots : cap)) return;  // (no memory access: most tile-less blocks leave here)
  const MmtAdamQueue* __restrict__ qd = (const MmtAdamQueue*)rider;
  const int tid = threadIdx.x, grp = tid >> 8, t256 = tid & 255;
  volatile int* ctl = (volatile int*)smem;
  float (*tile)[65] = (float (*)[65])(smem + 64 + grp * MMT_RIDER_TILE_BYTES);
  const int slot = slot_word & 0xffff;
  const int thresh = 1;  // the launch's reporting block (adam_rider_signal_x) has reached the end of its K-loop
  const MmtAdamQueue* __restrict__ chain = qd->chain;
  const int chain_stages = chain ? qd->chain_stages : 0;
  if (stages > qd->n_stages) stages = qd->n_stages;
  int s_own = 0, s_chain = 0;  // (thread 0's) first stages not known to be exhausted
  for (int pass = 0; max_pass == 0 || pass < max_pass; ++pass) {
    if (tid == 0) {
      int first = -1, n = 0, which = 0;
      const int done = live_total > 0 && max_pass == 0
          ? __hip_atomic_load(qd->state + MMT_RIDER_SLOT0 + slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
      if (live_total <= 0 || done < thresh) {
        if (chain && adam_rider_claim(chain, chain_stages, s_chain, first, n)) which = 1;
        else if (!adam_rider_claim(qd, stages, s_own, first, n)) first = -1;
      }
      ctl[0] = first; ctl[1] = n; ctl[2] = which;
    }
    __syncthreads();
    const int first = ctl[0], n = ctl[1], which = ctl[2];
    if (first < 0) break;  // (block-uniform)
    const MmtAdamQueue* __restrict__ q = which ? chain : qd;
    const float lr = q->lr_dev ? *q->lr_dev : q->lr;
    const int t_int = __builtin_amdgcn_readfirstlane(*(const int32_t*)q->step_dev) + 1;  // the step in progress
    const AdamHyper h = adam_hyper(lr, q->beta1, q->beta2, q->eps, q->weight_decay, t_int);
    for (int u = 0; u < n; u += G) {  // the claimed chunk, G entries per pass
      const bool valid = u + grp < n;
      bool tr = false;
      MmtAdamSeg seg = {};
      int lb = 0;
      if (valid) {
        const int k = first + u + grp;
        const int si = __builtin_amdgcn_readfirstlane(q->unit_seg[k]);
        lb = __builtin_amdgcn_readfirstlane(q->unit_blk[k]);
        seg = q->segs[si];
        tr = adam_unit_update(q->p, q->g, q->m, q->v, seg, lb, t256, tile, h);
      }
      __syncthreads();
      if (tr) adam_unit_store_t(seg, lb, t256, tile);
      __syncthreads();  // ctl and the tiles are free again
    }
  }
}
