// Fused masked-softmax attention of the video-BERT, forward and backward (gfx950, head dim 128).
// Replaces bert.py:141-168 (QK^T/sqrt(dh) + additive -10000 key mask -> softmax -> dropout -> .V ->
// merge heads) and its autograd backward; the (B,H,S,S) probability tensor is never materialised.
//
// Layout: qkv bf16 [rows, 3d] = [Q | K | V] from the fused QKV GEMM, head h at columns h*128 of each
// section; sample b owns rows cu[b]..cu[b+1] (dense: b*S..).  ctx bf16 [rows, d].
//
// All three kernels use the "swapped" MFMA form so that the softmax row (one query) is lane-local:
//   S^T[key][q] = K . Q^T            a = K fragment (ds_read_b128), b = Q fragment (registers)
//   O^T[d][q]  += V^T . P^T          a = V^T fragment (ds_read_b64_tr_b16 from the row-major V tile)
// A lane (q = lane&15, g = lane>>4) then holds scores for keys 16f + 4g + r, and after bf16 packing those
// registers ARE the b-operand of the second MFMA (k-index permutation kappa(g,j) shared with the
// transpose read), so P never leaves registers.
// 64x128 bf16 tiles live in LDS as 256-B rows whose 16-B chunks are XOR-swizzled with
// SWZ16(row) = ((row&7)<<1)|((row>>3)&1): conflict-free for both the b128 (row-per-lane) and the
// transpose reads.  Tiles arrive by LDS-DMA with the swizzle applied on the source address.
//
// Dropout mask of element (b,h,q,k): hash of (key, (b*H+h)*S4+q) then of (k>>1); q,k are the ORIGINAL positions of the
// tokens inside the sample (row_index[row] - b*S when the rows are packed), so a packed run draws exactly the mask of
// the dense run.  Backward regenerates it.
#include <stdlib.h>
#include "mmt_common.h"
#include "../../include/mmt_hip.h"

#define LOG2E 1.4426950408889634f
#define LN2 0.6931471805599453f
#define NEG_BIG (-1.0e30f)

// XOR swizzle of the 16-B chunks of a tile row.  Head dim 128 (256-B rows, 16 chunks): ((row&7)<<1)|((row>>3)&1);
// head dim 64 (128-B rows, 8 chunks; two rows share one 256-B bank line): (row>>1)&7.
template <int DH> __device__ __forceinline__ int swz(int r) {
  return DH == 128 ? (((r & 7) << 1) | ((r >> 3) & 1)) : ((r >> 1) & 7);
}

// 64 x DH bf16 tile by LDS-DMA: one wave-instruction moves 1 KiB = 64 / (DH/8) rows
template <int DH, int NW = 4>
__device__ __forceinline__ void stage64(const bf16_t* __restrict__ G, int64_t ld, int row0, int row_last,
                                        bf16_t* lds_tile, int wave, int lane) {
  constexpr int CH = DH / 8, RPI = 64 / CH, NI = 64 / NW / RPI;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int rbase = (wave * NI + i) * RPI;
    const int rr = rbase + lane / CH;
    const int c = (lane % CH) ^ swz<DH>(rr);
    const int gr = min(row0 + rr, row_last);
    const bf16_t* src = G + (int64_t)gr * ld + c * 8;
    __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(lds_tile + rbase * DH), 16, 0, 0);
  }
}

// same tile, rows gathered through an index list: row i of the tile = G[sel[min(i0 + i, n - 1)]]
template <int DH>
__device__ __forceinline__ void stage64_sel(const bf16_t* __restrict__ G, int64_t ld, const int32_t* __restrict__ sel,
                                            int i0, int n, bf16_t* lds_tile, int wave, int lane) {
  constexpr int CH = DH / 8, RPI = 64 / CH, NI = 16 / RPI;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int rbase = (wave * NI + i) * RPI;
    const int rr = rbase + lane / CH;
    const int c = (lane % CH) ^ swz<DH>(rr);
    const int gr = sel[min(i0 + rr, n - 1)];
    const bf16_t* src = G + (int64_t)gr * ld + c * 8;
    __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(lds_tile + rbase * DH), 16, 0, 0);
  }
}

template <int DH> __device__ __forceinline__ bf16x8_t frag_b128(const bf16_t* tile, int r, int chunk) {
  return *(const bf16x8_t*)(tile + r * DH + ((chunk ^ swz<DH>(r)) << 3));
}

// 8 contraction values (rows kappa(g,.) of k-chunk ks) for column colbase + (lane&15)
template <int DH> __device__ __forceinline__ bf16x8_t frag_tr(const bf16_t* tile, int ks, int colbase, int lane) {
  const int t = lane & 15, g = lane >> 4;
  const int col = colbase + 4 * (t & 3);
  const int r0 = ks * 32 + 4 * g + (t >> 2), r1 = r0 + 16;
  const int ch = col >> 3, w = col & 7;
  const bf16_t* p0 = tile + r0 * DH + ((ch ^ swz<DH>(r0)) << 3) + w;
  const bf16_t* p1 = tile + r1 * DH + ((ch ^ swz<DH>(r1)) << 3) + w;
  bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4_t*)LDS_PTR(p0));
  bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4_t*)LDS_PTR(p1));
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

__device__ __forceinline__ bf16x8_t pack8(const f32x4& a, const f32x4& b) {
  u32x4 u = {pack_bf2(a[0], a[1]), pack_bf2(a[2], a[3]), pack_bf2(b[0], b[1]), pack_bf2(b[2], b[3])};
  return __builtin_bit_cast(bf16x8_t, u);
}

__device__ __forceinline__ unsigned attn_rowkey(unsigned key, unsigned bh, unsigned S4, unsigned q) {
  return mix32(key ^ ((bh * S4 + q) * 0x9e3779b9U));
}
__device__ __forceinline__ bool attn_keep(unsigned rowkey, unsigned k, unsigned thr16) {
  const unsigned r = mix32(rowkey ^ (k >> 1));
  return ((k & 1) ? (r >> 16) : (r & 0xffffU)) >= thr16;
}

// A wave's 16 x DH result block (lane (row li, group lg) holds 4 consecutive columns of DH/16 fragments) written out as
// whole 16-byte chunks of full rows: transposed through a wave-private LDS corner `ob` (16 * (DH + 8) bf16).  row_ptr(r)
// gives the global address of row r's first column, or nullptr to skip the row.
template <int DH, typename RowPtr>
__device__ __forceinline__ void store_block16(bf16_t* ob, const f32x4* o, float mul, int lane, RowPtr row_ptr) {
  constexpr int PITCH = DH + 8, CPR = DH / 8;
  const int li = lane & 15, lg = lane >> 4;
#pragma unroll
  for (int fd = 0; fd < DH / 16; ++fd) {
    u32x2 v = {pack_bf2(o[fd][0] * mul, o[fd][1] * mul), pack_bf2(o[fd][2] * mul, o[fd][3] * mul)};
    *(u32x2*)(ob + li * PITCH + fd * 16 + 4 * lg) = v;
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the same wave wrote and reads
#pragma unroll
  for (int it = 0; it < 16 * CPR / 64; ++it) {
    const int idx = it * 64 + lane, r = idx / CPR, c = idx % CPR;
    bf16_t* dst = row_ptr(r);
    if (dst) *(u32x4*)(dst + c * 8) = *(const u32x4*)(ob + r * PITCH + c * 8);
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);  // the corner may be reused by the caller
}

#ifdef MMT_GEMM2_INSTR
// lab build only (python -m mmt_amd.build --instr; tools/attn_budget.py): per-block phase timestamps of the backward kernel
__device__ long long* g_attn_dbg = nullptr;
extern "C" int mmt_debug_set_attn_buffer(void* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_attn_dbg), &p, sizeof(p)); }
#define ATT_BLK() ((int64_t)blockIdx.x * 16)
#define ATT_MARK(k) do { if (g_attn_dbg && threadIdx.x == 0) g_attn_dbg[ATT_BLK() + (k)] = clock64(); } while (0)
#define ATT_SET(k, v) do { if (g_attn_dbg && threadIdx.x == 0) g_attn_dbg[ATT_BLK() + (k)] = (long long)(v); } while (0)
#else
#define ATT_MARK(k) do {} while (0)
#define ATT_SET(k, v) do {} while (0)
#endif

struct AttnArgs {
  const bf16_t* qkv; int64_t ld;       // [rows, 3d]
  const int32_t* cu; int S_dense;      // cu nullable => dense b*S_dense
  const float* mask_bias;              // [rows] 0 / -10000 (additive, bert.py:395)
  bf16_t* ctx; int64_t ldc;            // [rows, d]   (fwd: out; bwd: in)
  float* lse;                          // [rows, H] natural-log logsumexp of the scaled+masked scores
  const bf16_t* dctx;                  // [rows, d] bwd
  bf16_t* dqkv;                        // [rows, 3d] bwd out
  float* delta;                        // [rows, H] bwd scratch: rowsum(dO * O)
  int H, d, B; float scale;
  uint32_t drop_key, thr16; float drop_scale; int S4;
  const uint32_t* seed_dev;
  // Query subset (last encoder layer: only the rows that are read out need a context vector).  qsel[b*nq + i] is
  // the row of sample b's i-th selected query; ctx / lse / dctx / delta are then COMPACT [B*nq, .] buffers, while
  // qkv / dqkv keep the full token layout (every key still participates).
  const int32_t* qsel; int nq;
  // token packing: row_index[row] = b * S_dense + original position (nullable: rows are dense, position = row - b*S)
  const int32_t* row_index;
};

// original position of (packed) row `row` of sample b
__device__ __forceinline__ int orig_pos(const AttnArgs& a, int b, int off, int row) {
  return a.row_index ? a.row_index[row] - b * a.S_dense : row - off;
}
// keep flags of four consecutive (packed) keys whose original positions are kp[0..3]
__device__ __forceinline__ void keep4_keys(unsigned rowkey, const int* kp, bool contiguous, unsigned thr16, bool keep[4]) {
  if (contiguous) {  // dense rows: kp = k0 .. k0+3 with k0 % 4 == 0 -> two hashes serve four keys
    const unsigned k0 = (unsigned)kp[0];
    const unsigned r01 = mix32(rowkey ^ (k0 >> 1)), r23 = mix32(rowkey ^ ((k0 >> 1) + 1));
    keep[0] = (r01 & 0xffffU) >= thr16; keep[1] = (r01 >> 16) >= thr16;
    keep[2] = (r23 & 0xffffU) >= thr16; keep[3] = (r23 >> 16) >= thr16;
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) keep[e] = attn_keep(rowkey, (unsigned)kp[e], thr16);
  }
}

// ------------------------------------------------------------------------------------------------
// forward: grid (q tiles of 16 NW, H, B), NW waves x 16 queries.  NW = 8 (one block per CU) stages every key / value
// tile once per 128 queries instead of once per 64: half the L2 -> LDS traffic per CU for the same waves per CU.
// ------------------------------------------------------------------------------------------------
template <int DH, int NW>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) void attn_fwd_kernel(AttnArgs a) {
  __shared__ __attribute__((aligned(16))) bf16_t smem[2 * 2 * 64 * DH + 2 * 64 * 2 + 2 * 64 * 2];
  float* bias_s = (float*)(smem + 2 * 2 * 64 * DH);  // [2][64]
  int* kpos_s = (int*)(bias_s + 2 * 64);              // [2][64] original positions of the tile's keys
  // 1-D grid, (sample, head) fastest: consecutive block ids -- which the dispatcher deals out to the 8 XCDs in turn -- are the
  // same query tile of different (sample, head) pairs.  With (q tile, head, sample) as grid (x, y, z) every XCD received ONE
  // (q tile, head mod ..) combination, and under token packing the later q tiles are mostly empty: half the XCDs idled.
  const int nbh = a.H * a.B, bh = (int)blockIdx.x % nbh;
  const int b = bh / a.H, h = bh % a.H;
  const int off = a.cu ? a.cu[b] : b * a.S_dense;
  const int Sb = a.cu ? a.cu[b + 1] - off : a.S_dense;
  const int q0 = ((int)blockIdx.x / nbh) * (16 * NW);
  const int nqs = a.qsel ? a.nq : Sb;  // queries of this sample
  if (q0 >= nqs || Sb <= 0) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
  const int row_last = off + Sb - 1;
  const int nkt = (Sb + 63) >> 6;
  const bf16_t* Kg = a.qkv + a.d + h * DH;
  const bf16_t* Vg = a.qkv + 2 * a.d + h * DH;
  const bool dense_keys = a.row_index == nullptr;

  const int qi = q0 + wave * 16 + li;
  const int qic = min(qi, nqs - 1);
  const int qrow = a.qsel ? a.qsel[b * a.nq + qic] : off + qic;
  const int crow = a.qsel ? b * a.nq + qic : qrow;  // row in ctx / lse
  const int q_local = orig_pos(a, b, off, qrow);
  bf16x8_t qf[DH / 32];
#pragma unroll
  for (int kk = 0; kk < DH / 32; ++kk)
    qf[kk] = *(const bf16x8_t*)(a.qkv + (int64_t)qrow * a.ld + h * DH + kk * 32 + lg * 8);
  const unsigned rowkey = attn_rowkey(eff_key(a.drop_key, a.seed_dev), (unsigned)(b * a.H + h), (unsigned)a.S4, (unsigned)q_local);
  const float c1 = a.scale * LOG2E;

  f32x4 o[DH / 16];
#pragma unroll
  for (int i = 0; i < DH / 16; ++i) o[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float m_run = NEG_BIG, l_run = 0.f;

  auto stage = [&](int kt, int st) {
    bf16_t* base = smem + st * (2 * 64 * DH);
    stage64<DH, NW>(Kg, a.ld, off + kt * 64, row_last, base, wave, lane);
    stage64<DH, NW>(Vg, a.ld, off + kt * 64, row_last, base + 64 * DH, wave, lane);
    if (tid < 64) {
      const int k = kt * 64 + tid;
      bias_s[st * 64 + tid] = k < Sb ? a.mask_bias[off + k] * LOG2E : -INFINITY;
      kpos_s[st * 64 + tid] = k < Sb ? orig_pos(a, b, off, off + k) : k;
    }
  };
  stage(0, 0);
  for (int kt = 0; kt < nkt; ++kt) {
    const int cur = kt & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kt + 1 < nkt) stage(kt + 1, cur ^ 1);
    const bf16_t* Ks = smem + cur * (2 * 64 * DH);
    const bf16_t* Vs = Ks + 64 * DH;
    f32x4 s[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      s[f] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < DH / 32; ++kk)
        s[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_b128<DH>(Ks, f * 16 + li, kk * 4 + lg), qf[kk], s[f], 0, 0, 0);
    }
    float mt = NEG_BIG;
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      const f32x4 bv = *(const f32x4*)(bias_s + cur * 64 + f * 16 + 4 * lg);
#pragma unroll
      for (int r = 0; r < 4; ++r) { s[f][r] = s[f][r] * c1 + bv[r]; mt = fmaxf(mt, s[f][r]); }
    }
    mt = fmaxf(mt, __shfl_xor(mt, 16, 64));
    mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
    const float m_new = fmaxf(m_run, mt);
    const float alpha = exp2f(m_run - m_new);
    float psum = 0.f;
#pragma unroll
    for (int f = 0; f < 4; ++f) {
#pragma unroll
      for (int r = 0; r < 4; ++r) { s[f][r] = exp2f(s[f][r] - m_new); psum += s[f][r]; }
      if (a.thr16) {
        bool kp[4];
        keep4_keys(rowkey, kpos_s + cur * 64 + f * 16 + 4 * lg, dense_keys, a.thr16, kp);
#pragma unroll
        for (int r = 0; r < 4; ++r) s[f][r] = kp[r] ? s[f][r] * a.drop_scale : 0.f;
      }
    }
    psum += __shfl_xor(psum, 16, 64);
    psum += __shfl_xor(psum, 32, 64);
    l_run = l_run * alpha + psum;
    m_run = m_new;
#pragma unroll
    for (int i = 0; i < DH / 16; ++i) o[i] *= alpha;
    const bf16x8_t pb0 = pack8(s[0], s[1]), pb1 = pack8(s[2], s[3]);
#pragma unroll
    for (int fd = 0; fd < DH / 16; ++fd) {
      o[fd] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tr<DH>(Vs, 0, fd * 16, lane), pb0, o[fd], 0, 0, 0);
      o[fd] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tr<DH>(Vs, 1, fd * 16, lane), pb1, o[fd], 0, 0, 0);
    }
  }
  // Output through LDS (store_block16): stored directly, a lane's result is 8 partial-line writes 32 B apart (measured:
  // 3 us of an 18 us launch).
  __syncthreads();
  if (qi < nqs && lg == 0) a.lse[(int64_t)crow * a.H + h] = (m_run + log2f(l_run)) * LN2;
  store_block16<DH>(smem + wave * 16 * (DH + 8), o, 1.0f / l_run, lane, [&](int r) -> bf16_t* {
    const int qr = q0 + wave * 16 + r;
    if (qr >= nqs) return nullptr;
    return a.ctx + (a.qsel ? (int64_t)b * a.nq + qr : (int64_t)off + qr) * a.ldc + h * DH;
  });
}

// ------------------------------------------------------------------------------------------------
// backward, dQ role (+ delta output).  Same tiling as forward: one block per (q tile, H, B).
//   dA^T[key][q] = V . dO^T ; dS = P o (keep*dA*sc - delta) ; dQ^T[d][q] += K^T . dS^T
// ------------------------------------------------------------------------------------------------
template <int DH>
__device__ __forceinline__ void attn_bwd_dq_block(const AttnArgs& a, bf16_t* smem, int bx, int b, int h) {
  float* bias_s = (float*)(smem + 2 * 2 * 64 * DH);
  int* kpos_s = (int*)(bias_s + 2 * 64);
  const bool dense_keys = a.row_index == nullptr;
  const int off = a.cu ? a.cu[b] : b * a.S_dense;
  const int Sb = a.cu ? a.cu[b + 1] - off : a.S_dense;
  const int q0 = bx * 64;
  const int nqs = a.qsel ? a.nq : Sb;
  if (q0 >= nqs || Sb <= 0) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
  const int row_last = off + Sb - 1;
  const int nkt = (Sb + 63) >> 6;
  const bf16_t* Kg = a.qkv + a.d + h * DH;
  const bf16_t* Vg = a.qkv + 2 * a.d + h * DH;
  const int qi = q0 + wave * 16 + li;
  const bool q_ok = qi < nqs;
  const int qic = min(qi, nqs - 1);
  const int qrow = a.qsel ? a.qsel[b * a.nq + qic] : off + qic;
  const int crow = a.qsel ? b * a.nq + qic : qrow;  // row in ctx / dctx / lse / delta
  const int q_local = orig_pos(a, b, off, qrow);
  if (a.qsel && bx == 0) {
    // query-subset mode: dQ of the non-selected rows is zero -- this block (sample b, head h) clears its 128 columns
    // of every row of the sample; the selected rows are overwritten at the end (after the K-loop's barriers)
    for (int e = tid; e < Sb * (DH / 8); e += 256) {
      u32x4 z = {0, 0, 0, 0};
      *(u32x4*)(a.dqkv + (int64_t)(off + e / (DH / 8)) * a.ld + h * DH + (e % (DH / 8)) * 8) = z;
    }
  }
  bf16x8_t qf[DH / 32], dof[DH / 32];
  float dl = 0.f;
#pragma unroll
  for (int kk = 0; kk < DH / 32; ++kk) {
    qf[kk] = *(const bf16x8_t*)(a.qkv + (int64_t)qrow * a.ld + h * DH + kk * 32 + lg * 8);
    const u16x8 dv = *(const u16x8*)(a.dctx + (int64_t)crow * a.ldc + h * DH + kk * 32 + lg * 8);
    const u16x8 ov = *(const u16x8*)(a.ctx + (int64_t)crow * a.ldc + h * DH + kk * 32 + lg * 8);
    dof[kk] = __builtin_bit_cast(bf16x8_t, dv);
#pragma unroll
    for (int e = 0; e < 8; ++e) dl += bf2f(dv[e]) * bf2f(ov[e]);
  }
  dl += __shfl_xor(dl, 16, 64);
  dl += __shfl_xor(dl, 32, 64);
  if (q_ok && lg == 0) a.delta[(int64_t)crow * a.H + h] = dl;
  const float lse2 = a.lse[(int64_t)crow * a.H + h] * LOG2E;
  const unsigned rowkey = attn_rowkey(eff_key(a.drop_key, a.seed_dev), (unsigned)(b * a.H + h), (unsigned)a.S4, (unsigned)q_local);
  const float c1 = a.scale * LOG2E;

  f32x4 o[DH / 16];
#pragma unroll
  for (int i = 0; i < DH / 16; ++i) o[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

  auto stage = [&](int kt, int st) {
    bf16_t* base = smem + st * (2 * 64 * DH);
    stage64<DH>(Kg, a.ld, off + kt * 64, row_last, base, wave, lane);
    stage64<DH>(Vg, a.ld, off + kt * 64, row_last, base + 64 * DH, wave, lane);
    if (tid < 64) {
      const int k = kt * 64 + tid;
      bias_s[st * 64 + tid] = k < Sb ? a.mask_bias[off + k] * LOG2E : -INFINITY;
      kpos_s[st * 64 + tid] = k < Sb ? orig_pos(a, b, off, off + k) : k;
    }
  };
  stage(0, 0);
  ATT_MARK(1); ATT_SET(5, nkt); ATT_SET(6, 0); ATT_SET(7, Sb);
  for (int kt = 0; kt < nkt; ++kt) {
    const int cur = kt & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kt == 0) ATT_MARK(2);
    if (kt == 1) ATT_MARK(8);
    if (kt + 1 < nkt) stage(kt + 1, cur ^ 1);
    const bf16_t* Ks = smem + cur * (2 * 64 * DH);
    const bf16_t* Vs = Ks + 64 * DH;
    f32x4 s[4], da[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      s[f] = (f32x4){0.f, 0.f, 0.f, 0.f};
      da[f] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < DH / 32; ++kk) {
        s[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_b128<DH>(Ks, f * 16 + li, kk * 4 + lg), qf[kk], s[f], 0, 0, 0);
        da[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_b128<DH>(Vs, f * 16 + li, kk * 4 + lg), dof[kk], da[f], 0, 0, 0);
      }
    }
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      const f32x4 bv = *(const f32x4*)(bias_s + cur * 64 + f * 16 + 4 * lg);
      bool kp[4] = {true, true, true, true};
      if (a.thr16) keep4_keys(rowkey, kpos_s + cur * 64 + f * 16 + 4 * lg, dense_keys, a.thr16, kp);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = exp2f(s[f][r] * c1 + bv[r] - lse2);
        const float dp = kp[r] ? da[f][r] * a.drop_scale : 0.f;
        s[f][r] = p * (dp - dl);  // dS
      }
    }
    const bf16x8_t sb0 = pack8(s[0], s[1]), sb1 = pack8(s[2], s[3]);
#pragma unroll
    for (int fd = 0; fd < DH / 16; ++fd) {
      o[fd] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tr<DH>(Ks, 0, fd * 16, lane), sb0, o[fd], 0, 0, 0);
      o[fd] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tr<DH>(Ks, 1, fd * 16, lane), sb1, o[fd], 0, 0, 0);
    }
  }
  __syncthreads();  // stage buffers idle: each wave transposes its block in its own corner
  ATT_MARK(3);
  store_block16<DH>(smem + wave * 16 * (DH + 8), o, a.scale, lane, [&](int r) -> bf16_t* {
    const int qr = q0 + wave * 16 + r;
    if (qr >= nqs) return nullptr;
    const int64_t grow = a.qsel ? (int64_t)a.qsel[b * a.nq + qr] : (int64_t)off + qr;
    return a.dqkv + grow * a.ld + h * DH;
  });
}

// ------------------------------------------------------------------------------------------------
// backward, dK / dV role.  One block per (key tile of 64, H, B); each wave owns 16 keys, loops over q tiles.
//   S[q][key] = Q . K^T (a = Q frag from LDS, b = K frag in registers) ; dA = dO . V^T
//   dV^T[d][key] += dO^T . A ; dK^T[d][key] += Q^T . dS     (a = transpose reads of the dO / Q tiles)
// ------------------------------------------------------------------------------------------------
template <int DH>
__device__ __forceinline__ void attn_bwd_dkv_block(const AttnArgs& a, bf16_t* smem, int bx, int b, int h) {
  float* aux_s = (float*)(smem + 2 * 2 * 64 * DH);  // [2][3][64]: lse2, delta, rowkey(bits)
  const int off = a.cu ? a.cu[b] : b * a.S_dense;
  const int Sb = a.cu ? a.cu[b + 1] - off : a.S_dense;
  const int k0 = bx * 64;
  if (k0 >= Sb) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
  const int row_last = off + Sb - 1;
  const int nqs = a.qsel ? a.nq : Sb;
  const int nqt = (nqs + 63) >> 6;
  const bf16_t* Qg = a.qkv + h * DH;
  const bf16_t* dOg = a.dctx + h * DH;
  const int key_local = k0 + wave * 16 + li;
  const bool key_ok = key_local < Sb;
  const int krow = off + min(key_local, Sb - 1);
  const int key_pos = orig_pos(a, b, off, krow);  // RNG coordinate of this lane's key
  bf16x8_t kf[DH / 32], vf[DH / 32];
#pragma unroll
  for (int kk = 0; kk < DH / 32; ++kk) {
    kf[kk] = *(const bf16x8_t*)(a.qkv + (int64_t)krow * a.ld + a.d + h * DH + kk * 32 + lg * 8);
    vf[kk] = *(const bf16x8_t*)(a.qkv + (int64_t)krow * a.ld + 2 * a.d + h * DH + kk * 32 + lg * 8);
  }
  const float bias2 = key_ok ? a.mask_bias[krow] * LOG2E : -INFINITY;
  const float c1 = a.scale * LOG2E;
  const unsigned bh = (unsigned)(b * a.H + h);
  const unsigned dkey = eff_key(a.drop_key, a.seed_dev);

  f32x4 dk[DH / 16], dv[DH / 16];
#pragma unroll
  for (int i = 0; i < DH / 16; ++i) { dk[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

  auto stage = [&](int qt, int st) {
    bf16_t* base = smem + st * (2 * 64 * DH);
    if (a.qsel) {
      stage64_sel<DH>(Qg, a.ld, a.qsel + b * a.nq, qt * 64, a.nq, base, wave, lane);
      stage64<DH>(dOg, a.ldc, b * a.nq + qt * 64, b * a.nq + a.nq - 1, base + 64 * DH, wave, lane);
    } else {
      stage64<DH>(Qg, a.ld, off + qt * 64, row_last, base, wave, lane);
      stage64<DH>(dOg, a.ldc, off + qt * 64, row_last, base + 64 * DH, wave, lane);
    }
    {
      // delta = rowsum(dO * O) of this head is formed here (the dQ blocks of the same launch produce it too, but nothing
      // orders the two roles): four neighbouring lanes per query row, each one 8-column group of every 32 (the dQ role's
      // split), combined in its order ((g0 + g1) + (g2 + g3))
      const int ql = tid >> 2, g = tid & 3;
      const int q = qt * 64 + ql;
      const int qc = min(q, nqs - 1);
      const int row = a.qsel ? b * a.nq + qc : off + qc;               // row in lse / ctx / dctx
      float* ax = aux_s + st * 192;
      const bf16_t* dop = a.dctx + (int64_t)row * a.ldc + h * DH + g * 8;
      const bf16_t* op = a.ctx + (int64_t)row * a.ldc + h * DH + g * 8;
      float part = 0.f;
#pragma unroll
      for (int kk = 0; kk < DH / 32; ++kk) {
        const u16x8 dv = *(const u16x8*)(dop + kk * 32);
        const u16x8 ov = *(const u16x8*)(op + kk * 32);
#pragma unroll
        for (int e = 0; e < 8; ++e) part += bf2f(dv[e]) * bf2f(ov[e]);
      }
      part += __shfl_xor(part, 1, 64);
      part += __shfl_xor(part, 2, 64);
      if (g == 0) {
        const int qpos = orig_pos(a, b, off, a.qsel ? a.qsel[b * a.nq + qc] : off + qc);  // original position (RNG coordinate)
        ax[ql] = q < nqs ? a.lse[(int64_t)row * a.H + h] * LOG2E : INFINITY;  // +inf => P = 0 for dead rows
        ax[64 + ql] = part;
        ax[128 + ql] = __uint_as_float(attn_rowkey(dkey, bh, (unsigned)a.S4, (unsigned)qpos));
      }
    }
  };
  stage(0, 0);
  ATT_MARK(1); ATT_SET(5, nqt); ATT_SET(6, 1); ATT_SET(7, Sb);
  for (int qt = 0; qt < nqt; ++qt) {
    const int cur = qt & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (qt == 0) ATT_MARK(2);
    if (qt == 1) ATT_MARK(8);
    if (qt + 1 < nqt) stage(qt + 1, cur ^ 1);
    const bf16_t* Qs = smem + cur * (2 * 64 * DH);
    const bf16_t* dOs = Qs + 64 * DH;
    const float* ax = aux_s + cur * 192;
    f32x4 s[4], da[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      s[f] = (f32x4){0.f, 0.f, 0.f, 0.f};
      da[f] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < DH / 32; ++kk) {
        s[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_b128<DH>(Qs, f * 16 + li, kk * 4 + lg), kf[kk], s[f], 0, 0, 0);
        da[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_b128<DH>(dOs, f * 16 + li, kk * 4 + lg), vf[kk], da[f], 0, 0, 0);
      }
    }
    // lane (key = li, g) holds S[q = 16f + 4g + r][key]
    f32x4 pa[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      const f32x4 l2 = *(const f32x4*)(ax + f * 16 + 4 * lg);
      const f32x4 dlt = *(const f32x4*)(ax + 64 + f * 16 + 4 * lg);
      const f32x4 rkf = *(const f32x4*)(ax + 128 + f * 16 + 4 * lg);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = exp2f(s[f][r] * c1 + bias2 - l2[r]);
        bool keep = true;
        if (a.thr16) keep = attn_keep(__float_as_uint(rkf[r]), (unsigned)key_pos, a.thr16);
        pa[f][r] = keep ? p * a.drop_scale : 0.f;                 // A = dropout(P)
        const float dp = keep ? da[f][r] * a.drop_scale : 0.f;
        s[f][r] = p * (dp - dlt[r]);                              // dS
      }
    }
    const bf16x8_t ab0 = pack8(pa[0], pa[1]), ab1 = pack8(pa[2], pa[3]);
    const bf16x8_t sb0 = pack8(s[0], s[1]), sb1 = pack8(s[2], s[3]);
#pragma unroll
    for (int fd = 0; fd < DH / 16; ++fd) {
      dv[fd] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tr<DH>(dOs, 0, fd * 16, lane), ab0, dv[fd], 0, 0, 0);
      dv[fd] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tr<DH>(dOs, 1, fd * 16, lane), ab1, dv[fd], 0, 0, 0);
      dk[fd] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tr<DH>(Qs, 0, fd * 16, lane), sb0, dk[fd], 0, 0, 0);
      dk[fd] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tr<DH>(Qs, 1, fd * 16, lane), sb1, dk[fd], 0, 0, 0);
    }
  }
  __syncthreads();  // stage buffers idle: each wave transposes its blocks in its own corner
  ATT_MARK(3);
  auto krow_ptr = [&](int r, int section) -> bf16_t* {
    const int kl = k0 + wave * 16 + r;
    return kl < Sb ? a.dqkv + (int64_t)(off + kl) * a.ld + section * a.d + h * DH : nullptr;
  };
  store_block16<DH>(smem + wave * 16 * (DH + 8), dk, a.scale, lane, [&](int r) { return krow_ptr(r, 1); });
  store_block16<DH>(smem + wave * 16 * (DH + 8), dv, 1.0f, lane, [&](int r) { return krow_ptr(r, 2); });
}

// ------------------------------------------------------------------------------------------------
// backward, ONE launch, 1-D grid of (q_tiles + k_tiles) slots x (sample, head) pairs, the pair index fastest (see the
// forward kernel: consecutive ids go to different XCDs, so every XCD gets the same mix of live and empty tiles).  Slots
// alternate between the two roles, the longer dK/dV role first: slot 2 i = dK/dV of key tile i, slot 2 i + 1 = dQ of query
// tile i.  The two roles are independent (the dK/dV blocks form delta themselves), so the whole backward of a layer's
// attention is one node of the step graph and its two halves share the CUs instead of running back to back.
// ------------------------------------------------------------------------------------------------
template <int DH>
__global__ __launch_bounds__(256, 2) void attn_bwd_kernel(AttnArgs a, int q_tiles, int k_tiles) {
  __shared__ __attribute__((aligned(16))) bf16_t smem[2 * 2 * 64 * DH + 2 * 3 * 64 * 2];
  ATT_MARK(0);
  const int nbh = a.H * a.B, bh = (int)blockIdx.x % nbh, slot = (int)blockIdx.x / nbh;
  const int b = bh / a.H, h = bh % a.H;
  const int m = min(q_tiles, k_tiles);
  int role, tile;  // role 0 = dK/dV, 1 = dQ
  if (slot < 2 * m) { role = slot & 1; tile = slot >> 1; }
  else { role = q_tiles > k_tiles ? 1 : 0; tile = slot - m; }
  if (role) attn_bwd_dq_block<DH>(a, smem, tile, b, h);
  else attn_bwd_dkv_block<DH>(a, smem, tile, b, h);
#ifdef MMT_GEMM2_INSTR
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  ATT_MARK(4);
  ATT_SET(9, __builtin_amdgcn_s_getreg((31 << 11) | 20));
  ATT_SET(10, __builtin_amdgcn_s_getreg((31 << 11) | 4));
  ATT_SET(11, wall_clock64());
#endif
}

// test helper: materialise the attention dropout keep-mask, uint8 [B,H,S,S] (dense layout only)
__global__ void attn_mask_export_kernel(uint8_t* out, int B, int H, int S, int S4, uint32_t key_in, uint32_t thr16,
                                        const uint32_t* seed_dev) {
  const unsigned key = eff_key(key_in, seed_dev);
  const int64_t n = (int64_t)B * H * S * S;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int k = (int)(i % S), q = (int)((i / S) % S);
    const unsigned bh = (unsigned)(i / ((int64_t)S * S));
    out[i] = thr16 ? (uint8_t)attn_keep(attn_rowkey(key, bh, (unsigned)S4, (unsigned)q), (unsigned)k, thr16) : 1;
  }
}

// ------------------------------------------------------------------------------------------------
static int check_args(const void* qkv, int B, int S, int H, int d) {
  if (!qkv || B <= 0 || S <= 0 || H <= 0) return MMT_ERR_ARG;
  if (d != H * 128 && d != H * 64) return MMT_ERR_ARG;  // head dim 128 (every published video-BERT config) or 64 (BERT-base)
  return 0;
}

// nq queries per sample.  8-wave blocks (128 queries) when a sample has more than 64 queries; MMT_ATTN_FWD_WAVES=4 (lab:
// same-box A/B) keeps the 4-wave blocks of r01-r02.
static int launch_fwd(const AttnArgs& a, int nq, int H, int B, bool dh128, hipStream_t s) {
  static int waves = -1;
  if (waves < 0) {
    const char* e = getenv("MMT_ATTN_FWD_WAVES");
    waves = e ? atoi(e) : 8;
  }
  if (waves == 8 && nq > 64) {
    if (dh128) hipLaunchKernelGGL((attn_fwd_kernel<128, 8>), dim3(((nq + 127) / 128) * H * B), dim3(512), 0, s, a);
    else hipLaunchKernelGGL((attn_fwd_kernel<64, 8>), dim3(((nq + 127) / 128) * H * B), dim3(512), 0, s, a);
  } else {
    if (dh128) hipLaunchKernelGGL((attn_fwd_kernel<128, 4>), dim3(((nq + 63) / 64) * H * B), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((attn_fwd_kernel<64, 4>), dim3(((nq + 63) / 64) * H * B), dim3(256), 0, s, a);
  }
  return (int)hipGetLastError();
}

// tq query tiles (dQ role) + tk key tiles (dK/dV role) in one launch.  MMT_ATTN_BWD_SPLIT=1 (lab: same-box A/B) issues
// the two roles as two launches of the same kernel, the r02 structure.
static int launch_bwd(const AttnArgs& a, int tq, int tk, int H, int B, bool dh128, hipStream_t s) {
  static int split = -1;
  if (split < 0) {
    const char* e = getenv("MMT_ATTN_BWD_SPLIT");
    split = e ? atoi(e) : 0;
  }
  auto go = [&](int q_tiles, int k_tiles) {
    const int gx = (q_tiles + k_tiles) * H * B;
    if (dh128) hipLaunchKernelGGL(attn_bwd_kernel<128>, dim3(gx), dim3(256), 0, s, a, q_tiles, k_tiles);
    else hipLaunchKernelGGL(attn_bwd_kernel<64>, dim3(gx), dim3(256), 0, s, a, q_tiles, k_tiles);
  };
  if (split) {
    go(tq, 0);
    go(0, tk);
  } else {
    go(tq, tk);
  }
  return (int)hipGetLastError();
}

extern "C" int mmt_attn_fwd(const void* qkv, const int32_t* cu_seqlens, const float* mask_bias, void* ctx,
                            float* lse, int B, int S, int H, int d, float scale, uint32_t drop_key,
                            uint32_t thr16, float drop_scale, const uint32_t* seed_dev, const int32_t* row_index, void* stream) {
  if (int e = check_args(qkv, B, S, H, d)) return e;
  if (!mask_bias || !ctx || !lse) return MMT_ERR_ARG;
  AttnArgs a = {};
  a.qkv = (const bf16_t*)qkv; a.ld = 3 * (int64_t)d; a.cu = cu_seqlens; a.S_dense = S; a.mask_bias = mask_bias;
  a.ctx = (bf16_t*)ctx; a.ldc = d; a.lse = lse; a.H = H; a.d = d; a.B = B; a.scale = scale;
  a.drop_key = drop_key; a.thr16 = thr16; a.drop_scale = drop_scale; a.S4 = (S + 3) & ~3; a.seed_dev = seed_dev;
  a.row_index = row_index;
  return launch_fwd(a, S, H, B, d == H * 128, (hipStream_t)stream);
}

extern "C" int mmt_attn_bwd(const void* qkv, const int32_t* cu_seqlens, const float* mask_bias, const void* ctx,
                            const float* lse, const void* dctx, void* dqkv, float* delta, int B, int S, int H,
                            int d, float scale, uint32_t drop_key, uint32_t thr16, float drop_scale,
                            const uint32_t* seed_dev, const int32_t* row_index, void* stream) {
  if (int e = check_args(qkv, B, S, H, d)) return e;
  if (!mask_bias || !ctx || !lse || !dctx || !dqkv || !delta) return MMT_ERR_ARG;
  AttnArgs a = {};
  a.qkv = (const bf16_t*)qkv; a.ld = 3 * (int64_t)d; a.cu = cu_seqlens; a.S_dense = S; a.mask_bias = mask_bias;
  a.ctx = (bf16_t*)ctx; a.ldc = d; a.lse = (float*)lse; a.dctx = (const bf16_t*)dctx; a.dqkv = (bf16_t*)dqkv;
  a.delta = delta; a.H = H; a.d = d; a.B = B; a.scale = scale;
  a.drop_key = drop_key; a.thr16 = thr16; a.drop_scale = drop_scale; a.S4 = (S + 3) & ~3; a.seed_dev = seed_dev;
  a.row_index = row_index;
  const int tiles = (S + 63) / 64;
  return launch_bwd(a, tiles, tiles, H, B, d == H * 128, (hipStream_t)stream);
}

// Query-subset variants: only the rows qsel[b*nq + i] act as queries (all rows of a sample remain keys/values).
// ctx / lse / dctx / delta are compact [B*nq, .]; dqkv is the full [rows, 3d] buffer and must be ZERO on entry for the
// Q section of non-selected rows (the kernels write dQ of the selected rows and dK / dV of every row).
extern "C" int mmt_attn_fwd_rows(const void* qkv, const int32_t* cu_seqlens, const float* mask_bias, const int32_t* qsel,
                                 int nq, void* ctx, float* lse, int B, int S, int H, int d, float scale,
                                 uint32_t drop_key, uint32_t thr16, float drop_scale, const uint32_t* seed_dev, const int32_t* row_index, void* stream) {
  if (int e = check_args(qkv, B, S, H, d)) return e;
  if (!mask_bias || !ctx || !lse || !qsel || nq <= 0) return MMT_ERR_ARG;
  AttnArgs a = {};
  a.qkv = (const bf16_t*)qkv; a.ld = 3 * (int64_t)d; a.cu = cu_seqlens; a.S_dense = S; a.mask_bias = mask_bias;
  a.ctx = (bf16_t*)ctx; a.ldc = d; a.lse = lse; a.H = H; a.d = d; a.B = B; a.scale = scale;
  a.drop_key = drop_key; a.thr16 = thr16; a.drop_scale = drop_scale; a.S4 = (S + 3) & ~3; a.seed_dev = seed_dev;
  a.row_index = row_index;
  a.qsel = qsel; a.nq = nq;
  return launch_fwd(a, nq, H, B, d == H * 128, (hipStream_t)stream);
}

extern "C" int mmt_attn_bwd_rows(const void* qkv, const int32_t* cu_seqlens, const float* mask_bias, const int32_t* qsel,
                                 int nq, const void* ctx, const float* lse, const void* dctx, void* dqkv, float* delta,
                                 int B, int S, int H, int d, float scale, uint32_t drop_key, uint32_t thr16,
                                 float drop_scale, const uint32_t* seed_dev, const int32_t* row_index, void* stream) {
  if (int e = check_args(qkv, B, S, H, d)) return e;
  if (!mask_bias || !ctx || !lse || !dctx || !dqkv || !delta || !qsel || nq <= 0) return MMT_ERR_ARG;
  AttnArgs a = {};
  a.qkv = (const bf16_t*)qkv; a.ld = 3 * (int64_t)d; a.cu = cu_seqlens; a.S_dense = S; a.mask_bias = mask_bias;
  a.ctx = (bf16_t*)ctx; a.ldc = d; a.lse = (float*)lse; a.dctx = (const bf16_t*)dctx; a.dqkv = (bf16_t*)dqkv;
  a.delta = delta; a.H = H; a.d = d; a.B = B; a.scale = scale;
  a.drop_key = drop_key; a.thr16 = thr16; a.drop_scale = drop_scale; a.S4 = (S + 3) & ~3; a.seed_dev = seed_dev;
  a.row_index = row_index;
  a.qsel = qsel; a.nq = nq;
  return launch_bwd(a, (nq + 63) / 64, (S + 63) / 64, H, B, d == H * 128, (hipStream_t)stream);
}

extern "C" int mmt_attn_dropout_mask(uint8_t* out, int B, int H, int S, uint32_t drop_key, uint32_t thr16,
                                     const uint32_t* seed_dev, void* stream) {
  if (!out) return MMT_ERR_ARG;
  hipLaunchKernelGGL(attn_mask_export_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream, out, B, H, S,
                     (S + 3) & ~3, drop_key, thr16, seed_dev);
  return (int)hipGetLastError();
}
