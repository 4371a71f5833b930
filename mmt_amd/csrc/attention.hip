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
#include "attn_sched.h"
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

// ---- LDS access from inline asm ----------------------------------------------------------------------
// From the first LDS-DMA of a block until its final drain, EVERY LDS access of these kernels is inline asm.  hipcc cannot
// tell an LDS location a global_load_lds writes from one a later ds_read touches, so it puts s_waitcnt vmcnt(0) in front of
// every LDS access it can see that follows an LDS-DMA -- which made each key / value tile wait for the prefetch of the NEXT
// one (r03 ISA: vmcnt(0) between the DMA issue and the first fragment read of every iteration; a block was a chain of
// exposed memory round trips, 7-11 k cycles per 64-row tile for ~2 k cycles of work; tools/attn_budget.py).  Asm accesses are
// invisible to that pass; the price is hand-counted s_waitcnt lgkmcnt(N) (N <= 15 on gfx9-family encodings: at most 16
// reads are kept in flight) with the consumers tied to the wait (TIE) so that nothing is scheduled above it.
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(uintptr_t)LDS_PTR(p); }
__device__ __forceinline__ void lds_r128(u32x4& d, unsigned addr) { asm volatile("ds_read_b128 %0, %1" : "=v"(d) : "v"(addr)); }
// the two 4-row halves of a transposed 8 x 16 fragment: rows r0.. and r0 + 16.. of the same 16-byte chunk (the swizzle
// repeats every 16 rows, so the second address is the first + 16 rows: an instruction immediate)
template <int HALF> __device__ __forceinline__ void lds_rtr2(u32x2& lo, u32x2& hi, unsigned addr) {
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(addr));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(addr), "n"(HALF));
}
__device__ __forceinline__ void lds_w32(unsigned addr, unsigned v) { asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
template <int N> __device__ __forceinline__ void lgkm_wait() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }
#define TIE(x) asm volatile("" : "+v"(x))
// max / sum over the four lanes {l, l ^ 16, l ^ 32, l ^ 48} by v_permlane16_swap / v_permlane32_swap: __shfl_xor compiles to
// ds_bpermute_b32, an LGKM operation whose result the compiler waits for with lgkmcnt(0) -- i.e. for every fragment read
// still in flight
__device__ __forceinline__ float red4_max(float x) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  const float m = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
  const auto q = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);
  return fmaxf(__uint_as_float(q[0]), __uint_as_float(q[1]));
}
__device__ __forceinline__ float red4_sum(float x) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  const float m = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  const auto q = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);
  return __uint_as_float(q[0]) + __uint_as_float(q[1]);
}
__device__ __forceinline__ bf16x8_t as_bf8(const u32x4& v) { return __builtin_bit_cast(bf16x8_t, v); }
__device__ __forceinline__ bf16x8_t join_bf8(const u32x2& lo, const u32x2& hi) {
  const u32x4 v = {lo[0], lo[1], hi[0], hi[1]};
  return __builtin_bit_cast(bf16x8_t, v);
}
// per-lane byte offset (inside a 64 x DH tile) of the row-major fragment for tile row 16 f + (lane & 15), 16-byte chunk
// 4 kk + (lane >> 4): offset(f, kk) = (frag_b128_off ^ (kk << 6)) + f * 16 rows (the swizzle depends on lane & 15 only)
template <int DH> __device__ __forceinline__ unsigned frag_b128_off(int lane) {
  const int li = lane & 15, lg = lane >> 4;
  return (unsigned)(li * DH * 2 + ((lg ^ swz<DH>(li)) << 4));
}
// per-lane byte offset of the transposed fragment (k-chunk ks = 0, columns 0..15): fragment (ks, fd) is at
// (frag_tr_off ^ (fd << 5)) + ks * 32 rows; second half + 16 rows (see lds_rtr2).  Lane (t = lane & 15, g = lane >> 4) holds
// the contraction rows kappa(g, .) = 4 g + (t >> 2) + {0, 16} of column 4 (t & 3) .. + 3 -- the permutation the packed
// probabilities of the first MFMA come out in.
template <int DH> __device__ __forceinline__ unsigned frag_tr_off(int lane) {
  const int t = lane & 15, g = lane >> 4;
  const int col = 4 * (t & 3), r0 = 4 * g + (t >> 2);
  return (unsigned)((r0 * DH + (((col >> 3) ^ swz<DH>(r0)) << 3) + (col & 7)) * 2);
}

__device__ __forceinline__ bf16x8_t pack8(const f32x4& a, const f32x4& b) {
  u32x4 u = {pack_bf2(a[0], a[1]), pack_bf2(a[2], a[3]), pack_bf2(b[0], b[1]), pack_bf2(b[2], b[3])};
  return __builtin_bit_cast(bf16x8_t, u);
}

__device__ __forceinline__ unsigned attn_rowkey(unsigned key, unsigned bh, unsigned S4, unsigned q) {
  return mix32(key ^ ((bh * S4 + q) * 0x9e3779b9U));
}
// Dropout uniform of element (row key, key position k): both coordinates are hashed ONCE per row / per key (mix32, two
// quarter-rate 32-bit multiplies each: O(S) per block) and the S^2 per-element work is three full-rate instructions --
// u16 = bits 16..31 of the 24-bit product (rowkey ^ keymix) * C (v_xor, v_mul_u32_u24, v_lshrrev): every one of those bits
// depends on all lower bits of the XOR through the carries.  (r01-r03 hashed every (row, key pair) with a full mix32:
// 12 instructions per element with two slow multiplies, a third of the forward kernel's issue slots.)
__device__ __forceinline__ unsigned attn_keymix(unsigned k) { return mix32(k * 0x85ebca6bU + 0x1b873593U); }
__device__ __forceinline__ bool attn_keep_mixed(unsigned rowkey, unsigned keymix, unsigned thr16) {
  return (__umul24(rowkey ^ keymix, 0x9e3779U) >> 16) >= thr16;
}
__device__ __forceinline__ bool attn_keep(unsigned rowkey, unsigned k, unsigned thr16) {
  return attn_keep_mixed(rowkey, attn_keymix(k), thr16);
}
// raw v_exp_f32 (fast_exp2() adds a denormal-range rescue: compare + ldexp per call; arguments here are <= 0 or -inf)
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

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
  const float* dparts;                 // [rows, d/64] bwd in: sums of dO * O over 64-column groups (delta = their sum per head)
  int H, d, B; float scale;
  uint32_t drop_key, thr16; float drop_scale; int S4;
  const uint32_t* seed_dev;
  // Query subset (last encoder layer: only the rows that are read out need a context vector).  qsel[b*nq + i] is
  // the row of sample b's i-th selected query; ctx / lse / dctx / dparts are then COMPACT [B*nq, .] buffers, while
  // qkv / dqkv keep the full token layout (every key still participates).
  const int32_t* qsel; int nq;
  // token packing: row_index[row] = b * S_dense + original position (nullable: rows are dense, position = row - b*S)
  const int32_t* row_index;
  // backward only (nullable): block order built by attn_schedule_block (attn_sched.h), one item per block of the grid
  const int32_t* work;
};

// original position of (packed) row `row` of sample b
__device__ __forceinline__ int orig_pos(const AttnArgs& a, int b, int off, int row) {
  return a.row_index ? a.row_index[row] - b * a.S_dense : row - off;
}
// keep flags of four consecutive (packed) keys whose premixed position words are km[0..3]
__device__ __forceinline__ void keep4_keys(unsigned rowkey, const u32x4& km, unsigned thr16, bool keep[4]) {
#pragma unroll
  for (int e = 0; e < 4; ++e) keep[e] = attn_keep_mixed(rowkey, km[e], thr16);
}

// LDS of one block: two stages of [tile A | tile B] (64 x DH bf16 each) + three per-row arrays of SP = round_up(S, 64)
// words (forward / dQ: key-mask bias, original key position; dK/dV: lse, delta, dropout row key of every query).
template <int DH> __host__ __device__ constexpr int attn_stage_bytes() { return 2 * 64 * DH * 2; }
static inline size_t attn_lds_bytes(int DH, int S) { return (size_t)2 * 2 * 64 * DH * 2 + (size_t)3 * ((S + 63) & ~63) * 4; }

// ------------------------------------------------------------------------------------------------
// forward: 1-D grid of q tiles (16 NW queries) x (sample, head) pairs, the pair index fastest: consecutive block ids -- which
// the dispatcher deals out to the 8 XCDs in turn -- are the same q tile of different pairs.  (With (q tile, head, sample) as
// grid (x, y, z) every XCD received ONE q-tile index, and under token packing the later q tiles are mostly empty: half the
// XCDs idled, r04.)  NW = 8 (one block per CU) stages every key / value tile once per 128 queries.
// Memory schedule: the block's dependent chain is  cu_seqlens -> {LDS-DMA of key tiles 0 AND 1, Q rows, key-mask bias and
// key positions of the WHOLE sample} -> ONE wait -> tiles 0, 1 back to back; tile kt + 2 is requested when tile kt is done.
// ------------------------------------------------------------------------------------------------
template <int DH, int NW>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) void attn_fwd_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(256))) unsigned char att_smem[];
  constexpr int ROWB = DH * 2, TILEB = 64 * ROWB, STAGEB = 2 * TILEB, KK = DH / 32, FD = DH / 16, NG = 2 * FD / 4;
  constexpr int NDMA = 2 * (TILEB / 1024) / NW;  // LDS-DMA instructions per wave per stage
  bf16_t* smem = (bf16_t*)att_smem;
  const int SP = (a.S_dense + 63) & ~63;
  float* bias_s = (float*)(att_smem + 2 * STAGEB);  // [SP] key-mask bias (x log2 e), -inf beyond the sample
  int* kpos_s = (int*)(bias_s + SP);                 // [SP] original positions of the keys
  ATT_MARK(0);
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

  auto stage = [&](int kt) {
    bf16_t* base = smem + (kt & 1) * (STAGEB / 2);
    stage64<DH, NW>(Kg, a.ld, off + kt * 64, row_last, base, wave, lane);
    stage64<DH, NW>(Vg, a.ld, off + kt * 64, row_last, base + 64 * DH, wave, lane);
  };
  stage(0);
  if (nkt > 1) stage(1);
  // Every global load of the prologue goes out before the first of them is consumed (one exposed round trip, shared with
  // the LDS-DMA of the first two tiles): r03's prologue was a chain of four to five dependent round trips.
  constexpr int NT = NW * 64;
  const int qi = q0 + wave * 16 + li;
  const int qic = min(qi, nqs - 1);
  const int qrow = a.qsel ? a.qsel[b * a.nq + qic] : off + qic;
  const int crow = a.qsel ? b * a.nq + qic : qrow;  // row in ctx / lse
  bf16x8_t qf[KK];
#pragma unroll
  for (int kk = 0; kk < KK; ++kk)
    qf[kk] = *(const bf16x8_t*)(a.qkv + (int64_t)qrow * a.ld + h * DH + kk * 32 + lg * 8);
  const int kc0 = min(tid, Sb - 1);
  const float mb0 = a.mask_bias[off + kc0];
  int rp0 = off + kc0 + b * a.S_dense - off, qpr = qrow + b * a.S_dense - off;  // (dense rows: position = row - off)
  if (a.row_index) { rp0 = a.row_index[off + kc0]; qpr = a.row_index[qrow]; }
  const unsigned dkey = eff_key(a.drop_key, a.seed_dev);
  if (tid < nkt * 64) {
    lds_w32(lds_addr(bias_s + tid), __float_as_uint(tid < Sb ? mb0 * LOG2E : -INFINITY));
    lds_w32(lds_addr(kpos_s + tid), attn_keymix((unsigned)(rp0 - b * a.S_dense)));
  }
  for (int k = tid + NT; k < nkt * 64; k += NT) {  // (samples longer than the block is wide)
    const float bv = k < Sb ? a.mask_bias[off + k] * LOG2E : -INFINITY;
    const int kp = k < Sb ? orig_pos(a, b, off, off + k) : k;
    lds_w32(lds_addr(bias_s + k), __float_as_uint(bv));
    lds_w32(lds_addr(kpos_s + k), attn_keymix((unsigned)kp));
  }
  const int q_local = qpr - b * a.S_dense;
  const unsigned rowkey = attn_rowkey(dkey, (unsigned)(b * a.H + h), (unsigned)a.S4, (unsigned)q_local);
  const float c1 = a.scale * LOG2E;
  const unsigned ka = frag_b128_off<DH>(lane), va = frag_tr_off<DH>(lane);
  const unsigned s_base = lds_addr(smem), bias_a = lds_addr(bias_s) + 16 * lg, kpos_a = lds_addr(kpos_s) + 16 * lg;
#pragma unroll
  for (int kk = 0; kk < KK; ++kk) asm volatile("" : "+v"(qf[kk]));  // (the compiler's wait for these loads: here, not in the loop)

  f32x4 o[FD];
#pragma unroll
  for (int i = 0; i < FD; ++i) o[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float m_run = NEG_BIG, l_run = 0.f;
  ATT_MARK(1); ATT_SET(5, nkt); ATT_SET(6, 2); ATT_SET(7, Sb);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  ATT_MARK(2);

  for (int kt = 0; kt < nkt; ++kt) {
    if (kt >= 2) {  // tiles 0 and 1 landed in the prologue; later ones were requested two iterations ago
      if (kt + 1 < nkt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
    const unsigned kb = s_base + (kt & 1) * STAGEB + ka, vb = s_base + (kt & 1) * STAGEB + TILEB + va;
    // key-mask bias (+ key positions under dropout) of this lane's 16 keys, then the key fragments, kk by kk
    u32x4 bv[4], kp[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) lds_r128(bv[f], bias_a + (unsigned)((kt * 64 + f * 16) * 4));
    if (a.thr16) {
#pragma unroll
      for (int f = 0; f < 4; ++f) lds_r128(kp[f], kpos_a + (unsigned)((kt * 64 + f * 16) * 4));
    }
    u32x4 kr[KK][4];
#pragma unroll
    for (int f = 0; f < 4; ++f) lds_r128(kr[0][f], kb + (unsigned)(f * 16 * ROWB));
#pragma unroll
    for (int f = 0; f < 4; ++f) lds_r128(kr[1][f], (kb ^ 64u) + (unsigned)(f * 16 * ROWB));
    f32x4 s[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) s[f] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
      // in flight behind fragment kk: kk + 1 (if any).  (bias / positions were issued first: they have landed as well.)
      if (kk + 1 < KK) lgkm_wait<4>(); else lgkm_wait<0>();
#pragma unroll
      for (int f = 0; f < 4; ++f) TIE(kr[kk][f]);
      if (kk + 2 < KK) {
#pragma unroll
        for (int f = 0; f < 4; ++f) lds_r128(kr[kk + 2][f], (kb ^ (unsigned)((kk + 2) << 6)) + (unsigned)(f * 16 * ROWB));
      }
#pragma unroll
      for (int f = 0; f < 4; ++f) s[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf8(kr[kk][f]), qf[kk], s[f], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    // value fragments (transposed): groups of 4 fragments = 8 reads, two groups in flight under the softmax
    u32x2 vlo[NG][4], vhi[NG][4];
    auto vissue = [&](int g) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int idx = g * 4 + j, ks = idx / FD, fd = idx % FD;
        lds_rtr2<16 * ROWB>(vlo[g][j], vhi[g][j], ((vb + (unsigned)(ks * 32 * ROWB)) ^ (unsigned)(fd << 5)));
      }
    };
    vissue(0);
    if (NG > 1) vissue(1);
#pragma unroll
    for (int f = 0; f < 4; ++f) { TIE(bv[f]); }
    float mt = NEG_BIG;
#pragma unroll
    for (int f = 0; f < 4; ++f) {
#pragma unroll
      for (int r = 0; r < 4; ++r) { s[f][r] = s[f][r] * c1 + __uint_as_float(bv[f][r]); mt = fmaxf(mt, s[f][r]); }
    }
    mt = red4_max(mt);
    const float m_new = fmaxf(m_run, mt);
    const float alpha = fast_exp2(m_run - m_new);
    float psum = 0.f;
#pragma unroll
    for (int f = 0; f < 4; ++f) {
#pragma unroll
      for (int r = 0; r < 4; ++r) { s[f][r] = fast_exp2(s[f][r] - m_new); psum += s[f][r]; }
      if (a.thr16) {
        TIE(kp[f]);
        bool kq[4];
        keep4_keys(rowkey, kp[f], a.thr16, kq);
#pragma unroll
        for (int r = 0; r < 4; ++r) s[f][r] = kq[r] ? s[f][r] * a.drop_scale : 0.f;
      }
    }
    psum = red4_sum(psum);
    l_run = l_run * alpha + psum;
    m_run = m_new;
#pragma unroll
    for (int i = 0; i < FD; ++i) o[i] *= alpha;
    const bf16x8_t pb[2] = {pack8(s[0], s[1]), pack8(s[2], s[3])};
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      if (g + 1 < NG) lgkm_wait<8>(); else lgkm_wait<0>();
#pragma unroll
      for (int j = 0; j < 4; ++j) { TIE(vlo[g][j]); TIE(vhi[g][j]); }
      if (g + 2 < NG) vissue(g + 2);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int idx = g * 4 + j, ks = idx / FD, fd = idx % FD;
        o[fd] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(join_bf8(vlo[g][j], vhi[g][j]), pb[ks], o[fd], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (kt + 2 < nkt) {  // every wave is done with this stage: request tile kt + 2 into it
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      stage(kt + 2);
    }
  }
  // Output through LDS (store_block16): stored directly, a lane's result is 8 partial-line writes 32 B apart.
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();
  ATT_MARK(3);
  if (qi < nqs && lg == 0) a.lse[(int64_t)crow * a.H + h] = (m_run + log2f(l_run)) * LN2;
  store_block16<DH>(smem + wave * 16 * (DH + 8), o, 1.0f / l_run, lane, [&](int r) -> bf16_t* {
    const int qr = q0 + wave * 16 + r;
    if (qr >= nqs) return nullptr;
    return a.ctx + (a.qsel ? (int64_t)b * a.nq + qr : (int64_t)off + qr) * a.ldc + h * DH;
  });
#ifdef MMT_GEMM2_INSTR
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  ATT_MARK(4);
  ATT_SET(9, __builtin_amdgcn_s_getreg((31 << 11) | 20));
  ATT_SET(10, __builtin_amdgcn_s_getreg((31 << 11) | 4));
  ATT_SET(11, wall_clock64());
#endif
}

// ------------------------------------------------------------------------------------------------
// backward, dQ role.  Same tiling and memory schedule as the forward: one block per (q tile of 64, sample, head).
//   dA^T[key][q] = V . dO^T ; dS = P o (keep*dA*sc - delta) ; dQ^T[d][q] += K^T . dS^T
// delta[q] = rowsum(dO o O) arrives as per-64-column partial sums (dparts), written by the epilogue of the GEMM that
// produced dO (MmtEpilogue.dot_out) or by attn_delta_kernel below.
// ------------------------------------------------------------------------------------------------
template <int DH>
__device__ __forceinline__ void attn_bwd_dq_block(const AttnArgs& a, unsigned char* att_smem, int bx, int b, int h, int off, int Sb) {
  constexpr int NW = 4, ROWB = DH * 2, TILEB = 64 * ROWB, STAGEB = 2 * TILEB, KK = DH / 32, FD = DH / 16, NG = 2 * FD / 4;
  constexpr int NDMA = 2 * (TILEB / 1024) / NW;
  bf16_t* smem = (bf16_t*)att_smem;
  const int SP = (a.S_dense + 63) & ~63;
  float* bias_s = (float*)(att_smem + 2 * STAGEB);
  int* kpos_s = (int*)(bias_s + SP);
  const int q0 = bx * 64;
  const int nqs = a.qsel ? a.nq : Sb;
  if (q0 >= nqs || Sb <= 0) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
  const int row_last = off + Sb - 1;
  const int nkt = (Sb + 63) >> 6;
  const bf16_t* Kg = a.qkv + a.d + h * DH;
  const bf16_t* Vg = a.qkv + 2 * a.d + h * DH;
  auto stage = [&](int kt) {
    bf16_t* base = smem + (kt & 1) * (STAGEB / 2);
    stage64<DH, NW>(Kg, a.ld, off + kt * 64, row_last, base, wave, lane);
    stage64<DH, NW>(Vg, a.ld, off + kt * 64, row_last, base + 64 * DH, wave, lane);
  };
  stage(0);
  if (nkt > 1) stage(1);
  constexpr int NT = NW * 64;
  const int qi = q0 + wave * 16 + li;
  const int qic = min(qi, nqs - 1);
  const int qrow = a.qsel ? a.qsel[b * a.nq + qic] : off + qic;
  const int crow = a.qsel ? b * a.nq + qic : qrow;  // row in ctx / dctx / lse / dparts
  // (all global loads first, see the forward kernel)
  bf16x8_t qf[KK], dof[KK];
#pragma unroll
  for (int kk = 0; kk < KK; ++kk) {
    qf[kk] = *(const bf16x8_t*)(a.qkv + (int64_t)qrow * a.ld + h * DH + kk * 32 + lg * 8);
    dof[kk] = *(const bf16x8_t*)(a.dctx + (int64_t)crow * a.ldc + h * DH + kk * 32 + lg * 8);
  }
  float dpart[DH / 64];
#pragma unroll
  for (int p = 0; p < DH / 64; ++p) dpart[p] = a.dparts[(int64_t)crow * (a.d / 64) + h * (DH / 64) + p];
  const float lse_q = a.lse[(int64_t)crow * a.H + h];
  const int kc0 = min(tid, Sb - 1);
  const float mb0 = a.mask_bias[off + kc0];
  int rp0 = kc0 + b * a.S_dense, qpr = qrow + b * a.S_dense - off;
  if (a.row_index) { rp0 = a.row_index[off + kc0]; qpr = a.row_index[qrow]; }
  const unsigned dkey = eff_key(a.drop_key, a.seed_dev);
  if (a.qsel && bx == 0) {
    // query-subset mode: dQ of the non-selected rows is zero -- this block (sample b, head h) clears its DH columns
    // of every row of the sample; the selected rows are overwritten at the end (after the loop's barriers)
    for (int e = tid; e < Sb * (DH / 8); e += 256) {
      u32x4 z = {0, 0, 0, 0};
      *(u32x4*)(a.dqkv + (int64_t)(off + e / (DH / 8)) * a.ld + h * DH + (e % (DH / 8)) * 8) = z;
    }
  }
  if (tid < nkt * 64) {
    lds_w32(lds_addr(bias_s + tid), __float_as_uint(tid < Sb ? mb0 * LOG2E : -INFINITY));
    lds_w32(lds_addr(kpos_s + tid), attn_keymix((unsigned)(rp0 - b * a.S_dense)));
  }
  for (int k = tid + NT; k < nkt * 64; k += NT) {
    const float bv = k < Sb ? a.mask_bias[off + k] * LOG2E : -INFINITY;
    const int kp = k < Sb ? orig_pos(a, b, off, off + k) : k;
    lds_w32(lds_addr(bias_s + k), __float_as_uint(bv));
    lds_w32(lds_addr(kpos_s + k), attn_keymix((unsigned)kp));
  }
  float dl = 0.f;
#pragma unroll
  for (int p = 0; p < DH / 64; ++p) dl += dpart[p];
  const float lse2 = lse_q * LOG2E;
  const int q_local = qpr - b * a.S_dense;
  const unsigned rowkey = attn_rowkey(dkey, (unsigned)(b * a.H + h), (unsigned)a.S4, (unsigned)q_local);
  const float c1 = a.scale * LOG2E;
  const unsigned ka = frag_b128_off<DH>(lane), va = frag_tr_off<DH>(lane);
  const unsigned s_base = lds_addr(smem), bias_a = lds_addr(bias_s) + 16 * lg, kpos_a = lds_addr(kpos_s) + 16 * lg;
#pragma unroll
  for (int kk = 0; kk < KK; ++kk) { asm volatile("" : "+v"(qf[kk])); asm volatile("" : "+v"(dof[kk])); }

  f32x4 o[FD];
#pragma unroll
  for (int i = 0; i < FD; ++i) o[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  ATT_MARK(1); ATT_SET(5, nkt); ATT_SET(6, 0); ATT_SET(7, Sb);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  ATT_MARK(2);

  for (int kt = 0; kt < nkt; ++kt) {
    if (kt == 1) ATT_MARK(8);
    if (kt >= 2) {
      if (kt + 1 < nkt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
    const unsigned kb = s_base + (kt & 1) * STAGEB + ka, ktb = s_base + (kt & 1) * STAGEB + va;
    u32x4 bv[4], kp[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) lds_r128(bv[f], bias_a + (unsigned)((kt * 64 + f * 16) * 4));
    if (a.thr16) {
#pragma unroll
      for (int f = 0; f < 4; ++f) lds_r128(kp[f], kpos_a + (unsigned)((kt * 64 + f * 16) * 4));
    }
    // row-major fragments of the key tile (-> S) and of the value tile (-> dA), one k-chunk (8 reads) per batch
    u32x4 R[KK][8];
    auto bissue = [&](int kk) {
#pragma unroll
      for (int f = 0; f < 4; ++f) lds_r128(R[kk][f], (kb ^ (unsigned)(kk << 6)) + (unsigned)(f * 16 * ROWB));
#pragma unroll
      for (int f = 0; f < 4; ++f) lds_r128(R[kk][4 + f], (kb ^ (unsigned)(kk << 6)) + (unsigned)(TILEB + f * 16 * ROWB));
    };
    bissue(0);
    f32x4 s[4], da[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) { s[f] = (f32x4){0.f, 0.f, 0.f, 0.f}; da[f] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
      if (kk + 1 < KK) { bissue(kk + 1); lgkm_wait<8>(); } else { lgkm_wait<0>(); }
#pragma unroll
      for (int f = 0; f < 8; ++f) TIE(R[kk][f]);
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        s[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf8(R[kk][f]), qf[kk], s[f], 0, 0, 0);
        da[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf8(R[kk][4 + f]), dof[kk], da[f], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // transposed key fragments for dQ: two groups in flight under the dS arithmetic
    u32x2 tlo[NG][4], thi[NG][4];
    auto tissue = [&](int g) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int idx = g * 4 + j, ks = idx / FD, fd = idx % FD;
        lds_rtr2<16 * ROWB>(tlo[g][j], thi[g][j], ((ktb + (unsigned)(ks * 32 * ROWB)) ^ (unsigned)(fd << 5)));
      }
    };
    tissue(0);
    if (NG > 1) tissue(1);
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      TIE(bv[f]);
      bool kq[4] = {true, true, true, true};
      if (a.thr16) {
        TIE(kp[f]);
        keep4_keys(rowkey, kp[f], a.thr16, kq);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = fast_exp2(s[f][r] * c1 + __uint_as_float(bv[f][r]) - lse2);
        const float dp = kq[r] ? da[f][r] * a.drop_scale : 0.f;
        s[f][r] = p * (dp - dl);  // dS
      }
    }
    const bf16x8_t sb[2] = {pack8(s[0], s[1]), pack8(s[2], s[3])};
#ifdef MMT_ATTN_LAB_DS_SPLIT  // lab: dS = hi + lo, two bf16 MFMAs -- does bf16 rounding of dS explain the q / k error?
    bf16x8_t sl[2];
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
      f32x4 r0, r1;
#pragma unroll
      for (int r = 0; r < 4; ++r) { r0[r] = s[2 * h2][r] - (float)sb[h2][r]; r1[r] = s[2 * h2 + 1][r] - (float)sb[h2][4 + r]; }
      sl[h2] = pack8(r0, r1);
    }
#endif
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      if (g + 1 < NG) lgkm_wait<8>(); else lgkm_wait<0>();
#pragma unroll
      for (int j = 0; j < 4; ++j) { TIE(tlo[g][j]); TIE(thi[g][j]); }
      if (g + 2 < NG) tissue(g + 2);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int idx = g * 4 + j, ks = idx / FD, fd = idx % FD;
        o[fd] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(join_bf8(tlo[g][j], thi[g][j]), sb[ks], o[fd], 0, 0, 0);
#ifdef MMT_ATTN_LAB_DS_SPLIT
        o[fd] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(join_bf8(tlo[g][j], thi[g][j]), sl[ks], o[fd], 0, 0, 0);
#endif
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (kt + 2 < nkt) {
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      stage(kt + 2);
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
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
// backward, dK / dV role.  One block per (key tile of 64, sample, head); each wave owns 16 keys, loops over q tiles.
//   S[q][key] = Q . K^T (a = Q frag from LDS, b = K frag in registers) ; dA = dO . V^T
//   dV^T[d][key] += dO^T . A ; dK^T[d][key] += Q^T . dS     (a = transpose reads of the dO / Q tiles)
// lse, delta and the dropout row key of EVERY query of the sample go to LDS once, in the prologue.
// ------------------------------------------------------------------------------------------------
template <int DH>
__device__ __forceinline__ void attn_bwd_dkv_block(const AttnArgs& a, unsigned char* att_smem, int bx, int b, int h, int off, int Sb) {
  constexpr int NW = 4, ROWB = DH * 2, TILEB = 64 * ROWB, STAGEB = 2 * TILEB, KK = DH / 32, FD = DH / 16, NG = 2 * FD / 4;
  constexpr int NDMA = 2 * (TILEB / 1024) / NW;
  bf16_t* smem = (bf16_t*)att_smem;
  const int SP = (a.S_dense + 63) & ~63;
  float* lse_s = (float*)(att_smem + 2 * STAGEB);  // [SP] lse (x log2 e), +inf for dead rows
  float* dlt_s = lse_s + SP;                        // [SP] delta
  float* rk_s = dlt_s + SP;                         // [SP] dropout row key (bits)
  const int k0 = bx * 64;
  if (k0 >= Sb) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
  const int row_last = off + Sb - 1;
  const int nqs = a.qsel ? a.nq : Sb;
  const int nqt = (nqs + 63) >> 6;
  const bf16_t* Qg = a.qkv + h * DH;
  const bf16_t* dOg = a.dctx + h * DH;
  auto stage = [&](int qt) {
    bf16_t* base = smem + (qt & 1) * (STAGEB / 2);
    if (a.qsel) {
      stage64_sel<DH>(Qg, a.ld, a.qsel + b * a.nq, qt * 64, a.nq, base, wave, lane);
      stage64<DH, NW>(dOg, a.ldc, b * a.nq + qt * 64, b * a.nq + a.nq - 1, base + 64 * DH, wave, lane);
    } else {
      stage64<DH, NW>(Qg, a.ld, off + qt * 64, row_last, base, wave, lane);
      stage64<DH, NW>(dOg, a.ldc, off + qt * 64, row_last, base + 64 * DH, wave, lane);
    }
  };
  stage(0);
  if (nqt > 1) stage(1);
  constexpr int NT = NW * 64;
  const int key_local = k0 + wave * 16 + li;
  const bool key_ok = key_local < Sb;
  const int krow = off + min(key_local, Sb - 1);
  // (all global loads first, see the forward kernel)
  bf16x8_t kf[KK], vf[KK];
#pragma unroll
  for (int kk = 0; kk < KK; ++kk) {
    kf[kk] = *(const bf16x8_t*)(a.qkv + (int64_t)krow * a.ld + a.d + h * DH + kk * 32 + lg * 8);
    vf[kk] = *(const bf16x8_t*)(a.qkv + (int64_t)krow * a.ld + 2 * a.d + h * DH + kk * 32 + lg * 8);
  }
  const float mbk = a.mask_bias[krow];
  const int qc0 = min(tid, nqs - 1);
  const int qrow0 = a.qsel ? a.qsel[b * a.nq + qc0] : off + qc0;   // row of the first query this thread files
  const int mrow0 = a.qsel ? b * a.nq + qc0 : off + qc0;           // its row in lse / dparts
  float dp0[DH / 64];
#pragma unroll
  for (int p = 0; p < DH / 64; ++p) dp0[p] = a.dparts[(int64_t)mrow0 * (a.d / 64) + h * (DH / 64) + p];
  const float ls0 = a.lse[(int64_t)mrow0 * a.H + h];
  int kpr = krow + b * a.S_dense - off, qpr0 = qrow0 + b * a.S_dense - off;
  if (a.row_index) { kpr = a.row_index[krow]; qpr0 = a.row_index[qrow0]; }
  const unsigned dkey = eff_key(a.drop_key, a.seed_dev);
  const unsigned key_mix = attn_keymix((unsigned)(kpr - b * a.S_dense));  // RNG word of this lane's key
  const float bias2 = key_ok ? mbk * LOG2E : -INFINITY;
  const float c1 = a.scale * LOG2E;
  const unsigned bh = (unsigned)(b * a.H + h);
  if (tid < nqt * 64) {
    float dl = 0.f;
#pragma unroll
    for (int p = 0; p < DH / 64; ++p) dl += dp0[p];
    lds_w32(lds_addr(lse_s + tid), __float_as_uint(tid < nqs ? ls0 * LOG2E : INFINITY));  // +inf => P = 0 for dead rows
    lds_w32(lds_addr(dlt_s + tid), __float_as_uint(dl));
    lds_w32(lds_addr(rk_s + tid), attn_rowkey(dkey, bh, (unsigned)a.S4, (unsigned)(qpr0 - b * a.S_dense)));
  }
  for (int q = tid + NT; q < nqt * 64; q += NT) {  // (samples longer than the block is wide)
    const int qc = min(q, nqs - 1);
    const int row = a.qsel ? b * a.nq + qc : off + qc;  // row in lse / dparts
    float dl = 0.f;
#pragma unroll
    for (int p = 0; p < DH / 64; ++p) dl += a.dparts[(int64_t)row * (a.d / 64) + h * (DH / 64) + p];
    const float l2 = q < nqs ? a.lse[(int64_t)row * a.H + h] * LOG2E : INFINITY;
    const int qpos = orig_pos(a, b, off, a.qsel ? a.qsel[b * a.nq + qc] : off + qc);  // original position (RNG coordinate)
    lds_w32(lds_addr(lse_s + q), __float_as_uint(l2));
    lds_w32(lds_addr(dlt_s + q), __float_as_uint(dl));
    lds_w32(lds_addr(rk_s + q), attn_rowkey(dkey, bh, (unsigned)a.S4, (unsigned)qpos));
  }
  const unsigned ka = frag_b128_off<DH>(lane), va = frag_tr_off<DH>(lane);
  const unsigned s_base = lds_addr(smem), meta_a = lds_addr(lse_s) + 16 * lg;
  const unsigned SPB = (unsigned)SP * 4;
#pragma unroll
  for (int kk = 0; kk < KK; ++kk) { asm volatile("" : "+v"(kf[kk])); asm volatile("" : "+v"(vf[kk])); }

  f32x4 dk[FD], dv[FD];
#pragma unroll
  for (int i = 0; i < FD; ++i) { dk[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
  ATT_MARK(1); ATT_SET(5, nqt); ATT_SET(6, 1); ATT_SET(7, Sb);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  ATT_MARK(2);

  for (int qt = 0; qt < nqt; ++qt) {
    if (qt == 1) ATT_MARK(8);
    if (qt >= 2) {
      if (qt + 1 < nqt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
    const unsigned qb = s_base + (qt & 1) * STAGEB + ka, qtb = s_base + (qt & 1) * STAGEB + va;
    // row-major fragments of the Q tile (-> S) and of the dO tile (-> dA)
    u32x4 R[KK][8];
    auto bissue = [&](int kk) {
#pragma unroll
      for (int f = 0; f < 4; ++f) lds_r128(R[kk][f], (qb ^ (unsigned)(kk << 6)) + (unsigned)(f * 16 * ROWB));
#pragma unroll
      for (int f = 0; f < 4; ++f) lds_r128(R[kk][4 + f], (qb ^ (unsigned)(kk << 6)) + (unsigned)(TILEB + f * 16 * ROWB));
    };
    bissue(0);
    f32x4 s[4], da[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) { s[f] = (f32x4){0.f, 0.f, 0.f, 0.f}; da[f] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
      if (kk + 1 < KK) { bissue(kk + 1); lgkm_wait<8>(); } else { lgkm_wait<0>(); }
#pragma unroll
      for (int f = 0; f < 8; ++f) TIE(R[kk][f]);
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        s[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf8(R[kk][f]), kf[kk], s[f], 0, 0, 0);
        da[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf8(R[kk][4 + f]), vf[kk], da[f], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // lane (key = li, g) holds S[q = 16f + 4g + r][key]: lse / delta / row key of those queries, f = 0, 1 then f = 2, 3
    u32x4 ml[4], md[4], mr[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      const unsigned ma = meta_a + (unsigned)((qt * 64 + f * 16) * 4);
      lds_r128(ml[f], ma);
      lds_r128(md[f], ma + SPB);
      lds_r128(mr[f], ma + 2 * SPB);
    }
    // transposed fragments: groups 0 .. NG-1 from the dO tile (-> dV), NG .. 2 NG - 1 from the Q tile (-> dK)
    u32x2 tlo[2 * NG][4], thi[2 * NG][4];
    auto tissue = [&](int g) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int idx = (g % NG) * 4 + j, ks = idx / FD, fd = idx % FD;
        lds_rtr2<16 * ROWB>(tlo[g][j], thi[g][j], ((qtb + (unsigned)((g < NG ? TILEB : 0) + ks * 32 * ROWB)) ^ (unsigned)(fd << 5)));
      }
    };
    tissue(0);
    f32x4 pa[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      if (f == 0) lgkm_wait<14>();      // behind the first two rows' metadata: f = 2, 3 (6 reads) + group 0 (8)
      if (f == 2) lgkm_wait<8>();       // behind the rest: group 0
      TIE(ml[f]); TIE(md[f]); TIE(mr[f]);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = fast_exp2(s[f][r] * c1 + bias2 - __uint_as_float(ml[f][r]));
        bool keep = true;
        if (a.thr16) keep = attn_keep_mixed(mr[f][r], key_mix, a.thr16);
        pa[f][r] = keep ? p * a.drop_scale : 0.f;                 // A = dropout(P)
        const float dp = keep ? da[f][r] * a.drop_scale : 0.f;
        s[f][r] = p * (dp - __uint_as_float(md[f][r]));           // dS
      }
    }
    tissue(1);
    const bf16x8_t ab[2] = {pack8(pa[0], pa[1]), pack8(pa[2], pa[3])};
    const bf16x8_t sb[2] = {pack8(s[0], s[1]), pack8(s[2], s[3])};
#ifdef MMT_ATTN_LAB_DS_SPLIT
    bf16x8_t sl[2];
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
      f32x4 r0, r1;
#pragma unroll
      for (int r = 0; r < 4; ++r) { r0[r] = s[2 * h2][r] - (float)sb[h2][r]; r1[r] = s[2 * h2 + 1][r] - (float)sb[h2][4 + r]; }
      sl[h2] = pack8(r0, r1);
    }
#endif
#pragma unroll
    for (int g = 0; g < 2 * NG; ++g) {
      if (g + 1 < 2 * NG) lgkm_wait<8>(); else lgkm_wait<0>();
#pragma unroll
      for (int j = 0; j < 4; ++j) { TIE(tlo[g][j]); TIE(thi[g][j]); }
      if (g + 2 < 2 * NG) tissue(g + 2);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int idx = (g % NG) * 4 + j, ks = idx / FD, fd = idx % FD;
        if (g < NG) dv[fd] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(join_bf8(tlo[g][j], thi[g][j]), ab[ks], dv[fd], 0, 0, 0);
        else dk[fd] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(join_bf8(tlo[g][j], thi[g][j]), sb[ks], dk[fd], 0, 0, 0);
#ifdef MMT_ATTN_LAB_DS_SPLIT
        if (g >= NG) dk[fd] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(join_bf8(tlo[g][j], thi[g][j]), sl[ks], dk[fd], 0, 0, 0);
#endif
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (qt + 2 < nqt) {
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      stage(qt + 2);
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
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
// forward kernel: consecutive ids go to different XCDs, so every XCD gets the same mix of live and empty tiles).  The two
// roles are independent, so the whole backward of a layer's attention is one node of the step graph and
// its two halves share the CUs instead of running back to back.
// ------------------------------------------------------------------------------------------------
template <int DH>
__global__ __launch_bounds__(256, 2) void attn_bwd_kernel(AttnArgs a, int q_tiles, int k_tiles) {
  extern __shared__ __attribute__((aligned(256))) unsigned char att_smem[];
  ATT_MARK(0);
  if (a.work) {  // scheduled order (attn_sched.h): one scalar 16-byte load tells the block what it is
    const int32_t* w = a.work + (int64_t)blockIdx.x * 4;
    const int wb = w[0], wk = w[1], woff = w[2], wlen = w[3];
    if (wb >= 0) {
      const int h = wk & 0xff, role = (wk >> 8) & 1, tile = wk >> 16;
      if (role) attn_bwd_dq_block<DH>(a, att_smem, tile, wb, h, woff, wlen);
      else attn_bwd_dkv_block<DH>(a, att_smem, tile, wb, h, woff, wlen);
    }
  } else {
  const int nbh = a.H * a.B, bh = (int)blockIdx.x % nbh, slot = (int)blockIdx.x / nbh;
  const int b = bh / a.H, h = bh % a.H;
  const int off = a.cu ? a.cu[b] : b * a.S_dense;
  const int Sb = a.cu ? a.cu[b + 1] - off : a.S_dense;
  if (q_tiles < 0) {  // lab (MMT_ATTN_BWD_MERGE=1): tile i of BOTH roles in one block, one after the other
    attn_bwd_dkv_block<DH>(a, att_smem, slot, b, h, off, Sb);
    __syncthreads();
    if (slot < -q_tiles) attn_bwd_dq_block<DH>(a, att_smem, slot, b, h, off, Sb);
  } else {
  // All dK/dV tiles first (the longer role), then the dQ tiles.  Tiles past a sample's length exit at once and hand their
  // slot to the next block in line.  (Order without a work list: callers outside the engine, dense batches.)
  int role, tile;  // role 0 = dK/dV, 1 = dQ
  if (slot < k_tiles) { role = 0; tile = slot; }
  else { role = 1; tile = slot - k_tiles; }
  if (role) attn_bwd_dq_block<DH>(a, att_smem, tile, b, h, off, Sb);
  else attn_bwd_dkv_block<DH>(a, att_smem, tile, b, h, off, Sb);
  }
  }
#ifdef MMT_GEMM2_INSTR
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  ATT_MARK(4);
  ATT_SET(9, __builtin_amdgcn_s_getreg((31 << 11) | 20));
  ATT_SET(10, __builtin_amdgcn_s_getreg((31 << 11) | 4));
  ATT_SET(11, wall_clock64());
#endif
}

// delta partials for callers that do not get them from a GEMM epilogue: dparts[row, c] = sum over the 64 columns of group c
// of dO[row, .] * O[row, .] (fp32 accumulation of bf16 products), 16 lanes x 4 columns per (row, group).
__global__ __launch_bounds__(256) void attn_delta_kernel(const bf16_t* __restrict__ dctx, const bf16_t* __restrict__ ctx,
                                                         int64_t ld, float* __restrict__ dparts, int rows, int d,
                                                         const int32_t* __restrict__ cu, int B) {
  if (cu) rows = min(rows, cu[B]);  // packed batch: the buffers hold the live rows only, not B * S
  const int groups = d / 64;
  const int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t g = idx >> 4;
  const int l = (int)(idx & 15);
  if (g >= (int64_t)rows * groups) return;  // (a whole 16-lane group leaves together)
  const int row = (int)(g / groups), grp = (int)(g % groups);
  const u32x2 dv = *(const u32x2*)(dctx + (int64_t)row * ld + grp * 64 + l * 4);
  const u32x2 ov = *(const u32x2*)(ctx + (int64_t)row * ld + grp * 64 + l * 4);
  float part = bf2f((bf16_t)(dv[0] & 0xffff)) * bf2f((bf16_t)(ov[0] & 0xffff)) + bf2f((bf16_t)(dv[0] >> 16)) * bf2f((bf16_t)(ov[0] >> 16)) +
               bf2f((bf16_t)(dv[1] & 0xffff)) * bf2f((bf16_t)(ov[1] & 0xffff)) + bf2f((bf16_t)(dv[1] >> 16)) * bf2f((bf16_t)(ov[1] >> 16));
  part += __shfl_xor(part, 1, 64); part += __shfl_xor(part, 2, 64);
  part += __shfl_xor(part, 4, 64); part += __shfl_xor(part, 8, 64);
  if (l == 0) dparts[g] = part;
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

// dynamic LDS of the attention kernels (two stages + per-row arrays), raised above the 64 KiB default once per kernel
template <typename K> static int set_lds(K kernel, size_t bytes, size_t* configured) {
  if (bytes > *configured) {
    if (hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) return MMT_ERR_ARG;
    *configured = bytes;
  }
  return 0;
}

// nq queries per sample.  8-wave blocks (128 queries) when a sample has more than 64 queries; MMT_ATTN_FWD_WAVES=4 (lab:
// same-box A/B) keeps the 4-wave blocks of r01-r02.
static int launch_fwd(const AttnArgs& a, int nq, int H, int B, bool dh128, hipStream_t s) {
  static int waves = -1;
  static size_t conf[4] = {0, 0, 0, 0};
  if (waves < 0) {
    const char* e = getenv("MMT_ATTN_FWD_WAVES");
    waves = e ? atoi(e) : 8;
  }
  const size_t lds = attn_lds_bytes(dh128 ? 128 : 64, a.S_dense);
  if (lds > 160 * 1024) return MMT_ERR_ARG;
  if (waves == 8 && nq > 64) {
    const dim3 grid(((nq + 127) / 128) * H * B);
    if (dh128) { if (set_lds(attn_fwd_kernel<128, 8>, lds, &conf[0])) return MMT_ERR_ARG; hipLaunchKernelGGL((attn_fwd_kernel<128, 8>), grid, dim3(512), lds, s, a); }
    else { if (set_lds(attn_fwd_kernel<64, 8>, lds, &conf[1])) return MMT_ERR_ARG; hipLaunchKernelGGL((attn_fwd_kernel<64, 8>), grid, dim3(512), lds, s, a); }
  } else {
    const dim3 grid(((nq + 63) / 64) * H * B);
    if (dh128) { if (set_lds(attn_fwd_kernel<128, 4>, lds, &conf[2])) return MMT_ERR_ARG; hipLaunchKernelGGL((attn_fwd_kernel<128, 4>), grid, dim3(256), lds, s, a); }
    else { if (set_lds(attn_fwd_kernel<64, 4>, lds, &conf[3])) return MMT_ERR_ARG; hipLaunchKernelGGL((attn_fwd_kernel<64, 4>), grid, dim3(256), lds, s, a); }
  }
  return (int)hipGetLastError();
}

// tq query tiles (dQ role) + tk key tiles (dK/dV role) in one launch.  MMT_ATTN_BWD_SPLIT=1 (lab: same-box A/B) issues
// the two roles as two launches of the same kernel, the r02 structure.  delta_ready = 0: the delta partials are formed
// here (one more launch); 1: the caller's `delta` buffer already holds them (MmtEpilogue.dot_out of the dO GEMM).
static int launch_bwd(const AttnArgs& a, int tq, int tk, int H, int B, bool dh128, int delta_ready, int rows_c, hipStream_t s) {
  static int split = -1;
  static size_t conf[2] = {0, 0};
  if (split < 0) {
    const char* e = getenv("MMT_ATTN_BWD_SPLIT");
    split = e ? atoi(e) : 0;
  }
  const size_t lds = attn_lds_bytes(dh128 ? 128 : 64, a.S_dense);
  if (lds > 160 * 1024) return MMT_ERR_ARG;
  if (dh128) { if (set_lds(attn_bwd_kernel<128>, lds, &conf[0])) return MMT_ERR_ARG; }
  else { if (set_lds(attn_bwd_kernel<64>, lds, &conf[1])) return MMT_ERR_ARG; }
  if (!delta_ready) {
    const int64_t lanes = (int64_t)rows_c * (a.d / 64) * 16;
    hipLaunchKernelGGL(attn_delta_kernel, dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, s, a.dctx, (const bf16_t*)a.ctx, a.ldc,
                       (float*)a.dparts, rows_c, a.d, a.qsel ? nullptr : a.cu, B);
  }
  auto go = [&](int q_tiles, int k_tiles) {
    const int gx = (q_tiles + k_tiles) * H * B;
    if (dh128) hipLaunchKernelGGL(attn_bwd_kernel<128>, dim3(gx), dim3(256), lds, s, a, q_tiles, k_tiles);
    else hipLaunchKernelGGL(attn_bwd_kernel<64>, dim3(gx), dim3(256), lds, s, a, q_tiles, k_tiles);
  };
  if (a.work) {
    go(tq, tk);
  } else if (split == 2 && tq <= tk) {  // lab: merged roles (see the kernel)
    const int gx = tk * H * B;
    if (dh128) hipLaunchKernelGGL(attn_bwd_kernel<128>, dim3(gx), dim3(256), lds, s, a, -tq, tk);
    else hipLaunchKernelGGL(attn_bwd_kernel<64>, dim3(gx), dim3(256), lds, s, a, -tq, tk);
  } else if (split) {
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

// The block order of the backward for a packed batch (attn_sched.h): work = mmt_attn_schedule_words(B, S, H) int32 words.
__global__ __launch_bounds__(512) void attn_schedule_kernel(AttnSched s) { attn_schedule_block(s); }
extern "C" int64_t mmt_attn_schedule_words(int B, int S, int H) { return (int64_t)2 * ((S + 63) / 64) * B * H * 4; }
static bool attn_schedulable(int B, int H) { return (B * H) % 8 == 0 && B * H <= 8 * 64 * ATT_SCHED_CHUNKS && H <= 255; }
extern "C" int mmt_attn_schedule(const int32_t* cu_seqlens, int B, int S, int H, int32_t* work, void* stream) {
  if (!cu_seqlens || !work || B <= 0 || S <= 0 || H <= 0 || !attn_schedulable(B, H)) return MMT_ERR_ARG;
  AttnSched sc = {cu_seqlens, work, B, H, (S + 63) / 64};
  hipLaunchKernelGGL(attn_schedule_kernel, dim3(1), dim3(512), 0, (hipStream_t)stream, sc);
  return (int)hipGetLastError();
}

// work (nullable): the schedule of THIS batch (mmt_attn_schedule, or the engine's rider in the embedding LayerNorm launch);
// same results, blocks in longest-first order.
extern "C" int mmt_attn_bwd_ex(const void* qkv, const int32_t* cu_seqlens, const float* mask_bias, const void* ctx,
                               const float* lse, const void* dctx, void* dqkv, float* delta, int delta_ready, int B, int S,
                               int H, int d, float scale, uint32_t drop_key, uint32_t thr16, float drop_scale,
                               const uint32_t* seed_dev, const int32_t* row_index, const int32_t* work, void* stream) {
  if (int e = check_args(qkv, B, S, H, d)) return e;
  if (!mask_bias || !ctx || !lse || !dctx || !dqkv || !delta) return MMT_ERR_ARG;
  AttnArgs a = {};
  a.qkv = (const bf16_t*)qkv; a.ld = 3 * (int64_t)d; a.cu = cu_seqlens; a.S_dense = S; a.mask_bias = mask_bias;
  a.ctx = (bf16_t*)ctx; a.ldc = d; a.lse = (float*)lse; a.dctx = (const bf16_t*)dctx; a.dqkv = (bf16_t*)dqkv;
  a.dparts = delta; a.H = H; a.d = d; a.B = B; a.scale = scale;
  a.drop_key = drop_key; a.thr16 = thr16; a.drop_scale = drop_scale; a.S4 = (S + 3) & ~3; a.seed_dev = seed_dev;
  a.row_index = row_index;
  if (work && (!cu_seqlens || !attn_schedulable(B, H))) return MMT_ERR_ARG;
  a.work = work;
  const int tiles = (S + 63) / 64;
  return launch_bwd(a, tiles, tiles, H, B, d == H * 128, delta_ready, B * S, (hipStream_t)stream);
}
extern "C" int mmt_attn_bwd(const void* qkv, const int32_t* cu_seqlens, const float* mask_bias, const void* ctx,
                            const float* lse, const void* dctx, void* dqkv, float* delta, int B, int S, int H,
                            int d, float scale, uint32_t drop_key, uint32_t thr16, float drop_scale,
                            const uint32_t* seed_dev, const int32_t* row_index, void* stream) {
  return mmt_attn_bwd_ex(qkv, cu_seqlens, mask_bias, ctx, lse, dctx, dqkv, delta, 0, B, S, H, d, scale, drop_key, thr16, drop_scale,
                         seed_dev, row_index, nullptr, stream);
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

extern "C" int mmt_attn_bwd_rows_ex(const void* qkv, const int32_t* cu_seqlens, const float* mask_bias, const int32_t* qsel,
                                    int nq, const void* ctx, const float* lse, const void* dctx, void* dqkv, float* delta,
                                    int delta_ready, int B, int S, int H, int d, float scale, uint32_t drop_key, uint32_t thr16,
                                    float drop_scale, const uint32_t* seed_dev, const int32_t* row_index, void* stream) {
  if (int e = check_args(qkv, B, S, H, d)) return e;
  if (!mask_bias || !ctx || !lse || !dctx || !dqkv || !delta || !qsel || nq <= 0) return MMT_ERR_ARG;
  AttnArgs a = {};
  a.qkv = (const bf16_t*)qkv; a.ld = 3 * (int64_t)d; a.cu = cu_seqlens; a.S_dense = S; a.mask_bias = mask_bias;
  a.ctx = (bf16_t*)ctx; a.ldc = d; a.lse = (float*)lse; a.dctx = (const bf16_t*)dctx; a.dqkv = (bf16_t*)dqkv;
  a.dparts = delta; a.H = H; a.d = d; a.B = B; a.scale = scale;
  a.drop_key = drop_key; a.thr16 = thr16; a.drop_scale = drop_scale; a.S4 = (S + 3) & ~3; a.seed_dev = seed_dev;
  a.row_index = row_index;
  a.qsel = qsel; a.nq = nq;
  return launch_bwd(a, (nq + 63) / 64, (S + 63) / 64, H, B, d == H * 128, delta_ready, B * nq, (hipStream_t)stream);
}
extern "C" int mmt_attn_bwd_rows(const void* qkv, const int32_t* cu_seqlens, const float* mask_bias, const int32_t* qsel,
                                 int nq, const void* ctx, const float* lse, const void* dctx, void* dqkv, float* delta,
                                 int B, int S, int H, int d, float scale, uint32_t drop_key, uint32_t thr16,
                                 float drop_scale, const uint32_t* seed_dev, const int32_t* row_index, void* stream) {
  return mmt_attn_bwd_rows_ex(qkv, cu_seqlens, mask_bias, qsel, nq, ctx, lse, dctx, dqkv, delta, 0, B, S, H, d, scale, drop_key,
                              thr16, drop_scale, seed_dev, row_index, stream);
}

extern "C" int mmt_attn_dropout_mask(uint8_t* out, int B, int H, int S, uint32_t drop_key, uint32_t thr16,
                                     const uint32_t* seed_dev, void* stream) {
  if (!out) return MMT_ERR_ARG;
  hipLaunchKernelGGL(attn_mask_export_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream, out, B, H, S,
                     (S + 3) & ~3, drop_key, thr16, seed_dev);
  return (int)hipGetLastError();
}
