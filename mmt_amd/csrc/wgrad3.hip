// Grouped weight gradients on 256 x 256 tiles (gfx950, r04):  out[N, K2] = sum_rows A[rows, N]^T . B[rows, K2]  (+ bias gradient)
//
// The 128 x 128 tiles of wgrad_phased_kernel (gemm.hip) pull 32 KiB out of L2 per 64 contracted rows for 2 MFLOP: at the
// ~20-23 B/clk a CU ingests that is 3x the MFMA time, and the kernel sits at 0.33-0.35 of the MFMA peak on configs[4]'s
// d = 1024 layers whatever its schedule (PMC: 2.8x the algorithmic bytes, re-read by the 4 rounds of
// tiles).  A 256 x 256 tile moves half the bytes per MAC and configs[4]'s four weight gradients are exactly 256 such tiles
// -- one per CU, every operand byte fetched once per tile row / column.
//
// Schedule: gemm3.hip's eight-phase template (two wave groups a barrier apart, a 64-row unit of the contraction in four
// phases, four 16 KiB half-tiles per LDS buffer requested one per phase and never drained), re-derived for K-MAJOR operands:
//   * both operands arrive as [64 contracted rows][128 columns] half-tiles (256-byte row segments) in the swizzled layout
//     of wgrad_phased_kernel, and the MFMA operands (8 consecutive k per lane) come out of them through
//     ds_read_b64_tr_b16 -- two transpose reads where the NT kernel has one ds_read_b128;
//   * v_mfma_f32_16x16x32_bf16: a wave owns 128 x 64 outputs = 8 x 4 fragments (128 accumulator registers), a phase is one
//     64 x 32 quadrant = 16 MFMAs (512 cycles, as the NT kernel's 8 of 32x32x16);
//   * LDS-DMA by buffer_load ... lds with the descriptor re-based per unit and num_records cut at the live row count: the
//     rows of a ragged last unit read as zeros (no zero-fill pass, no clamping).
// The bias gradient (column sums of A) is formed on the VALU from the A fragments the blocks of the first tile column hold
// anyway: two MFMAs against an all-ones operand per phase, fragment wn by wave wn of a wave row (all four hold the same
// fragments; v_dot2c on the VALU was tried: ~20 cycles apiece beside the MFMAs, 170 per phase against ~35 for the two MFMAs).
// The first version summed all eight fragments in ONE wave by shift / and / add (~100 VALU instructions in each of phases 1
// and 3): those waves reached their barriers ~450 cycles late, and since every block is one per CU the launch waited for
// them -- ~100 of 508 us (parts-off lab: tools/wgrad3_lab.py).
// Measured (configs[4], 14.4 k live rows): 566 -> 479 us per launch inside the step, 0.34 -> 0.40 of the MFMA peak.  Parts-off
// lab (tools/wgrad3_lab.py): the transpose reads cost nothing against plain 8-byte reads, and with the fragment reads OR the
// MFMAs compiled out the launch still takes ~425 us -- a skeleton of requests, waits and barriers (the NT kernel with the same
// schedule: 0.53); DESIGN section 7.
#include "mmt_common.h"
#include "../../include/mmt_hip.h"
#include <type_traits>

#define W3_BUF 65536   // bytes per LDS buffer: [A-top | A-bottom | B-left | B-right], 16 KiB each
#define W3_AT 0
#define W3_AB 16384
#define W3_BL 32768
#define W3_BR 49152

#define W3_TIE(x) asm volatile("" : "+v"(x))
// one fragment = 16 columns x 32 contracted rows: two transpose reads 16 rows (4096 bytes) apart
#if defined(MMT_W3_LAB_NOREADS)  // lab (wrong results): no fragment reads at all
#define W3_RD(lo, hi, base, OFF) do { asm volatile("" : "=v"(lo) : "v"(base)); asm volatile("" : "=v"(hi) : "v"(base)); } while (0)
#elif defined(MMT_W3_LAB_PLAINREADS)  // lab (wrong results): the same bytes by plain 8-byte reads
#define W3_RD(lo, hi, base, OFF) do {                                                          \
    asm volatile("ds_read_b64 %0, %1 offset:" #OFF : "=v"(lo) : "v"(base));                   \
    asm volatile("ds_read_b64 %0, %1 offset:" #OFF "+4096" : "=v"(hi) : "v"(base));           \
  } while (0)
#else
#define W3_RD(lo, hi, base, OFF) do {                                                          \
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:" #OFF : "=v"(lo) : "v"(base));            \
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:" #OFF "+4096" : "=v"(hi) : "v"(base));    \
  } while (0)
#endif

template <int N> __device__ __forceinline__ void w3_vmwait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void w3_vmwait_rt(int n) {  // (wave-uniform n from {0, 2, 4, 6, 8})
  if (n >= 8) w3_vmwait<8>();
  else if (n == 6) w3_vmwait<6>();
  else if (n == 4) w3_vmwait<4>();
  else if (n == 2) w3_vmwait<2>();
  else w3_vmwait<0>();
}
__device__ __forceinline__ void w3_barrier() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
__device__ __forceinline__ bf16x8_t w3_join(const u32x2& lo, const u32x2& hi) {
  const u32x4 v = {lo[0], lo[1], hi[0], hi[1]};
  return __builtin_bit_cast(bf16x8_t, v);
}
// acc + the sum of the 8 bf16 values of a fragment register set: four v_dot2c_f32_bf16 against (1, 1)
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ float w3_sum8(const u32x2& lo, const u32x2& hi, float acc) {
  const bf16x2_t ones = {(__bf16)1.0f, (__bf16)1.0f};
  // (elements into scalars first: __builtin_bit_cast of a vector ELEMENT yields element 0 with this compiler)
  const unsigned a0 = lo[0], a1 = lo[1], a2 = hi[0], a3 = hi[1];
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a0), ones, acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a1), ones, acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a2), ones, acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a3), ones, acc, false);
  return acc;
}
// the bias sums of fragment I (both k-sub-steps) -- wave wn of a wave row takes fragment wn: the four waves hold the same A
// fragments, so the work is spread over them and no wave's phase grows by more than eight instructions
#ifdef MMT_W3_BIAS_DOT2  // (the VALU form: ~20 cycles per v_dot2c beside the MFMAs, 170 per phase -- tools/wgrad3_budget.py)
#define W3_BSUM(DST, I) do { DST[0] = w3_sum8(fal[I][0], fah[I][0], DST[0]); DST[0] = w3_sum8(fal[I][1], fah[I][1], DST[0]); } while (0)
#else  // two more MFMAs against an all-ones operand: every lane of a column ends up with the column's sum over the unit
#define W3_BSUM(DST, I) do {                                                                                                  \
    DST = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, w3_join(fal[I][0], fah[I][0]), DST, 0, 0, 0);                           \
    DST = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, w3_join(fal[I][1], fah[I][1]), DST, 0, 0, 0);                           \
  } while (0)
#endif
#define W3_BSUM_MINE(DST) do {                                        \
    if (wn == 0) W3_BSUM(DST, 0); else if (wn == 1) W3_BSUM(DST, 1);  \
    else if (wn == 2) W3_BSUM(DST, 2); else W3_BSUM(DST, 3);          \
  } while (0)

#ifdef MMT_GEMM2_INSTR
__device__ long long* g_wgrad3_dbg = nullptr;  // [blocks][2 groups][20]
extern "C" int mmt_debug_set_wgrad3_buffer(long long* p) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_wgrad3_dbg), &p, sizeof(p));
}
#endif

__global__ __launch_bounds__(512) void wgrad3_kernel(MmtWgradGroup g) {
  extern __shared__ __attribute__((aligned(256))) unsigned char smem_raw[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // (scalar: everything derived from it stays out of the VGPRs)
  const int wm = wave >> 2, wn = wave & 3;  // waves 0-3 / 4-7 share the SIMDs pairwise: the two groups of the schedule
  const int li = lane & 15, lg = lane >> 4;
  const int id = xcd_remap((int)blockIdx.x, (int)gridDim.x);
  int p = 0;
#pragma unroll 1
  for (int q = 1; q < g.count; ++q)
    if (id >= g.item[q].tile_begin) p = q;
  const MmtWgradItem& it = g.item[p];
  const int tiles_k = it.K2 / 256, tiles_n = it.N / 256;
  const int tile = id - it.tile_begin;
  // consecutive tile ids (= one XCD, xcd_remap) share a panel of the LARGER operand: its bytes enter that XCD's L2 once.
  // (tn-major for all items, PMC: 1.36 GB fetched per launch at configs[4] for 0.53 GB of operands -- dW2's 177 MB B operand
  // went to four XCDs.)
  const bool k_major = it.K2 > it.N;
  const int tn = k_major ? tile % tiles_n : tile / tiles_k, tk = k_major ? tile / tiles_n : tile % tiles_k;
  const int n0 = tn * 256, k0 = tk * 256;
  const int64_t lda = it.lda, ldb = it.ldb;
  const int nrows = it.reserved > 0 ? it.reserved
                    : it.n_rows_dev ? min(*it.n_rows_dev, g.rows)
                                    : (g.n_rows_dev ? min(*g.n_rows_dev, g.rows) : g.rows);
  const int KT = (nrows + 63) >> 6;  // units of 64 contracted rows
  const bool want_bias = it.bias_out != nullptr && tk == 0;

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  f32x4 bs_top = {0.f, 0.f, 0.f, 0.f}, bs_bot = {0.f, 0.f, 0.f, 0.f};  // column sums of A: fragment wn of the top / bottom half
  bf16x8_t ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;

  // ---- LDS-DMA sources: this wave moves rows 8 wave + 4 i + (lane >> 4), i = 0, 1, of every half-tile; 16 lanes x 16 B per
  // row segment; the LDS image is lane-linear and the 16-byte chunk c of row r lies at chunk c ^ ((r & 7) << 1), i.e. the
  // swizzle goes onto the SOURCE column.  Half-tile column q (0..127) of
  //   A-top:  A column n0 + 128 (q >> 6) + (q & 63)        A-bottom: the same + 64
  //   B-left: B column k0 +  64 (q >> 5) + (q & 31)        B-right:  the same + 32
  unsigned oat[2], obl[2];  // per-lane byte offsets from (row 0 of the unit, column n0 / k0); A-bottom / B-right: + 128 / + 64
  unsigned lrow[2];         // LDS byte offset of the instruction's first row inside a half-tile (wave-uniform)
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = wave * 8 + i * 4 + (lane >> 4);
    const int c = (lane & 15) ^ ((r & 7) << 1);  // source chunk (8 columns) of this lane
    const int q = c * 8;
    oat[i] = (unsigned)((int64_t)r * lda * 2 + (128 * (q >> 6) + (q & 63)) * 2);
    obl[i] = (unsigned)((int64_t)r * ldb * 2 + (64 * (q >> 5) + (q & 31)) * 2);
    lrow[i] = (unsigned)(wave * 8 + i * 4) * 256;
  }
  const bf16_t* Abase = (const bf16_t*)it.A + n0;
  const bf16_t* Bbase = (const bf16_t*)it.B + k0;
  auto dma = [&](const bf16_t* base, int64_t ld, const unsigned (&o)[2], auto shift_c, unsigned half, int u) {  // half-tile of unit u -> buffer u & 1
    constexpr int SHIFT = decltype(shift_c)::value;  // byte shift of the half-tile's columns: the SCALAR offset (an immediate
    // instruction offset would move the LDS destination as well)
    const int rows_left = nrows - u * 64;  // > 0; rows past it read as zeros (beyond num_records)
    const int bytes = min(rows_left, 64) * ((int)ld * 2);  // (<= 64 rows of <= 2^20 elements: 32-bit scalar arithmetic)
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(base + (int64_t)u * 64 * ld), 0, bytes, 0x00020000);
    unsigned char* dst = smem_raw + (u & 1) * W3_BUF + half;
#pragma unroll
    for (int i = 0; i < 2; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(dst + lrow[i]), 16, (int)o[i], SHIFT, 0, 0);
  };
#define W3_DMA_AT(u) dma(Abase, lda, oat, std::integral_constant<int, 0>{}, W3_AT, u)
#define W3_DMA_AB(u) dma(Abase, lda, oat, std::integral_constant<int, 128>{}, W3_AB, u)
#define W3_DMA_BL(u) dma(Bbase, ldb, obl, std::integral_constant<int, 0>{}, W3_BL, u)
#define W3_DMA_BR(u) dma(Bbase, ldb, obl, std::integral_constant<int, 64>{}, W3_BR, u)

  // ---- fragment addresses inside a half-tile: lane (t = lane & 15, g = lane >> 4) addresses columns colbase + 4 (t & 3) ..
  // + 3 of row 4 g + (t >> 2) (k-sub-step 0, first half); + 16 rows = + 4096 B, k-sub-step 1 = + 8192 B: immediates.
  // colbase (a multiple of 16) only XORs the chunk index.
  unsigned t_off;
  {
    const int t = lane & 15, gq = lane >> 4;
    const int col = 4 * (t & 3), r0 = 4 * gq + (t >> 2);
    t_off = (unsigned)((r0 * 128 + (((col >> 3) ^ ((r0 & 7) << 1)) << 3) + (col & 7)) * 2);
  }
  const unsigned lds0 = (unsigned)(uintptr_t)LDS_PTR(smem_raw);
  const unsigned a_off = t_off ^ (unsigned)(wm * 64 * 2);   // + (i << 5) per fragment
  const unsigned b_off = t_off ^ (unsigned)(wn * 32 * 2);   // + (j << 5) per fragment

  // ---- prologue: unit 0 complete, A-top and B-left of unit 1 ----
  if (KT > 0) {  // (nothing to contract: the gradient is zero)
  W3_DMA_AT(0); W3_DMA_BL(0); W3_DMA_BR(0); W3_DMA_AB(0);
  if (KT > 1) { W3_DMA_AT(1); W3_DMA_BL(1); w3_vmwait<8>(); }
  else w3_vmwait<4>();
  w3_barrier();              // A-top(0), B-left(0) of every wave have landed
  if (wm == 1) w3_barrier(); // group 1 runs one barrier behind group 0 from here on

  // fragment registers: A fragments of the current row half (4 fragments x 2 k-sub-steps), right B fragments, left B fragments
  // of the current / next unit
  u32x2 fal[4][2], fah[4][2], frl[2][2], frh[2][2], fll[2][2][2], flh[2][2][2];
#ifdef MMT_W3_LAB_NOMFMA  // lab (wrong results): the operands are consumed by one cheap op, the matrix pipe idles
#define W3_MFMA(I, J, BL_, BH_, AL_, AH_) acc[I][J][0] += __uint_as_float(BL_[0] ^ AH_[1])
#else
#define W3_MFMA(I, J, BL_, BH_, AL_, AH_) acc[I][J] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w3_join(BL_, BH_), w3_join(AL_, AH_), acc[I][J], 0, 0, 0)
#endif
  // lab build (python -m mmt_amd.build --instr): per-phase s_memtime ticks, as gemm3.hip -- [phase][0] reads + requests + vmcnt,
  // [1] barrier, [2] lgkm wait + 16 MFMAs, [3] barrier; read back by tools/wgrad3_budget.py through mmt_debug_set_wgrad3_buffer
#ifdef MMT_GEMM2_INSTR
  long long w3t[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, w3p = clock64();
  const long long w3t0 = w3p;
  int w3ph = 0;
#define W3_TICK(k) do { const long long n_ = clock64(); w3t[w3ph * 4 + (k)] += n_ - w3p; w3p = n_; } while (0)
#define W3_PHASE(p) w3ph = (p)
#else
#define W3_TICK(k) do {} while (0)
#define W3_PHASE(p) do {} while (0)
#endif
#define W3_SEG_BEGIN() do { W3_TICK(0); w3_barrier(); W3_TICK(1); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_setprio(1); } while (0)
#define W3_SEG_END() do { __builtin_amdgcn_s_setprio(0); __builtin_amdgcn_sched_barrier(0); W3_TICK(2); w3_barrier(); W3_TICK(3); } while (0)
  {  // left B fragments of unit 0 (landed: the prologue's wait)
    const unsigned xb = lds0 + W3_BL + b_off;
    W3_RD(fll[0][0][0], flh[0][0][0], xb, 0); W3_RD(fll[0][0][1], flh[0][0][1], xb, 8192);
    W3_RD(fll[0][1][0], flh[0][1][0], xb ^ 32u, 0); W3_RD(fll[0][1][1], flh[0][1][1], xb ^ 32u, 8192);
  }
  auto unit = [&](int t, auto cur_c) {
    constexpr int CUR = decltype(cur_c)::value, NXT = CUR ^ 1;
    const unsigned bo = lds0 + (unsigned)(t & 1) * W3_BUF, bn = lds0 + (unsigned)((t & 1) ^ 1) * W3_BUF;
    const bool last = t + 1 >= KT, last2 = t + 2 >= KT;
    // ---- phase 1: quadrant (top, left) ----
    W3_PHASE(0);
    {
      const unsigned xa = bo + W3_AT + a_off;
      W3_RD(fal[0][0], fah[0][0], xa, 0);        W3_RD(fal[0][1], fah[0][1], xa, 8192);
      W3_RD(fal[1][0], fah[1][0], xa ^ 32u, 0);  W3_RD(fal[1][1], fah[1][1], xa ^ 32u, 8192);
      W3_RD(fal[2][0], fah[2][0], xa ^ 64u, 0);  W3_RD(fal[2][1], fah[2][1], xa ^ 64u, 8192);
      W3_RD(fal[3][0], fah[3][0], xa ^ 96u, 0);  W3_RD(fal[3][1], fah[3][1], xa ^ 96u, 8192);
    }
    if (!last) W3_DMA_BR(t + 1);
    w3_vmwait_rt(last ? 2 : 8);  // B-right of THIS unit (requested five phases ago) has landed: read in phase 2
    W3_SEG_BEGIN();
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) { W3_TIE(fal[i][ks]); W3_TIE(fah[i][ks]); }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) { W3_TIE(fll[CUR][j][ks]); W3_TIE(flh[CUR][j][ks]); }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) W3_MFMA(i, j, fll[CUR][j][ks], flh[CUR][j][ks], fal[i][ks], fah[i][ks]);
    if (want_bias) W3_BSUM_MINE(bs_top);
    W3_SEG_END();
    // ---- phase 2: quadrant (top, right) ----
    W3_PHASE(1);
    {
      const unsigned xb = bo + W3_BR + b_off;
      W3_RD(frl[0][0], frh[0][0], xb, 0);        W3_RD(frl[0][1], frh[0][1], xb, 8192);
      W3_RD(frl[1][0], frh[1][0], xb ^ 32u, 0);  W3_RD(frl[1][1], frh[1][1], xb ^ 32u, 8192);
    }
    if (!last) W3_DMA_AB(t + 1);
    w3_vmwait_rt(last ? 0 : 8);  // A-bottom of this unit has landed: read in phase 3
    W3_SEG_BEGIN();
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) { W3_TIE(frl[j][ks]); W3_TIE(frh[j][ks]); }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) W3_MFMA(i, 2 + j, frl[j][ks], frh[j][ks], fal[i][ks], fah[i][ks]);
    W3_SEG_END();
    // ---- phase 3: quadrant (bottom, right) ----
    W3_PHASE(2);
    {
      const unsigned xa = bo + W3_AB + a_off;
      W3_RD(fal[0][0], fah[0][0], xa, 0);        W3_RD(fal[0][1], fah[0][1], xa, 8192);
      W3_RD(fal[1][0], fah[1][0], xa ^ 32u, 0);  W3_RD(fal[1][1], fah[1][1], xa ^ 32u, 8192);
      W3_RD(fal[2][0], fah[2][0], xa ^ 64u, 0);  W3_RD(fal[2][1], fah[2][1], xa ^ 64u, 8192);
      W3_RD(fal[3][0], fah[3][0], xa ^ 96u, 0);  W3_RD(fal[3][1], fah[3][1], xa ^ 96u, 8192);
    }
    if (!last2) W3_DMA_AT(t + 2);
    w3_vmwait_rt(last ? 0 : (last2 ? 4 : 6));  // A-top and B-left of the NEXT unit have landed: read in phase 4 / its phase 1
    W3_SEG_BEGIN();
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) { W3_TIE(fal[i][ks]); W3_TIE(fah[i][ks]); }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) W3_MFMA(4 + i, 2 + j, frl[j][ks], frh[j][ks], fal[i][ks], fah[i][ks]);
    if (want_bias) W3_BSUM_MINE(bs_bot);
    W3_SEG_END();
    // ---- phase 4: quadrant (bottom, left): its fragments are in registers; the NEXT unit's left B fragments are read ----
    W3_PHASE(3);
    if (!last) {
      const unsigned yb = bn + W3_BL + b_off;
      W3_RD(fll[NXT][0][0], flh[NXT][0][0], yb, 0);        W3_RD(fll[NXT][0][1], flh[NXT][0][1], yb, 8192);
      W3_RD(fll[NXT][1][0], flh[NXT][1][0], yb ^ 32u, 0);  W3_RD(fll[NXT][1][1], flh[NXT][1][1], yb ^ 32u, 8192);
    }
    if (!last2) W3_DMA_BL(t + 2);
    W3_SEG_BEGIN();
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) W3_MFMA(4 + i, j, fll[CUR][j][ks], flh[CUR][j][ks], fal[i][ks], fah[i][ks]);
    W3_SEG_END();
  };
  for (int t = 0; t < KT; t += 2) {
    unit(t, std::integral_constant<int, 0>{});
    if (t + 1 < KT) unit(t + 1, std::integral_constant<int, 1>{});
  }
  if (wm == 0) w3_barrier();  // group 0 catches up with the extra barrier group 1 took
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#ifdef MMT_GEMM2_INSTR
  if (g_wgrad3_dbg && (tid == 0 || tid == 256)) {  // wave 0 of either group
    long long* d = g_wgrad3_dbg + ((int64_t)blockIdx.x * 2 + wm) * 20;
    for (int k = 0; k < 16; ++k) d[k] = w3t[k];
    d[16] = clock64() - w3t0; d[17] = KT; d[18] = want_bias ? 1 : 0;
  }
#endif
  }
  // ---- store: acc[i][j][e] = out[n0 + 128 wm + 64 (i >> 2) + 16 (i & 3) + li][k0 + 64 wn + 32 (j >> 1) + 16 (j & 1) + 4 lg + e] ----
  float* __restrict__ out = it.out;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int n = n0 + wm * 128 + (i >> 2) * 64 + (i & 3) * 16 + li;
    if (n < it.N_out) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k2 = k0 + wn * 64 + (j >> 1) * 32 + (j & 1) * 16 + lg * 4;
        if (k2 + 3 < it.K2_out && !(it.ldo & 3)) {
          *(f32x4*)(out + (int64_t)n * it.ldo + k2) = acc[i][j];
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (k2 + e < it.K2_out) out[(int64_t)n * it.ldo + k2 + e] = acc[i][j][e];
        }
      }
    }
  }
  if (want_bias) {  // the four lane groups hold the sums of different contracted rows; this wave owns fragments wn and 4 + wn
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      float b = h ? bs_bot[0] : bs_top[0];
#ifdef MMT_W3_BIAS_DOT2
      b += __shfl_xor(b, 16, 64);
      b += __shfl_xor(b, 32, 64);
#endif
      const int n = n0 + wm * 128 + h * 64 + wn * 16 + li;
      if (lg == 0 && n < it.N_out) it.bias_out[n] = b;
    }
  }
}

// tiles of 256 x 256; no split slabs.  Called by mmt_wgrad_grouped (gemm.hip) when every item qualifies.
int mmt_wgrad3_launch(const MmtWgradGroup& h, int tiles, hipStream_t s) {
  constexpr int lds = 2 * W3_BUF;
  static bool configured = false;
  if (!configured) {
    if (hipFuncSetAttribute((const void*)wgrad3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return MMT_ERR_ARG;
    configured = true;
  }
  hipLaunchKernelGGL(wgrad3_kernel, dim3(tiles), dim3(512), lds, s, h);
  return (int)hipGetLastError();
}
