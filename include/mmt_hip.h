/* mmt_hip.h -- C ABI of libmmt_hip.so: the MI355X (gfx950) hot path of MMT.
 *
 * The reference (gabeur/mmt) is pure Python on ATen; it has no FFI layer.  The boundary a
 * maintainer binds is therefore the set of ATen op chains on the hot path;
 * each entry point below names the reference code it replaces (paths relative to the reference
 * root).  Python binds these through ctypes (mmt_amd/_lib.py); INTEGRATION.md shows the stub.
 *
 * Conventions (last row):
 *   - every pointer is a DEVICE pointer owned by the caller (torch-allocated); no ownership moves;
 *   - calls are asynchronous on `stream` (a hipStream_t passed as void*); no hipMalloc, no device
 *     sync, no global mutable state => safe under hipGraph capture;
 *   - return 0 on success, negative MMT_ERR_* on bad arguments, positive hipError_t on launch error;
 *   - bf16 tensors are raw uint16 bit patterns; "ld*" are leading dimensions in ELEMENTS;
 *   - activations are token-major [rows, channels]; row buffers are allocated with the row count
 *     rounded up to MMT_ROW_ALIGN so that GEMM tiles never need row bounds checks;
 *   - `n_rows_dev` (nullable) points at the live row count on the device (variable-length packing
 *     of valid tokens): tiles at or beyond it exit early, reductions over rows stop there;
 *   - dropout is a counter-based RNG keyed by (key, original element index): the backward pass
 *     regenerates masks from the same (key, threshold) instead of storing them.
 */
#ifndef MMT_HIP_H_
#define MMT_HIP_H_
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MMT_ABI_VERSION 3  /* r06: MmtEpilogue.rider*, MmtBertBatch.rider* / live_rows_hint, MmtAdamQueue (r04: MmtEpilogue.dot_*, mmt_attn_bwd*_ex) */
#define MMT_ROW_ALIGN 256

#define MMT_ERR_ARG (-1)    /* unsupported shape / null pointer */
#define MMT_ERR_ALIGN (-2)  /* pointer or leading dimension not 16-byte aligned */

int mmt_abi_version(void);
const char* mmt_build_info(void);

/* ---- GEMM epilogues ------------------------------------------------------------------------ */
enum {
  MMT_EPI_BF16 = 0,          /* out(bf16) = acc                                                 */
  MMT_EPI_BIAS_BF16 = 1,     /* out(bf16) = acc + bias[n]              bert.py:137-139 Q/K/V     */
  MMT_EPI_BIAS_GELU = 2,     /* out(bf16) = acc + bias; out2(bf16) = gelu_erf(out)  bert.py:217-220 */
  MMT_EPI_BIAS_DROP_RES = 3, /* out(f32) = dropout(acc + bias) + res   bert.py:186-188,234-236 (pre-LN sum) */
  MMT_EPI_DGELU = 4,         /* out(bf16) = acc * gelu_erf'(aux)       backward of bert.py:37-53 */
  MMT_EPI_ADD_F32 = 5,       /* out(f32) = acc + res                   dgrad + residual gradient */
  MMT_EPI_F32 = 6,           /* out(f32) = acc                                                   */
  MMT_EPI_BIAS_F32 = 7       /* out(f32) = acc + bias[n]               model.py:724 ReduceDim.fc */
};

typedef struct MmtEpilogue {
  const float* bias;        /* [N] fp32                                                         */
  const float* res;         /* [M, ldres] fp32 residual / addend                                */
  int64_t ldres;
  void* out2;               /* second bf16 output [M, ldout2] (MMT_EPI_BIAS_GELU)               */
  int64_t ldout2;
  const void* aux;          /* bf16 [M, ldaux] pre-activation (MMT_EPI_DGELU)                   */
  int64_t ldaux;
  float* colsum;            /* nullable: [ceil(M/128), N] per-row-tile column sums of `out`     */
  const int32_t* row_index; /* nullable: row -> original token index (b*S+s) for the RNG        */
  const uint32_t* seed_dev; /* nullable: per-step seed in device memory, hashed into drop_key   */
  uint32_t drop_key;        /* dropout stream key (seed, site, layer mixed by the host)         */
  uint32_t drop_thr16;      /* keep iff u16 >= thr16; 0 disables dropout                        */
  float drop_scale;         /* 1 / (1 - thr16/65536)                                            */
  int32_t reserved;         /* 0 = auto tile; 1 = force 128x128; 2 = force 128x64 (tests/tuning)       */
  /* MMT_EPI_BF16 only, nullable: dot_out[row, n / 64] = sum over the 64 output columns of group n / 64 of
   * out(bf16)[row, n] * dot_src(bf16)[row, n] -- with out = dO = dA . Wo and dot_src = O (the attention context) these are
   * the "delta" sums of the attention backward (rowsum(dO * O) per head = DH / 64 groups), formed while dO is still in
   * registers instead of by re-reading dO and O (model/bert.py:141-168 backward).  dot_out: fp32 [M, N / 64]. */
  const void* dot_src;
  int64_t lddot;
  float* dot_out;
  /* r06, nullable: an optimizer work queue (MmtAdamQueue in DEVICE memory, see "Adam riders" below).  Blocks of this
   * launch that have no tile to compute -- tiles past the live row count, plus the extra blocks the launcher appends when
   * a rider is attached -- take entries of the queue's first rider_limit STAGES (parameters whose gradients are final before
   * this launch) and run the Adam update on them for as long as the GEMM's own blocks are still running; rider_slot names
   * the "finished blocks" counter of this launch inside the queue's state.  Tiles that do not host riders ignore the fields. */
  const void* rider;
  int32_t rider_limit;
  int32_t rider_slot;
  /* r06: with n_rows_dev set, the live row count as the HOST knows it (0 = unknown: the tile choice then prices the problem
   * at all M rows).  Only mmt_gemm_select_tile reads it; the kernels read n_rows_dev. */
  int32_t live_rows_hint;
  int32_t rider_cap;        /* rider blocks at work in this launch, chip-wide (0 = the library's default, 64) */
} MmtEpilogue;

/* C[M,N] = A[M,K] . B[N,K]^T  (both operands K-contiguous bf16, fp32 accumulate on MFMA).
 * Replaces every nn.Linear forward on the path (bert.py:137-139,186,218,234; model.py:724) and,
 * with B = W^T copies, the input-gradient GEMMs of their backward.
 * Requirements: K % 64 == 0, N % 64 == 0, rows allocated to a multiple of 128. */
int mmt_gemm_nt_bf16(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                     int M, int N, int K, int epilogue, const MmtEpilogue* epi,
                     const int32_t* n_rows_dev, void* stream);

/* The tile mmt_gemm_nt_bf16 runs a problem on (13 / 14 / 18: gemm2.hip, 21: gemm3.hip, 24 / 25: gemm5.hip, 1 / 2: the 4-wave
 * 128x128 / 128x64 kernel of gemm.hip): a pure function of the shape, the epilogue and the live row count the host knows
 * (packed != 0: n_rows_dev given; live_rows_hint as MmtEpilogue.live_rows_hint; has_colsum / has_dot_out: the epilogue's
 * optional outputs; reserved as MmtEpilogue.reserved).  Host-only, no GPU needed (tests/test_host_cpu.py pins the policy). */
int mmt_gemm_select_tile(int epilogue, int M, int N, int K, int packed, int live_rows_hint, int has_colsum, int has_dot_out,
                         int reserved);

/* Split-K form for skinny problems (M up to a few hundred rows, long K): K-slices run as independent blocks writing fp32
 * partial slabs into `ws`, a second kernel sums them in a fixed order and applies the epilogue
 * (MMT_EPI_BF16 / F32 / BIAS_F32 / ADD_F32 / BIAS_DROP_RES).  Same results as mmt_gemm_nt_bf16 up to fp32 summation order. */
int64_t mmt_gemm_nt_splitk_workspace_floats(int M, int N, int K);
int mmt_gemm_nt_splitk(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int M, int N, int K,
                       int epilogue, const MmtEpilogue* epi, float* ws, void* stream);
/* The same with explicit control: splits (<= 0: up to 16), wide != 0: 128x128 tiles (N % 128 == 0), n_rows_dev (nullable):
 * device count of live rows, no_epilogue != 0: only the partial slabs are produced (slab s at ws + s*round_up(M,128)*N,
 * leading dimension N) for a consumer that sums them itself. */
int mmt_gemm_nt_splitk_ex(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int M, int N, int K,
                          int epilogue, const MmtEpilogue* epi, float* ws, int splits, int wide, const int32_t* n_rows_dev,
                          int no_epilogue, void* stream);
/* NN operand form: C[M,N] = A[M,K] . B[K,N] with B row-major [K, N] -- a weight W[out, in] as stored, used for the
 * input gradient dX = dY . W (autograd of every nn.Linear of model/bert.py:146-150,186,218,234) without a transposed
 * copy: the tile is staged as it lies and the MFMA fragments come from ds_read_b64_tr_b16.  Epilogues: MMT_EPI_BF16,
 * MMT_EPI_F32, MMT_EPI_ADD_F32, MMT_EPI_DGELU (no colsum).  Same alignment rules as mmt_gemm_nt_bf16; N % 64 == 0. */
int mmt_gemm_nn_bf16(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int M, int N, int K,
                     int epilogue, const MmtEpilogue* epi, const int32_t* n_rows_dev, void* stream);
int mmt_gemm_nn_splitk_ex(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int M, int N, int K,
                          int epilogue, const MmtEpilogue* epi, float* ws, int splits, int wide, const int32_t* n_rows_dev,
                          int no_epilogue, void* stream);

/* Several independent C_i[M_i,N_i] = A_i . B_i^T (+ bias_i) in ONE launch (epilogue MMT_EPI_BIAS_F32 / MMT_EPI_F32):
 * the per-expert ReduceDim.fc projections of model/model.py:426-437 (seven GEMMs with different K). */
#define MMT_GEMM_GROUP_MAX 16
typedef struct MmtGemmItem {
  const void* A; const void* B; void* C; const float* bias; /* bf16 [M,lda], bf16 [N,ldb], fp32 [M,ldc], fp32 [N] */
  int64_t lda, ldb, ldc;
  int32_t M, N, K, tile_begin;
  const int32_t* n_rows_dev; /* nullable: live rows of this problem on the device (<= M); tiles past them exit */
} MmtGemmItem;
int mmt_gemm_nt_grouped(const MmtGemmItem* items, int n, int epilogue, void* stream);

/* dW[N,K2] (+)= sum_rows A[rows,N]^T . B[rows,K2]   (weight gradients: contraction over tokens).
 * A, B are row-major bf16 [rows, *]; the result is written as fp32 `splits` partial slabs
 * ws[splits][N*K2] which mmt_reduce_slabs sums.  Replaces autograd's weight-gradient mm for every
 * nn.Linear on the path.  N % 128 == 0, K2 % 128 == 0. */
int mmt_gemm_tn_bf16(const void* A, int64_t lda, const void* B, int64_t ldb, float* ws,
                     int rows, int N, int K2, int splits, const int32_t* n_rows_dev, void* stream);
int mmt_reduce_slabs(const float* ws, int splits, int64_t count, float* out, int accumulate,
                     void* stream);
/* the same for a weight-gradient slab set and its bias-gradient slab set in ONE launch (out = sum of `splits` slabs) */
int mmt_reduce_slabs_pair(const float* ws_a, int64_t count_a, float* out_a, const float* ws_b, int64_t count_b,
                          float* out_b, int splits, void* stream);

/* Grouped weight gradients: for every item, out[N_out, K2_out] = sum_rows A[rows,N]^T . B[rows,K2] written
 * directly as fp32 (no split-K slabs), and bias_out[n] = sum_rows A[rows, n] (nullable) -- all items in ONE
 * launch, one 128x128 tile per block.  Replaces autograd's weight AND bias gradients of the four nn.Linear of a
 * BertLayer (bert.py:137-139,186,218,234) resp. of all ReduceDim.fc (model.py:724).  N % 128 == 0, K2 % 128 == 0
 * (operands padded); N_out / K2_out / ldo (0 = N / K2 / K2_out) un-pad the stored result. */
#define MMT_WGRAD_MAX 16
typedef struct MmtWgradItem {
  const void* A;   /* bf16 [rows, lda] : dY                                                          */
  const void* B;   /* bf16 [rows, ldb] : X                                                           */
  float* out;      /* fp32 [N_out, ldo]                                                              */
  float* bias_out; /* fp32 [N_out] or NULL                                                           */
  int64_t lda, ldb, ldo;
  int32_t N, K2, N_out, K2_out, tile_begin;
  int32_t reserved; /* > 0: this item contracts over exactly `reserved` rows (overrides rows / n_rows_dev)     */
  float* slab;      /* splits > 1: partial results [splits][N_out, ldo] (summed by the caller, e.g.            */
  float* bias_slab; /*   mmt_col_reduce_multi) and partial bias gradients [splits][N_out]                       */
  int32_t splits, reserved2; /* reserved2: set by mmt_wgrad_grouped (tile patch shape), callers leave it 0 */
  const int32_t* n_rows_dev; /* nullable: this item contracts over *n_rows_dev rows (device; overrides the group's) */
} MmtWgradItem;
typedef struct MmtWgradGroup {
  MmtWgradItem item[MMT_WGRAD_MAX];
  const int32_t* n_rows_dev; /* nullable live row count on device                                    */
  int32_t count, rows;
} MmtWgradGroup;
int mmt_wgrad_grouped(const MmtWgradGroup* g, void* stream);

/* out[i] (+)= sum_s ws[s][i] over a [rows, cols_ws] slab keeping only the first cols_out columns
 * (un-pads the K-padded ReduceDim weight gradients). */
int mmt_reduce_slabs_2d(const float* ws, int splits, int rows, int cols_ws, int cols_out, float* out,
                        int accumulate, void* stream);

/* ---- LayerNorm / embeddings (norm.hip) ---------------------------------------------------------
 * h = LN(z) over the last dim, fp32 statistics.  bert.py:188,236 (BertSelfOutput / BertOutput
 * layer_norm; z is the pre-LN sum written by MMT_EPI_BIAS_DROP_RES).  d % 256 == 0, d <= 1024. */
int mmt_ln_fwd(const float* z, const float* gamma, const float* beta, float eps, float* h32, void* h16,
               float* mean, float* rstd, int rows, int d, const int32_t* n_rows_dev, void* stream);
/* The hidden -> hidden projection of BertSelfOutput WITH its LayerNorm in one launch (gemm_ln.hip; r05):
 *   z = dropout(A[M,K] . W[N,K]^T + bias) + res ;  h = LN(z)          bert.py:185-188
 * = mmt_gemm_nt_bf16(MMT_EPI_BIAS_DROP_RES) followed by mmt_ln_fwd, same results.  N == 512 (a block owns 32 whole rows),
 * K % 64 == 0.  res / z_out / h32 fp32 [M, 512] (z_out / h32 contiguous, h32 nullable), h16 bf16 [M, 512], mean / rstd [M];
 * dropout element (row_index[row] or row, column) of the stream hash(drop_key, *seed_dev), drop_thr16 = 0: none. */
int mmt_gemm_nt_ln_fwd(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, const float* res,
                       int64_t ldres, const int32_t* row_index, uint32_t drop_key, uint32_t drop_thr16, float drop_scale,
                       const uint32_t* seed_dev, float* z_out, const float* gamma, const float* beta, float eps, float* h32,
                       void* h16, float* mean, float* rstd, int M, int N, int K, const int32_t* n_rows_dev, void* stream);
/* h = dropout(LN(features + type_emb[type_ids] + pos_emb[pos_ids]))   bert.py:87-105.
 * z_save receives the pre-LN sum (needed by backward). pos_ids may be NULL (pos_enc='none'). */
int mmt_embed_ln_fwd(const float* features, const int32_t* type_ids, const int32_t* pos_ids,
                     const float* type_emb, const float* pos_emb, float* z_save, const float* gamma,
                     const float* beta, float eps, float* h32, void* h16, float* mean, float* rstd,
                     int rows, int d, const int32_t* n_rows_dev, const int32_t* row_index,
                     uint32_t drop_key, uint32_t thr16, float drop_scale, const uint32_t* seed_dev,
                     void* stream);
/* the same launch carrying one extra block that writes the attention backward's block order of this batch (sched_work
 * nullable = no rider; arguments as for mmt_attn_schedule with tiles = ceil(S / 64)) */
int mmt_embed_ln_fwd_sched(const float* features, const int32_t* type_ids, const int32_t* pos_ids,
                           const float* type_emb, const float* pos_emb, float* z_save, const float* gamma,
                           const float* beta, float eps, float* h32, void* h16, float* mean, float* rstd,
                           int rows, int d, const int32_t* n_rows_dev, const int32_t* row_index,
                           uint32_t drop_key, uint32_t thr16, float drop_scale, const uint32_t* seed_dev,
                           const int32_t* sched_cu, int sched_B, int sched_H, int sched_tiles, int32_t* sched_work,
                           void* stream);
/* Compact-row variant: LN over `rows` compact rows; fp32 row i goes to h32[dst_rows[i]] (h16 compact, nullable). */
int mmt_ln_fwd_scatter(const float* z, const float* gamma, const float* beta, float eps, float* h32,
                       const int32_t* dst_rows, void* h16, float* mean, float* rstd, int rows, int d, void* stream);
/* dst[i] = src[rows[i]] (+ idx_out[i] = idx_in ? idx_in[rows[i]] : rows[i]) / dst[rows[i]] (+)= src[i]; fp32 rows of d
 * (rows must be distinct for the accumulating scatter). */
int mmt_rows_gather(const float* src, const int32_t* rows, int n, int d, float* dst, const int32_t* idx_in,
                    int32_t* idx_out, void* stream);
int mmt_rows_scatter(const float* src, const int32_t* rows, int n, int d, float* dst, int accumulate, void* stream);
/* Embedding-table gradient written directly (dtable[v] = sum of g[r] over token rows r < min(rows, *n_rows_dev) with
 * ids[r] == v, row order; rows of unused ids are zeroed): for large tables with few ids in use (BERT-base's 512-row
 * position table).  The small video-BERT tables use mmt_table_grad_partials + mmt_col_reduce. */
int mmt_table_grad_direct(const float* g, const int32_t* ids, int rows, int d, int vocab, const int32_t* n_rows_dev,
                          float* dtable, void* stream);
/* Word-embedding gradient of the text tower (HF BertEmbeddings.word_embeddings, reached from model/model.py:371-376):
 * dtable[id] = sum of g[i] over token rows i with ids[i] == id, in row order (deterministic); rows with
 * id == padding_idx or outside [0, vocab) contribute nothing; n_rows_dev (nullable) bounds n on the device.  dtable
 * [vocab, d] must be zero on entry.  The forward
 * lookup is mmt_rows_gather(table, ids, ...). */
int mmt_embedding_grad(const float* g, const int32_t* ids, int n, int d, int vocab, int padding_idx,
                       const int32_t* n_rows_dev, float* dtable, void* stream);
/* Text-side token packing (drops the padded tokens of model/model.py:353-369; valid when only the [CLS] row of the
 * last layer is read, :378-379).  int64 [B, W] inputs (token_type_ids nullable = 0, position_ids nullable = 0..W-1);
 * int32 outputs: counts [B], cu_seqlens [B+1], n_rows_dev [1], ids/types/pos/row_index [>= B*W] (first *n_rows_dev
 * entries written; row_index = dense coordinate b*W + t), cls_rows [B] = first kept row of each sample. */
int mmt_text_plan(const int64_t* input_ids, const int64_t* token_type_ids, const int64_t* position_ids,
                  const int64_t* attention_mask, int B, int W, int32_t* counts, int32_t* cu_seqlens, int32_t* n_rows_dev,
                  int32_t* ids, int32_t* types, int32_t* pos, int32_t* row_index, int32_t* cls_rows, void* stream);
/* The reduction of split-K partial slabs (mmt_gemm_nt_splitk_ex with no_epilogue = 1; geometry from
 * mmt_gemm_splitk_geometry) folded into the LayerNorm pass next to it -- the compact last encoder layer, where a launch
 * costs more than the work it carries:
 *   mmt_splitk_ln_fwd : z = sum_s slab_s + bias -> dropout -> + residual (res[res_rows ? res_rows[i] : i]); h = LN(z)
 *                       (bert.py:185-189 / 233-237).  RNG row coordinate: rowidx[i] if given, else
 *                       row_index[res_rows[i]] (written to rowidx_out for the kernels that follow).
 *   mmt_ln_bwd_slabs  : mmt_ln_bwd with dout = sum_s slab_s + res. */
int mmt_gemm_splitk_geometry(int M, int N, int K, int splits_requested, int* splits, int64_t* slab_stride);
int mmt_splitk_ln_fwd(const float* slabs, int splits, int64_t slab_stride, const float* bias, const float* res,
                      const int32_t* res_rows, const int32_t* rowidx, const int32_t* row_index, int32_t* rowidx_out,
                      uint32_t drop_key, uint32_t thr16, float drop_scale, const uint32_t* seed_dev, float* z_out,
                      const float* gamma, const float* beta, float eps, float* h32, void* h16, float* mean, float* rstd,
                      int rows, int d, void* stream);
int mmt_ln_bwd_slabs(const float* slabs, int splits, int64_t slab_stride, const float* res, const float* z,
                     const float* mean, const float* rstd, const float* gamma, float* dz, void* dy, float* partials,
                     int rows, int d, int drop_mode, const int32_t* row_index, uint32_t drop_key, uint32_t thr16,
                     float drop_scale, const uint32_t* seed_dev, void* stream);
/* the same on a packed batch: n_rows_dev (nullable) = device count of live rows, rows = the capacity launched for */
int mmt_splitk_ln_fwd_ex(const float* slabs, int splits, int64_t slab_stride, const float* bias, const float* res,
                         const int32_t* res_rows, const int32_t* rowidx, const int32_t* row_index, int32_t* rowidx_out,
                         uint32_t drop_key, uint32_t thr16, float drop_scale, const uint32_t* seed_dev, float* z_out,
                         const float* gamma, const float* beta, float eps, float* h32, void* h16, float* mean, float* rstd,
                         int rows, int d, const int32_t* n_rows_dev, void* stream);
int mmt_ln_bwd_slabs_ex(const float* slabs, int splits, int64_t slab_stride, const float* res, const float* z,
                        const float* mean, const float* rstd, const float* gamma, float* dz, void* dy, float* partials,
                        int rows, int d, int drop_mode, const int32_t* n_rows_dev, const int32_t* row_index,
                        uint32_t drop_key, uint32_t thr16, float drop_scale, const uint32_t* seed_dev, void* stream);

/* LayerNorm backward.  drop_mode 0: none; 1: the LN input was dropout(y)+res -> dy(bf16) = mask*dz*scale;
 * 2: dropout followed the LN (embeddings) -> dout is masked first.  `partials` receives
 * [ceil(rows/rpb)][3][d] per-block column sums (dgamma, dbeta, dbias) for mmt_col_reduce. */
int mmt_ln_bwd_rows_per_block(int rows);
int mmt_ln_bwd(const float* dout, const float* z, const float* mean, const float* rstd, const float* gamma,
               float* dz, void* dy, float* partials, int rows, int d, int drop_mode,
               const int32_t* n_rows_dev, const int32_t* row_index, uint32_t drop_key, uint32_t thr16,
               float drop_scale, const uint32_t* seed_dev, void* stream);
/* out_j[c] (+)= sum_b partials[b][j][c], j < nvec <= 4 (NULL outputs are skipped); fixed order. */
int mmt_col_reduce(const float* partials, int nblocks, int nvec, int d, float* out0, float* out1,
                   float* out2, float* out3, int accumulate, void* stream);
/* Batched form: every job j writes out[k][c] = sum_b partials[b][k][c] (k < nout <= nvec) in ONE launch. */
#define MMT_COLRED_MAX 32
typedef struct MmtColReduceJob {
  const float* partials; /* [nblocks][nvec][d]                                                      */
  float* out[4];         /* nout outputs of d floats each (NULL entries are skipped)                */
  int32_t nblocks, nvec, nout, d;
} MmtColReduceJob;
int mmt_col_reduce_multi(const MmtColReduceJob* jobs, int n, void* stream);
/* dtable[v] (+)= sum of g[row] over rows with ids[row] == v: gradient of nn.Embedding (bert.py:78-81)
 * as a deterministic segmented sum (no atomics); scratch = mmt_table_grad_scratch_floats() floats. */
int64_t mmt_table_grad_scratch_floats(int vocab, int d);
int mmt_table_grad(const float* g, const int32_t* ids, int rows, int d, int vocab,
                   const int32_t* n_rows_dev, float* scratch, float* dtable, int accumulate, void* stream);
/* The two halves of mmt_table_grad: chunk partials [mmt_table_grad_chunks()][vocab*d] into `scratch`, to be summed
 * by mmt_col_reduce / mmt_col_reduce_multi (nvec = 1, d = vocab*d). */
int mmt_table_grad_partials(const float* g, const int32_t* ids, int rows, int d, int vocab,
                            const int32_t* n_rows_dev, float* scratch, void* stream);
/* both tables of the video BERT (ids1 nullable) as one-hot x gradient products on the exact-fp32 matrix cores, one launch;
 * same scratch layout as mmt_table_grad_partials ([chunks][vocab][d]); d % 128 == 0, vocabularies of at most 128 rows */
int mmt_table_grad_partials_pair(const float* g, const int32_t* ids0, int vocab0, float* scratch0, const int32_t* ids1,
                                 int vocab1, float* scratch1, int rows, int d, const int32_t* n_rows_dev, void* stream);
int mmt_table_grad_chunks(void);
/* partials[blk][c] = column sums of a bf16 matrix over 32-row blocks (bias gradients). */
int mmt_colsum_bf16(const void* x, int64_t ld, int rows, int cols, const int32_t* n_rows_dev,
                    float* partials, void* stream);

/* ---- fused attention (attention.hip), head dim 128 ----------------------------------------------
 * Replaces bert.py:141-168: softmax(QK^T/sqrt(dh) + mask_bias[key]) -> dropout -> .V, merged heads.
 * qkv bf16 [rows, 3d]; sample b owns rows cu_seqlens[b]..cu_seqlens[b+1] (NULL => b*S..(b+1)*S);
 * mask_bias fp32 [rows] is the additive key mask (0 / -10000, bert.py:395); lse fp32 [rows, H]. */
int mmt_attn_fwd(const void* qkv, const int32_t* cu_seqlens, const float* mask_bias, void* ctx, float* lse,
                 int B, int S, int H, int d, float scale, uint32_t drop_key, uint32_t thr16,
                 float drop_scale, const uint32_t* seed_dev,
                 const int32_t* row_index, void* stream);
/* Backward of the above (autograd of bert.py:141-168): dqkv bf16 [rows, 3d].
 * delta fp32 [rows, d/64]: sums of dctx * ctx over the 64-column groups of a row (delta of head h = its DH/64 groups).
 * mmt_attn_bwd forms them itself (scratch, one extra launch); mmt_attn_bwd_ex(delta_ready = 1) takes them as INPUT --
 * the epilogue of the GEMM that produced dctx writes them (MmtEpilogue.dot_src / dot_out). */
/* work (nullable, packed batches with (B * H) % 8 == 0): the block order of THIS batch, mmt_attn_schedule_words(B, S, H) int32
 * words written by mmt_attn_schedule (or by the engine, riding in its embedding LayerNorm launch): longest blocks first,
 * empty slots at the end of the grid, every block of a (sample, head) on one XCD.  Same results as without. */
int mmt_attn_bwd_ex(const void* qkv, const int32_t* cu_seqlens, const float* mask_bias, const void* ctx,
                    const float* lse, const void* dctx, void* dqkv, float* delta, int delta_ready, int B, int S, int H,
                    int d, float scale, uint32_t drop_key, uint32_t thr16, float drop_scale,
                    const uint32_t* seed_dev, const int32_t* row_index, const int32_t* work, void* stream);
int64_t mmt_attn_schedule_words(int B, int S, int H);
int mmt_attn_schedule(const int32_t* cu_seqlens, int B, int S, int H, int32_t* work, void* stream);
int mmt_attn_bwd(const void* qkv, const int32_t* cu_seqlens, const float* mask_bias, const void* ctx,
                 const float* lse, const void* dctx, void* dqkv, float* delta, int B, int S, int H, int d,
                 float scale, uint32_t drop_key, uint32_t thr16, float drop_scale,
                 const uint32_t* seed_dev,
                 const int32_t* row_index, void* stream);
/* Query-subset attention (last encoder layer: only the rows that are read out need a context vector): queries are the
 * rows qsel[b*nq + i]; ctx / lse / dctx / delta are compact [B*nq, .] (delta: [B*nq, d/64]); qkv / dqkv keep the full layout.  dqkv must be
 * zero on entry in the Q section of the non-selected rows. */
int mmt_attn_fwd_rows(const void* qkv, const int32_t* cu_seqlens, const float* mask_bias, const int32_t* qsel, int nq,
                      void* ctx, float* lse, int B, int S, int H, int d, float scale, uint32_t drop_key, uint32_t thr16,
                      float drop_scale, const uint32_t* seed_dev,
                 const int32_t* row_index, void* stream);
int mmt_attn_bwd_rows_ex(const void* qkv, const int32_t* cu_seqlens, const float* mask_bias, const int32_t* qsel, int nq,
                         const void* ctx, const float* lse, const void* dctx, void* dqkv, float* delta, int delta_ready,
                         int B, int S, int H, int d, float scale, uint32_t drop_key, uint32_t thr16, float drop_scale,
                         const uint32_t* seed_dev, const int32_t* row_index, void* stream);
int mmt_attn_bwd_rows(const void* qkv, const int32_t* cu_seqlens, const float* mask_bias, const int32_t* qsel, int nq,
                      const void* ctx, const float* lse, const void* dctx, void* dqkv, float* delta, int B, int S, int H,
                      int d, float scale, uint32_t drop_key, uint32_t thr16, float drop_scale,
                      const uint32_t* seed_dev,
                 const int32_t* row_index, void* stream);
/* (all four) row_index (nullable): token packing -- row_index[row] = b*S + original position of the token; the
 * attention-probability dropout mask is keyed on ORIGINAL (query, key) positions, so packed and dense runs draw the same
 * mask. */
/* Test helper: the keep-mask mmt_attn_fwd draws, uint8 [B,H,S,S] (dense layout). */
int mmt_attn_dropout_mask(uint8_t* out, int B, int H, int S, uint32_t drop_key, uint32_t thr16,
                          const uint32_t* seed_dev, void* stream);

/* ---- parameter packing / optimizer (elementwise.hip) --------------------------------------------- */
#define MMT_PACK_MAX 24
typedef struct MmtPackItem {
  const float* src;  /* fp32 master [rows, cols]                                                   */
  void* dst;         /* bf16 [rows, dst_ld], columns cols..dst_ld zero-filled                      */
  void* dst_t;       /* nullable bf16 transposed copy [dst_t_rows >= cols, dst_t_ld >= rows]       */
  int32_t rows, cols, dst_ld, dst_t_ld, dst_t_rows, reserved;
} MmtPackItem;
/* bf16 shadows of the fp32 master weights (and W^T copies for the input-gradient GEMMs). */
int mmt_pack_weights(const MmtPackItem* items, int n, void* stream);
/* torch.optim.Adam step (train.py:100) over one flat fp32 buffer; step_dev = 1-based step on device.
 * lr_dev (nullable): device float that overrides `lr` -- the learning-rate schedule (StepLR + warm-up,
 * trainer/trainer.py:150-160, train.py:101-103) then works under HIP-graph replay without re-capturing. */
int mmt_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t count,
                  float lr, float beta1, float beta2, float eps, float weight_decay,
                  const int32_t* step_dev, const float* lr_dev, void* stream);

/* The same optimizer step (train.py:100) that ALSO refreshes the bf16 shadows of the GEMM weights (and their W^T
 * copies), so that no re-packing pass over the weights runs between optimizer and the next forward.  The flat buffer is
 * described as segments in ascending order that tile [0, count): dst == NULL: a plain span of `count` elements;
 * otherwise the fp32 matrix [rows, cols] at `offset` (cols % 4 == 0) whose bf16 copy dst [rows, dst_ld] and optional
 * transpose dst_t [cols, dst_t_ld >= rows, % 8 == 0] are rewritten from the updated values.  segs_host / segs_dev: the
 * same table in host memory (grid layout) and device memory (read by the kernel; it must stay valid under graph replay). */
#define MMT_ADAM_SEG_MAX 160
typedef struct MmtAdamSeg {
  int64_t offset, count;
  void* dst;
  void* dst_t;
  int32_t rows, cols, dst_ld, dst_t_ld;
} MmtAdamSeg;
int mmt_adam_fused_blocks(const MmtAdamSeg* seg);
int mmt_adam_step_fused(float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                        const MmtAdamSeg* segs_host, const MmtAdamSeg* segs_dev, int n_segs, float lr, float beta1,
                        float beta2, float eps, float weight_decay, int32_t* step_dev, const float* lr_dev,
                        int bump_step, void* stream);
/* ---- Adam riders (r06): the optimizer step OFF the critical path of a single-rank step -------------------------------
 * train.py:100 + trainer/trainer.py:203-204 run `optimizer.step()` after the whole backward; the fused kernel above is
 * HBM-bound (0.8 GB per step at config B: 119 us of a 1.3 ms step) and strictly serial behind it.  Here the same update is
 * cut into the fused kernel's units of work (one 64x64 tile of a shadowed matrix or 4096 elements of a plain span) and
 * put in a QUEUE ordered by the stage of the backward at which a unit's gradients are final.  The GEMM launches of the
 * backward carry the queue as a "rider" (MmtEpilogue.rider): blocks with no tile of their own take units off the queue
 * while the launch's GEMM blocks are still running (idle CUs / the last partial round of tiles stream Adam's bytes under
 * MFMA-bound work), and stop as soon as the first block of the launch's last round has finished.  mmt_adam_step_queue
 * runs whatever is left (units nobody took, and units that are never ready early) and advances the step count: weights,
 * moments and bf16 shadows after it are BIT-IDENTICAL to mmt_adam_step_fused (same per-element arithmetic, adam_unit.h).
 * All fields of MmtAdamQueue are device pointers / values; the struct itself lives in device memory (kernels read it
 * through MmtEpilogue.rider) with a host copy for the launcher. */
#define MMT_RIDER_SLOTS 1024
#define MMT_RIDER_STAGES 72
#define MMT_RIDER_STATE_WORDS (MMT_RIDER_STAGES + 1 + MMT_RIDER_SLOTS + 64 + 64 * 64)
typedef struct MmtAdamQueue {
  float *p, *m, *v;           /* flat master weights and Adam moments                                             */
  const float* g;             /* flat gradients                                                                   */
  const MmtAdamSeg* segs;     /* device segment table (as mmt_adam_step_fused)                                    */
  const int32_t* unit_seg;    /* [n_units] queue entry k -> segment                                               */
  const int32_t* unit_blk;    /* [n_units] queue entry k -> block inside the segment (tile / 4096-element chunk)  */
  int32_t* state;             /* int32[MMT_RIDER_STATE_WORDS]: [s] entries of stage s CLAIMED this step (a plain
                               * fetch-add: may overshoot the stage's size), [MMT_RIDER_STAGES] ticket, then the
                               * finished-block counters of the hosting launches, two statistics words (entries riders
                               * ran, steps; never reset) and 64 first-level ticket words on lines of their own; all but
                               * the statistics are zero between steps                                             */
  int32_t* step_dev;          /* int32[2] = {steps taken so far, unused}                                          */
  const float* lr_dev;        /* nullable: device learning rate                                                   */
  const struct MmtAdamQueue* chain;  /* nullable: a second queue (another flat buffer) whose first chain_stages stages
                               * a rider block drains FIRST (the text tower's leftovers under the video backward)  */
  float lr, beta1, beta2, eps, weight_decay;
  int32_t n_units;
  int32_t chain_stages;
  int32_t n_stages;           /* stages whose entries may be ridden; entries [stage_begin[n_stages], n_units) are only
                               * final when the backward ends and always run in mmt_adam_step_queue                */
  int32_t stage_begin[MMT_RIDER_STAGES + 1];  /* entries of stage s: [stage_begin[s], stage_begin[s + 1])        */
} MmtAdamQueue;
/* The rest of the step's optimizer work + the step count: entries [state[0], n_units) of the queue, one block each
 * (entries already taken exit at once); the last block to finish zeroes the queue state and stores steps + 1. */
int mmt_adam_step_queue(const MmtAdamQueue* q_host, const MmtAdamQueue* q_dev, void* stream);
/* Measurement / test hook: `blocks` rider blocks of 512 threads with nothing else to do drain the first `limit` stages. */
int mmt_adam_rider_probe(const MmtAdamQueue* q_dev, int limit, int blocks, void* stream);

/* bump_step = 0: step_dev[0] is this launch's step number t (bias correction), as mmt_adam_step.
 * bump_step = 1: step_dev is int32[2] = {steps taken so far, 0}; the launch is step step_dev[0] + 1 and stores it.
 * bump_step = 2: the launch is step step_dev[0] + 1 as well but leaves the count alone: one optimizer step issued as
 *   several launches over sub-ranges of the segment table (each as soon as its gradients are final), the last with 1. */

/* ---- video tokens (assemble.hip) -------------------------------------------------------------------
 * model.py:426-437 (ReduceDim per expert) + :485-567 (token assembly), see assemble.hip.
 * mmt_video_plan: seed_bump (nullable) = the per-step dropout seed word of the encoder, incremented by the plan kernel
 * (a training forward needs a fresh seed anyway: one launch less than a separate increment). */
#define MMT_MAX_EXPERTS 16
typedef struct MmtExpertIO {
  const float* feat;     /* [B, T, D] fp32  features[mod]                                        */
  const float* maxpool;  /* [B, D] fp32     features_maxpool[mod]                                */
  const float* ind;      /* [B, T] fp32     features_ind[mod]                                    */
  const float* t;        /* [B, T] fp32     features_t[mod]                                      */
  void* x;               /* bf16 [rows_pad, Dpad] GEMM input, COMPACT rows (see MmtVideoSrc)              */
  float* y;              /* fp32 [rows_pad, d] ReduceDim.fc output (pre-normalisation)           */
  void* dy;              /* bf16 [rows_pad, d] gradient wrt y (backward)                         */
  int32_t D, Dpad, type_idx, rows_pad;
  /* K-split of a wide expert's projection (rgb 2048, scene 2208 ...): y holds the first K chunk's product + bias,
   * y_part[i] the other chunks' partial products (same shape as y); the scatter kernels add them up. */
  const float* y_part[2];
  int32_t n_part, reserved;
} MmtExpertIO;
/* Token plan: slot[b*S+s] -> row (or -1), cu_seqlens[B+1], *n_rows_dev, and per-row row_index (b*S+s),
 * type_ids, pos_ids (clamp(features_t,0,max_pos) model.py:516-520), mask_bias, agg_row[b*M+m].
 * pack=0 keeps all S=1+M*(T+1) slots; pack=1 drops padded FEA tokens (exact, see assemble.hip). */
/* Source-row compaction maps written by mmt_video_plan (device): the ReduceDim projections only see live rows.
 * Compact source matrix of expert e (x / y / dy of MmtExpertIO): rows [0, B) = the max-pooled vectors, rows B + i = the
 * valid feature rows in (sample, time) order. */
typedef struct MmtVideoSrc {
  int32_t* src_row; /* [token rows] compact source row of every live token inside its expert's matrix (-1: CLS) */
  int32_t* src_cnt; /* [M] live source rows per expert = B + valid feature rows                                   */
  int32_t* xsrc;    /* [M, B*T] feature row b*T + t behind compact row B + i                                      */
} MmtVideoSrc;
int mmt_video_plan(const MmtExpertIO* experts, int M, int B, int T, int pack, int max_pos, int32_t* counts,
                   int32_t* cu_seqlens, int32_t* n_rows_dev, int32_t* slot, int32_t* row_index, int32_t* type_ids,
                   int32_t* pos_ids, float* mask_bias, int32_t* agg_row, uint32_t* seed_bump, const MmtVideoSrc* src,
                   void* stream);
int mmt_video_cast(const MmtExpertIO* experts, int M, int B, int T, const MmtVideoSrc* src, void* stream);
int mmt_video_scatter(const MmtExpertIO* experts, int M, int B, int T, int d, const int32_t* n_rows_dev,
                      const int32_t* row_index, const MmtVideoSrc* src, float* features, void* stream);
int mmt_video_scatter_bwd(const MmtExpertIO* experts, int M, int B, int T, int d, const int32_t* n_rows_dev,
                          const int32_t* row_index, const MmtVideoSrc* src, const float* dfeatures, void* stream);

/* ---- read-out, similarity, losses (simloss.hip) --------------------------------------------------- */
/* vid_embds[i] = F.normalize(last_hidden[agg_row[i]]), i < B*M      model.py:583-587,621-625 */
int mmt_readout_fwd(const float* last_hidden, const int32_t* agg_row, int BM, int d, float* vid_embds,
                    float* inv_norm, void* stream);
/* writes the B*M AGG rows of dlast_hidden (all other rows must be zero beforehand) */
int mmt_readout_bwd(const float* vid_embds, const float* inv_norm, const float* dvid_embds,
                    const int32_t* agg_row, int BM, int d, float* dlast_hidden, void* stream);
/* sharded_cross_view_inner_product, model.py:789-837: txt [NT,M,d], vid [NV,M,d], tw [NT,M], vw [NV,M]
 * -> sims [NT,NV] (rows = text); dots = NT*NV*M floats saved for the backward (its layout is private to the pair of
 * calls; mmt_sims_bwd OVERWRITES it).  From 64 rows/columns on (the global batch of a data-parallel step) the per-expert
 * dot products and both embedding gradients run as batched exact-fp32 MFMA GEMMs. */
int mmt_sims_fwd(const float* txt, const float* vid, const float* tw, const float* vw, int NT, int NV, int M,
                 int d, float* sims, float* dots, void* stream);
int mmt_sims_bwd(const float* txt, const float* vid, const float* tw, const float* vw, float* dots,
                 const float* dsims, int NT, int NV, int M, int d, float* dtxt, float* dvid, float* dtw,
                 float* dvw, void* stream);
/* Small batches (n <= mmt_simloss_small_max_n() pairs, one caption per video -- the single-rank training step): the loss
 * (kind 0: MaxMarginRankingLoss(margin, fix_norm), loss.py:38-65; kind 1: InfoNceLoss, :68-81), its gradient wrt the
 * similarity matrix and the whole similarity backward (model.py:789-837) in ONE launch from the sims / dots of
 * mmt_sims_fwd.  dtxt/dvid [n, M, d], dtw/dvw [n, M] (each nullable).  dlast (nullable) + inv_norm [n*M]: also the
 * backward of the read-out normalisation (model.py:621-625): dlast[out_rows ? out_rows[i] : i] for i = v*M + m. */
int mmt_simloss_small_max_n(void);
int mmt_simloss_bwd_small(const float* txt, const float* vid, const float* tw, const float* vw, const float* sims,
                          const float* dots, int n, int M, int d, int kind, float margin, int fix_norm, float* loss,
                          float* dtxt, float* dvid, float* dtw, float* dvw, const float* inv_norm,
                          const int32_t* out_rows, float* dlast, void* stream);

/* MaxMarginRankingLoss.forward (loss.py:38-65): loss scalar + d loss/d sims; partial = n floats scratch. */
int mmt_maxmargin(const float* sims, int n, float margin, int fix_norm, float* partial, float* loss,
                  float* grad, void* stream);
/* InfoNceLoss.forward (loss.py:68-81); scratch = 3n floats. */
int mmt_infonce(const float* sims, int n, float* scratch, float* loss, float* grad, void* stream);

/* ---- evaluation path at scale (retrieval.hip): SURVEY.md 8f.1 ------------------------------------------------
 * Replaces the CPU-side sharded_cross_view_inner_product + numpy ranking of trainer/trainer.py:396-447.
 * mmt_sims_eval: same maths as mmt_sims_fwd but as one exact-fp32 MFMA GEMM over K = M*d (no [NT,NV,M] tensor, no
 * backward); ws = mmt_sims_eval_workspace_floats() floats.  Text rows are b*C + cap (model.py:805,822). */
int64_t mmt_sims_eval_workspace_floats(int NT, int NV, int M, int d);
int mmt_sims_eval(const float* txt, const float* vid, const float* tw, const float* vw, int NT, int NV, int M, int d,
                  float* ws, float* sims, void* stream);
/* Tie-averaged retrieval ranks (0-based, model/metric.py:90-121, 153-243) of sims [NQ = NV*cpv, NV]:
 * t2v_rank[q] for every text query; v2t_rank[i] = best rank among video i's unmasked captions (+inf if none).
 * qmask (nullable) uint8 [NQ]: 1 = real caption (query_masks).  scratch: NQ floats. */
int mmt_retrieval_ranks(const float* sims, const uint8_t* qmask, int NQ, int NV, float* t2v_rank, float* v2t_rank,
                        float* scratch, void* stream);

/* ---- row-sharded similarity + max-margin loss for very large global batches (largesim.hip) --------------------
 * BASELINE.json configs[4] / SURVEY.md 8e: rank r owns the text rows r0..r0+b of the n x n similarity; same maths as
 * model.py:789-837 + loss.py:38-65 without the [n,n,M] weight tensor or the 2n^2 index vectors.  The two big GEMMs
 * (S = T'V'^T, P = G'V', Q = G'^T T') run on mmt_gemm_nt_bf16 / mmt_wgrad_grouped; these are the passes in between
 * (host orchestration: mmt_amd/large_sim.py).
 *   mmt_ls_fold_bf16: out16[r, m*d+c] = bf16(w[r,m] x[r,m,c]), rows R..Rpad zero.
 *   mmt_ls_finish   : S[t,v] /= sum_m tw[t,m] vw[v,m] (0 -> 1e-5), in place.
 *   mmt_ls_diag     : diag_local[t] = S[t, r0+t] / den from the RAW numerators (before the division below).
 *   mmt_ls_counts_ex: rowcnt[t] +=, colcnt[c] += (both zeroed by the caller), un-normalised hinge sums per column block
 *                     loss_part[t, cb], cb < mmt_ls_col_blocks(n); finish = 1 also divides the raw numerators (one sweep);
 *                     finish = 2 divides on the fly and leaves S raw (mmt_ls_grad_ex(raw = 1) divides again: the row block
 *                     is never rewritten).
 *   mmt_ls_counts   : the same with finish = 0.
 *   mmt_ls_grad     : G16[t,v] = bf16((dL/dS)/den) from the global counts; gs_part[t,cb,m] = sum over column block cb of
 *                     G' S vw[v,m].  mmt_ls_grad_ex: raw = 1 when S still holds the raw numerators.
 *   vw_t (nullable, mmt_ls_counts_ex / mmt_ls_grad_ex): vw transposed to [M, n] -- coalesced 16-byte loads in the sweeps.
 * n, ld, ldg multiples of 4.
 *   mmt_ls_unfold   : dx[r,m,:] = w[r,m] P[r,m*d:], dw[r,m] = <x[r,m], P[r,m]> - gsub[r,m]. */
int mmt_ls_fold_bf16(const float* x, const float* w, int R, int Rpad, int M, int d, void* out16, void* stream);
/* dst[c, r] = src[r, c], bf16, rows and cols multiples of 128 (ld in elements, multiples of 8): K-contiguous operands for the
 * NT GEMMs of the backward (V'^T, G'^T, T'^T). */
int mmt_transpose_bf16(const void* src, int64_t ld_src, int rows, int cols, void* dst, int64_t ld_dst, void* stream);
int mmt_ls_finish(float* S, int64_t ld, const float* tw, const float* vw, int b, int n, int M, void* stream);
int mmt_ls_col_blocks(int n);
int mmt_ls_diag(const float* S, int64_t ld, const float* tw, const float* vw, int b, int n, int M, int r0, float* diag_local,
                void* stream);
int mmt_ls_counts_ex(float* S, int64_t ld, const float* diag, const float* tw, const float* vw, const float* vw_t, int M, int finish, int b, int n,
                     int r0, float margin, int32_t* rowcnt, int32_t* colcnt, float* loss_part, void* stream);
int mmt_ls_counts(const float* S, int64_t ld, const float* diag, int b, int n, int r0, float margin, int32_t* rowcnt,
                  int32_t* colcnt, float* loss_part, void* stream);
int mmt_ls_grad(const float* S, int64_t ld, const float* diag, const float* tw, const float* vw, const int32_t* rowcnt,
                const int32_t* colcnt_total, int b, int n, int M, int r0, float margin, float inv_norm, void* G16,
                int64_t ldg, float* gs_part, void* stream);
int mmt_ls_grad_ex(const float* S, int64_t ld, const float* diag, const float* tw, const float* vw, const float* vw_t, const int32_t* rowcnt,
                   const int32_t* colcnt_total, int b, int n, int M, int r0, float margin, float inv_norm, void* G16,
                   int64_t ldg, float* gs_part, int raw, void* stream);
int mmt_ls_unfold(const float* P, int64_t ldp, const float* x, const float* w, const float* gsub, int R, int M, int d,
                  float* dx, float* dw, void* stream);

/* ---- text heads (texthead.hip), fp32 ------------------------------------------------------------------
 * GatedEmbeddingUnit per expert (model.py:683-702, 736-750) + text MoE weights (model.py:262-283,618),
 * batched over the M experts. */
typedef struct MmtSgemm {      /* C_b[i][j] = beta*C_b[i][j] + sum_k A_b[i*sai + k*sak] * B_b[j*sbj + k*sbk] + bias_b[j] */
  const float* A[MMT_MAX_EXPERTS];
  const float* B[MMT_MAX_EXPERTS];
  float* C[MMT_MAX_EXPERTS];
  const float* bias[MMT_MAX_EXPERTS];
  int64_t sai, sak, sbj, sbk, ldc;
  int32_t batch, M, N, K;
  float beta;
  int32_t reserved;
} MmtSgemm;
int mmt_sgemm_batched(const MmtSgemm* g, void* stream);

typedef struct MmtTextHeads {
  const float *w1[MMT_MAX_EXPERTS], *b1[MMT_MAX_EXPERTS];            /* text_GU.<mod>.fc      [d,K],[d]   */
  const float *w2[MMT_MAX_EXPERTS], *b2[MMT_MAX_EXPERTS];            /* text_GU.<mod>.cg.fc   [d,d],[d]   */
  const float *bn_gamma[MMT_MAX_EXPERTS], *bn_beta[MMT_MAX_EXPERTS]; /* cg.batch_norm weight/bias         */
  float *running_mean[MMT_MAX_EXPERTS], *running_var[MMT_MAX_EXPERTS];
  const float *moe_w[MMT_MAX_EXPERTS], *moe_b[MMT_MAX_EXPERTS];      /* moe_fc_txt.<mod>      [1,K],[1]   */
  float *g_w1[MMT_MAX_EXPERTS], *g_b1[MMT_MAX_EXPERTS], *g_w2[MMT_MAX_EXPERTS], *g_b2[MMT_MAX_EXPERTS];
  float *g_bn_gamma[MMT_MAX_EXPERTS], *g_bn_beta[MMT_MAX_EXPERTS], *g_moe_w[MMT_MAX_EXPERTS], *g_moe_b[MMT_MAX_EXPERTS];
} MmtTextHeads;
int64_t mmt_text_heads_workspace_floats(int N, int M, int d);
/* Optional extras of the small-batch path (mmt_text_heads_fast() != 0: N <= 32 caption rows, every training
 * configuration of the reference), all nullable / zero:
 *   moe_drop_*: nn.Dropout in front of the MoE logits (model.py:274) applied on the fly by the kernels that read the
 *     text (pass text_moe = NULL): stream key = hash(moe_drop_key, *seed_dev) at forward time, stored in *key_dev and
 *     re-read from there by the backward (which runs after the seed has moved on);
 *   num_batches_tracked: int64 [M], += 1 per training forward with use_bn (BatchNorm1d bookkeeping, on the device). */
/* The video side's token plan (+ feature cast) riding along as extra blocks of the text heads' first two launches
 * (N <= 32 caption rows, mmt_text_heads_fast): the arguments of mmt_video_plan / mmt_video_cast, which the caller then
 * does NOT call for this forward.  Two dependent launches (13 + 7 us under graph replay) leave the step; results are
 * those of the separate launches bit for bit (the dropout-seed bump moves from the plan to the second launch, i.e. it
 * still follows the text heads' read of the seed and precedes the encoder's). */
typedef struct MmtVideoFront {
  const MmtExpertIO* experts;
  int32_t M, B, T, pack, max_pos, do_cast;  /* do_cast = 0: the features arrive as bf16 (RaggedFeatures) */
  int32_t *counts, *cu_seqlens, *n_rows_dev, *slot, *row_index, *type_ids, *pos_ids;
  float* mask_bias;
  int32_t* agg_row;
  uint32_t* seed_bump;
  const MmtVideoSrc* src;
} MmtVideoFront;
typedef struct MmtTextHeadsOpts {
  uint32_t moe_drop_key, moe_drop_thr16;
  float moe_drop_scale;
  int32_t reserved;
  const uint32_t* seed_dev;
  uint32_t* key_dev;
  int64_t* num_batches_tracked;
  const MmtVideoFront* video_front;  /* nullable; forward only, small path only (MMT_ERR_ARG otherwise) */
} MmtTextHeadsOpts;
int mmt_text_heads_fast(int N, int M, int d, int K);
/* text [N = B*C, K] -> text_embds (B, M, C, d) L2-normalised, text_weights (B, C, M) (NULL: txt_wgh='none').
 * use_bn: txt_pro 'gbn' (1) / 'gem' (0).  training: batch statistics + running-stat update.
 * text_moe (nullable): the copy of text the MoE branch reads (after moe_txt_dropout, model.py:274). */
int mmt_text_heads_fwd(const MmtTextHeads* h, const float* text, const float* text_moe, int N, int C, int M, int d,
                       int K, int use_bn, int training, float* ws, float* text_embds, float* text_weights,
                       const MmtTextHeadsOpts* opts, void* stream);
/* gradients are WRITTEN through the g_* pointers; dtext [N, K] (nullable) receives the gradient for the text
 * tower; w1_all = the M fc.weight matrices contiguous as [M*d, K].  dtext_moe [N, K] (nullable): gradient wrt the MoE
 * branch's input (with the on-the-fly dropout: already masked, i.e. the gradient wrt text through that branch). */
int mmt_text_heads_bwd(const MmtTextHeads* h, const float* text, const float* text_moe, const float* w1_all, int N,
                       int C, int M, int d, int K, int use_bn, int training, float* ws, const float* dtext_embds,
                       const float* text_weights, const float* dtext_weights, float* dtext, float* dtext_moe,
                       const MmtTextHeadsOpts* opts, void* stream);

/* ---- whole-encoder engine (bert_engine.hip) --------------------------------------------------------
 * One call runs every kernel of model/bert.py BertModel.forward (bert.py:371-414, without the unused
 * pooler) resp. its autograd backward on `stream`.  Pointers in MmtBertLayer/MmtBertModel address the
 * caller's flat parameter / shadow / gradient buffers (mmt_amd/flat.py); `ws` is a caller-allocated
 * workspace of mmt_bert_workspace_bytes() that carries the saved activations from forward to backward. */
typedef struct MmtBertLayer {
  const void *wqkv, *wqkv_t, *wo, *wo_t, *w1, *w1_t, *w2, *w2_t;              /* bf16 shadows            */
  const float *bqkv, *bo, *ln1_g, *ln1_b, *b1, *b2, *ln2_g, *ln2_b;           /* fp32 master             */
  float *g_wqkv, *g_bqkv, *g_wo, *g_bo, *g_ln1_g, *g_ln1_b, *g_w1, *g_b1, *g_w2, *g_b2, *g_ln2_g, *g_ln2_b;
} MmtBertLayer;

typedef struct MmtBertModel {
  int32_t hidden, layers, heads, inter, max_pos, type_vocab;
  float ln_eps, p_hidden, p_attn;
  int32_t reserved;
  const float *pos_emb, *type_emb, *emb_ln_g, *emb_ln_b;
  float *g_pos_emb, *g_type_emb, *g_emb_ln_g, *g_emb_ln_b;
  const MmtBertLayer* layer; /* host array [layers] */
} MmtBertModel;

typedef struct MmtBertBatch {
  const float* features;      /* [rows_alloc, hidden] fp32 token features (bert.py:98 `features`)        */
  const int32_t* type_ids;    /* [rows] token_type_ids                                                  */
  const int32_t* pos_ids;     /* [rows] position_ids, NULL for pos_enc='none'                           */
  const float* mask_bias;     /* [rows] (1 - attention_mask) * -10000   (bert.py:386-395)               */
  const int32_t* cu_seqlens;  /* [batch+1] or NULL (dense: sample b owns rows b*seq..)                  */
  const int32_t* row_index;   /* [rows] row -> b*seq+s (RNG coordinates) or NULL (identity)             */
  const int32_t* n_rows_dev;  /* live row count on device or NULL (= rows)                              */
  const uint32_t* seed_dev;   /* per-step dropout seed on device or NULL                                */
  int32_t rows, rows_alloc, batch, seq;
  /* Optional: the only rows of sequence_output the caller will read (CENet: the AGG token of every expert,
   * model.py:583-587), out_rows[b*n_out_per_sample + i], each sample's in ascending order.  The last layer then runs
   * everything after the K/V projection on those rows only (exact: all of it is row-wise) and returns them COMPACT:
   * out_last[i] = sequence_output[out_rows[i]] for i < batch * n_out_per_sample; `dlast` is read the same way.  The
   * compact mode is taken iff batch * n_out_per_sample <= mmt_bert_tail_capacity(rows_alloc).  NULL = every row. */
  const int32_t* out_rows;
  int32_t n_out_per_sample;
  /* MMT_FORK_* bits: which launches of the BACKWARD leave `stream` for `side_stream` (they are off the critical
   * path of the step: nothing on `stream` reads what they write until the optimizer does).  0 = everything on `stream`. */
  int32_t fork;
  /* second hipStream_t of the caller (nullable = no forking).  The engine orders the two streams with events
   * (mmt_stream_fork), which become graph edges under stream capture: the forked kernels then sit on a parallel branch
   * of the captured step.  Without MMT_FORK_JOIN the work on `side_stream` is still pending when the call returns: the
   * caller joins (mmt_stream_fork(side_stream, stream)) before anything on `stream` consumes a parameter gradient.
   * Under stream CAPTURE, MMT_FORK_WGRAD without MMT_FORK_JOIN is only valid when all layer ranges of one backward are
   * captured into the SAME graph: the "weight gradients of layer l + 2 are done" events a range waits for must have been
   * recorded in the capture that waits (ranges captured as separate graphs pass MMT_FORK_JOIN on every call). */
  void* side_stream;
  /* r06 (backward only, all nullable / 0): the optimizer queue the backward's GEMM launches carry (MmtAdamQueue in device
   * memory), rider_limits[l] (HOST array [layers]) = leading stages of the queue whose gradients are final when layer l's
   * backward starts, rider_slot0 = first finished-block counter this model's launches may use (layer l uses slots
   * rider_slot0 + 8 l .. + 8 l + 7). */
  const void* rider;
  const int32_t* rider_limits;
  int32_t rider_slot0;       /* (bits 16..31: rider blocks at work per launch, 0 = the library's default) */
  /* r06: the live row count as the HOST knows it (the collator counts valid tokens before the upload), 0 = unknown.  Only
   * the tile choice reads it (the kernels read n_rows_dev); without it a packed batch is priced at rows (every tile live). */
  int32_t live_rows_hint;
} MmtBertBatch;

enum {
  MMT_FORK_WGRAD = 1,   /* the grouped weight-gradient launch of every layer (trainer/trainer.py:203: autograd's
                         * mm-backward nodes for the weights, which nothing in the remaining backward depends on)  */
  MMT_FORK_EARLY = 2,   /* ... cut in two launches issued as soon as their operands exist (FFN pair after the
                         * dGELU GEMM, attention pair after the attention backward)                               */
  MMT_FORK_REDUCE = 4,  /* LayerNorm gamma/beta and embedding-table reductions                                    */
  MMT_FORK_JOIN = 8,    /* `stream` waits for `side_stream` before mmt_bert_backward_range returns                */
  /* not a fork bit, same field: a range that ends at layer 0 stops BEFORE the embedding stage (layer 0's parameter
   * gradients are final then: a data-parallel caller starts their reduction); the embedding stage follows as its own
   * call with l_hi = l_lo = -1 */
  MMT_RANGE_LAYERS_ONLY = 16
};

/* `to` waits for everything enqueued on `from` so far (hipEventRecord + hipStreamWaitEvent on an event of an internal
 * ring).  Under stream capture this is the fork (main -> side) resp. join (side -> main) edge of the captured graph. */
int mmt_stream_fork(void* from, void* to);

int64_t mmt_bert_workspace_bytes(const MmtBertModel* m, int rows_alloc);
int mmt_bert_tail_capacity(int rows_alloc);
/* out_last: fp32 [rows_alloc, hidden] last-layer hidden states (sequence_output).  training != 0 enables
 * dropout with p_hidden / p_attn. */
int mmt_bert_forward(const MmtBertModel* m, const MmtBertBatch* b, void* ws, float* out_last, int training,
                     void* stream);
/* dlast: fp32 [rows_alloc, hidden] gradient of sequence_output (overwritten as scratch);
 * dfeatures: fp32 [rows_alloc, hidden] gradient wrt `features`; parameter gradients are WRITTEN (not
 * accumulated) through the g_* pointers. */
int mmt_bert_backward(const MmtBertModel* m, const MmtBertBatch* b, void* ws, float* dlast, float* dfeatures,
                      int training, void* stream);
/* The same backward, layers l_hi .. l_lo only (descending; the embedding stage runs with layer 0), so that a
 * data-parallel caller can start the all-reduce of a finished layer's gradients while the layers below still run
 * (the reference's DataParallel reduces after the whole backward, trainer/trainer.py:185-199).  Calls must cover
 * layers-1 .. 0 in descending order with the same dlast / dfeatures / ws.  l_hi = l_lo = -1: the embedding stage alone
 * (see MMT_RANGE_LAYERS_ONLY). */
int mmt_bert_backward_range(const MmtBertModel* m, const MmtBertBatch* b, void* ws, float* dlast, float* dfeatures,
                            int training, int l_hi, int l_lo, void* stream);

/* Measurement hook (bench.py): arm n pairs of caller-created hipEvent_t; each mmt_bert_forward then
 * records one pair around layer 0's FFN up-projection GEMM launch (the dominant kernel) on its stream.
 * The arrays must stay alive until the events have been read.  Pass NULL/0 to disarm. */
int mmt_probe_arm(void** start_events, void** stop_events, int n);
/* the same per site: 0 = FFN up-projection GEMM, 1 = FFN down-projection GEMM, 2 = grouped weight gradients, 3 / 4 = attention
 * forward / backward (all of layer 0) */
int mmt_probe_arm_site(int site, void** start_events, void** stop_events, int n);
int mmt_probe_count_site(int site);
/* y = dropout(x) (n % 4 == 0, 16-byte aligned) with the counter-based RNG of the engine: nn.Dropout in front of the text
 * MoE logits (model/model.py:274).  Forward: key = hash(drop_key, *seed_dev), stored to key_save (nullable); backward:
 * pass the saved key as key_load (the encoder advances the seed in between) and the gradient as x. */
int mmt_dropout_f32(const float* x, float* y, int64_t n, uint32_t drop_key, uint32_t thr16, float scale,
                    const uint32_t* seed_dev, uint32_t* key_save, const uint32_t* key_load, void* stream);
/* Measurement hook (tools/dispatch_lab.py): `blocks` workgroups of `threads` threads with `lds_bytes` of dynamic LDS, each
 * spinning for `spin` clock ticks -- the workgroup dispatch rate as a function of the workgroup's shape. */
int mmt_debug_dispatch_probe(int blocks, int threads, int lds_bytes, int spin, float* sink, void* stream);
int mmt_probe_count(void);

#ifdef __cplusplus
}
#endif
#endif /* MMT_HIP_H_ */
