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

#define MMT_ABI_VERSION 1
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
  uint32_t drop_key;        /* dropout stream key (seed, site, layer mixed by the host)         */
  uint32_t drop_thr16;      /* keep iff u16 >= thr16; 0 disables dropout                        */
  float drop_scale;         /* 1 / (1 - thr16/65536)                                            */
  int32_t reserved;         /* 0 = auto tile; 1 = force 128x128; 2 = force 128x64 (tests/tuning)       */
} MmtEpilogue;

/* C[M,N] = A[M,K] . B[N,K]^T  (both operands K-contiguous bf16, fp32 accumulate on MFMA).
 * Replaces every nn.Linear forward on the path (bert.py:137-139,186,218,234; model.py:724) and,
 * with B = W^T copies, the input-gradient GEMMs of their backward.
 * Requirements: K % 64 == 0, N % 64 == 0, rows allocated to a multiple of 128. */
int mmt_gemm_nt_bf16(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                     int M, int N, int K, int epilogue, const MmtEpilogue* epi,
                     const int32_t* n_rows_dev, void* stream);

/* dW[N,K2] (+)= sum_rows A[rows,N]^T . B[rows,K2]   (weight gradients: contraction over tokens).
 * A, B are row-major bf16 [rows, *]; the result is written as fp32 `splits` partial slabs
 * ws[splits][N*K2] which mmt_reduce_slabs sums.  Replaces autograd's weight-gradient mm for every
 * nn.Linear on the path.  N % 128 == 0, K2 % 128 == 0. */
int mmt_gemm_tn_bf16(const void* A, int64_t lda, const void* B, int64_t ldb, float* ws,
                     int rows, int N, int K2, int splits, const int32_t* n_rows_dev, void* stream);
int mmt_reduce_slabs(const float* ws, int splits, int64_t count, float* out, int accumulate,
                     void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MMT_HIP_H_ */
