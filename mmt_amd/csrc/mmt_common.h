// Device-side helpers shared by the gfx950 kernels of the MMT hot path.
// CDNA4 only: 64-lane wavefronts, MFMA 16x16x32 bf16, LDS-DMA (global_load_lds), ds_read_b64_tr_b16.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MMT_WAVE 64

typedef unsigned short bf16_t;  // raw bf16 bits in memory
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(4))) unsigned short u16x4;
typedef __attribute__((ext_vector_type(8))) unsigned short u16x8;

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((unsigned)v) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {  // round-to-nearest-even (v_cvt_pk_bf16_f32)
  return __builtin_bit_cast(unsigned short, (__bf16)f);
}
__device__ __forceinline__ unsigned pack_bf2(float lo, float hi) {
  return (unsigned)f2bf(lo) | ((unsigned)f2bf(hi) << 16);
}

// ---- counter-based dropout RNG -------------------------------------------------------------
// One 32-bit hash of (seed-derived key, 64-bit element index >> 1) yields two 16-bit uniforms;
// element e keeps iff u16(e) >= thr16, thr16 = round(p * 65536).  The index is always expressed in
// ORIGINAL (sample, position, channel) coordinates so masks do not depend on row packing and the
// backward pass regenerates them instead of storing them.
__device__ __forceinline__ unsigned mix32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ unsigned rng_pair(unsigned key, unsigned long long pair_idx) {
  unsigned hi = (unsigned)(pair_idx >> 32), lo = (unsigned)pair_idx;
  return mix32(mix32(key ^ (hi * 0x9e3779b9U)) ^ lo);
}
__device__ __forceinline__ bool keep_elem(unsigned key, unsigned long long idx, unsigned thr16) {
  unsigned r = rng_pair(key, idx >> 1);
  unsigned u = (idx & 1) ? (r >> 16) : (r & 0xffffU);
  return u >= thr16;
}
// four consecutive elements starting at idx (idx % 4 == 0): two hashes
__device__ __forceinline__ void keep4(unsigned key, unsigned long long idx, unsigned thr16, bool k[4]) {
  unsigned r0 = rng_pair(key, idx >> 1), r1 = rng_pair(key, (idx >> 1) + 1);
  k[0] = (r0 & 0xffffU) >= thr16; k[1] = (r0 >> 16) >= thr16;
  k[2] = (r1 & 0xffffU) >= thr16; k[3] = (r1 >> 16) >= thr16;
}

// Per-step seed lives in device memory (kernel arguments are frozen under hipGraph replay): the
// effective stream key is a hash of the by-value site key and *seed_dev.
__device__ __forceinline__ unsigned eff_key(unsigned key, const unsigned* __restrict__ seed_dev) {
  return seed_dev ? mix32(key ^ mix32(*seed_dev + 0x632be5abU)) : key;
}

__device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_erf_grad_f(float x) {
  // d/dx [x * Phi(x)] = Phi(x) + x * phi(x)
  float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
  float pdf = 0.39894228040143268f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// XCD-aware bijective remap of a 1-D block id: blocks that land on the same XCD (id % 8) get a
// contiguous chunk of tile ids, so neighbouring tiles share operand panels in that XCD's L2.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, j = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
}
