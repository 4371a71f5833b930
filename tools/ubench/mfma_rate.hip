// Lab: issue rate of the two bf16 MFMA shapes on gfx950, one wave per SIMD (4 waves per block, one block per CU) and two.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_rate tools/ubench/mfma_rate.hip && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// the weight-gradient kernel's pattern: 32 accumulators, 4 A fragments x 8 B fragments, every MFMA a different operand pair
__global__ __launch_bounds__(512) void kw(long long* out, float* sink, int iters) {
  bf16x8_t a[4], b[8];
  for (int i = 0; i < 4; ++i) for (int e = 0; e < 8; ++e) a[i][e] = (__bf16)(float)(threadIdx.x + e + i);
  for (int j = 0; j < 8; ++j) for (int e = 0; e < 8; ++e) b[j][e] = (__bf16)(float)(e + j);
  f32x4 c[4][8];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) c[i][j] = (f32x4){0, 0, 0, 0};
  __syncthreads();
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) c[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], c[i][j], 0, 0, 0);
    asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]));
  }
  float s = 0;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) s += c[i][j][0];
  const long long t1 = clock64();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  if (s == 123.456f) sink[0] = s;
}

// Does a partner wave's LDS-DMA stream slow the MFMAs of the computing wave on the same SIMD?  8 waves: waves 0-3 (one
// per SIMD) run the weight-gradient MFMA pattern; waves 4-7 (their SIMD partners) issue `loads` global_load_lds
// instructions (1 KiB each, L2-resident source) per 32 MFMAs of the partner -- or plain VALU work (mode 2), or nothing.
__global__ __launch_bounds__(512) void kmix(long long* out, float* sink, const unsigned short* src, int iters, int loads, int mode) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (wave >= 4) {
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
      if (mode == 1) {
        for (int l = 0; l < loads; ++l) {
          const unsigned short* g = src + ((size_t)((blockIdx.x * 4 + (wave - 4)) * 64 + ((it * loads + l) & 63)) * 64 + lane) * 8;
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                           (__attribute__((address_space(3))) void*)(lds + ((wave - 4) * 8 + (l & 7)) * 1024), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      } else if (mode == 2) {
        for (int l = 0; l < loads * 8; ++l) acc = acc * 1.0001f + (float)l;
        asm volatile("" : "+v"(acc));
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc == 123.f) sink[1] = acc;
    return;
  }
  bf16x8_t a[4], b[8];
  for (int i = 0; i < 4; ++i) for (int e = 0; e < 8; ++e) a[i][e] = (__bf16)(float)(threadIdx.x + e + i);
  for (int j = 0; j < 8; ++j) for (int e = 0; e < 8; ++e) b[j][e] = (__bf16)(float)(e + j);
  f32x4 c[4][8];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) c[i][j] = (f32x4){0, 0, 0, 0};
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) c[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], c[i][j], 0, 0, 0);
    asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]));
  }
  float s = 0;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) s += c[i][j][0];
  const long long t1 = clock64();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  if (s == 123.456f) sink[0] = s;
}

template <int SHAPE>
__global__ void k(long long* out, float* sink, int iters) {
  bf16x8_t a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(threadIdx.x + e); b[e] = (__bf16)(float)(e + 1); }
  f32x4 c4[16];
  f32x16 c16[4];
  for (int i = 0; i < 16; ++i) c4[i] = (f32x4){0, 0, 0, 0};
  for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) c16[i][e] = 0;
  __syncthreads();
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    if (SHAPE == 16) {
#pragma unroll
      for (int i = 0; i < 16; ++i) c4[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c4[i], 0, 0, 0);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) c16[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c16[i], 0, 0, 0);
    }
  }
  float s = 0;
  for (int i = 0; i < 16; ++i) s += c4[i][0];
  for (int i = 0; i < 4; ++i) s += c16[i][0];
  const long long t1 = clock64();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  if (s == 123.456f) sink[0] = s;
}

int main() {
  long long* out; float* sink;
  hipMalloc(&out, 1024 * 8); hipMalloc(&sink, 4);
  const int iters = 2000;
  for (int threads : {256, 512}) {
    for (int shape : {16, 32}) {
      for (int rep = 0; rep < 2; ++rep) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        if (shape == 16) hipLaunchKernelGGL(k<16>, dim3(256), dim3(threads), 0, 0, out, sink, iters);
        else hipLaunchKernelGGL(k<32>, dim3(256), dim3(threads), 0, 0, out, sink, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long h[256]; hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
        double avg = 0; for (int i = 0; i < 256; ++i) avg += h[i]; avg /= 256;
        const int per_it = shape == 16 ? 16 : 4;
        const double flops = 256.0 * (threads / 64) * iters * per_it * (shape == 16 ? 16.0 * 16 * 32 * 2 : 32.0 * 32 * 16 * 2);
        if (rep) printf("%d waves/CU, mfma %s: %.1f clock64 ticks per instruction per wave, kernel %.3f ms, %.0f TFLOP/s\n", threads / 64,
                        shape == 16 ? "16x16x32" : "32x32x16", avg / iters / per_it, ms, flops / ms / 1e9);
      }
    }
  }
  unsigned short* src; hipMalloc(&src, (size_t)256 * 4 * 64 * 64 * 16); hipMemset(src, 0, (size_t)256 * 4 * 64 * 64 * 16);
  hipFuncSetAttribute((const void*)kmix, hipFuncAttributeMaxDynamicSharedMemorySize, 32768);
  for (int mode : {0, 1, 2}) {
    for (int loads : {8, 16}) {
      for (int rep = 0; rep < 2; ++rep) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(kmix, dim3(256), dim3(512), 32768, 0, out, sink, src, iters, loads, mode);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long h[256]; hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
        double avg = 0; for (int i = 0; i < 256; ++i) avg += h[i]; avg /= 256;
        if (rep) printf("partner mode %d (%s), %2d per 32 MFMAs: %.1f ticks per MFMA for the computing wave, kernel %.3f ms\n", mode,
                        mode == 0 ? "idle" : mode == 1 ? "global_load_lds 1 KiB" : "8 VALU ops per unit", loads, avg / iters / 32, ms);
      }
      if (mode == 0) break;
    }
  }
  for (int threads : {256, 512}) {
    for (int rep = 0; rep < 2; ++rep) {
      hipLaunchKernelGGL(kw, dim3(256), dim3(threads), 0, 0, out, sink, iters);
      hipDeviceSynchronize();
      long long h[256]; hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
      double avg = 0; for (int i = 0; i < 256; ++i) avg += h[i]; avg /= 256;
      if (rep) printf("%d waves/CU, wgrad pattern (32 accumulators, 4 x 8 fragments): %.1f ticks per MFMA per wave\n", threads / 64, avg / iters / 32);
    }
  }
  return 0;
}
