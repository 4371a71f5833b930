// Lab: how fast does ONE CU pull GEMM-operand-shaped data out of L2 / MALL, as a function of how it asks?
// Every GEMM of this library saturates at ~21 B/clk/CU of L2 -> LDS ingest.  This program streams a
// 256-row band of a row-major bf16 matrix per block (one block of 8 waves per CU, 8 blocks share a band like the tiles of a
// tile row), K-step by K-step, with
//   SEG  = contiguous bytes per row and request (128 = a 64-deep K-step of one row, the GEMMs' shape; 256; 512; 1024)
//   MODE = 0: buffer_load_dwordx4 ... lds (LDS-DMA)   1: global_load_dwordx4 into VGPRs   2: global_load_lds_dwordx4
//   DEPTH = 1-KiB pieces a wave keeps in flight
// and prints bytes / shader cycle / CU.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/ingest_rate tools/ubench/ingest_rate.hip && /tmp/ingest_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

template <int SEG, int MODE, int DEPTH, int IW = 8>
__global__ __launch_bounds__(512) void ingest(const unsigned short* __restrict__ A, int64_t ld, int rows, int ksteps, long long* out,
                                              unsigned* sink) {
  extern __shared__ __attribute__((aligned(256))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int LPR = SEG / 16;        // lanes per row segment
  constexpr int RPP = 64 / LPR;        // rows per 1-KiB piece
  constexpr int KCH = SEG / 128;       // 64-deep K-steps one piece row covers
  // the 8 blocks that share a band sit on ONE XCD (block b runs on XCD b % 8), as the tiles of a tile row do after xcd_remap
  const int band = (((int)blockIdx.x % 8) * 4 + ((int)blockIdx.x / 8) / 8) % (rows / 256);
  // per K-step the block moves 256 rows x 128 B = 32 KiB = 32 pieces = 4 per wave; a wave's piece p of "super-step" s (KCH K-steps)
  // covers rows 32 wave + RPP * (p % (32 / RPP)) .. and column chunk p / (32 / RPP)
  const unsigned short* base = A + (int64_t)band * 256 * ld;
  constexpr int PPS = 4 * KCH * (8 / IW);  // pieces per ISSUING wave and super-step (IW of the 8 waves issue everything)
  unsigned acc = 0;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000);
  __syncthreads();
  const long long t0 = clock64();
  if (wave < IW) {
  int inflight = 0;
  for (int s = 0; s < ksteps / KCH; ++s) {
#pragma unroll
    for (int p = 0; p < PPS; ++p) {
      const int row = (wave * (256 / IW) + p * RPP + lane / LPR) & 255;
      const int64_t off = (int64_t)row * ld * 2 + (int64_t)s * SEG + (lane % LPR) * 16;  // bytes
      if (MODE == 0) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(smem + ((wave * PPS + p) & 63) * 1024), 16, (int)off, 0, 0, 0);
      } else if (MODE == 2) {
        __builtin_amdgcn_global_load_lds(GLB_PTR((const unsigned char*)base + off), LDS_PTR(smem + ((wave * PPS + p) & 63) * 1024), 16, 0, 0);
      } else {
        const u32x4 v = *(const u32x4*)((const unsigned char*)base + off);
        acc ^= v[0] ^ v[3];
      }
      if (MODE != 1 && ++inflight >= DEPTH) {
        if (DEPTH >= 16) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
        else if (DEPTH >= 8) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
        else if (DEPTH >= 4) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  const long long t1 = clock64();
  if (tid == 0) out[blockIdx.x] = t1 - t0;
  if (acc == 0x12345u) sink[0] = acc;
}

template <int SEG, int MODE, int DEPTH, int IW = 8>
static void run(const unsigned short* A, int64_t ld, int rows, int ksteps, long long* dout, unsigned* sink, const char* name) {
  hipFuncSetAttribute((const void*)ingest<SEG, MODE, DEPTH, IW>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  std::vector<long long> h(256);
  double best = 1e30;
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL((ingest<SEG, MODE, DEPTH, IW>), dim3(256), dim3(512), 65536, 0, A, ld, rows, ksteps, dout, sink);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), dout, 256 * sizeof(long long), hipMemcpyDeviceToHost);
    double m = 0;
    for (int i = 0; i < 256; ++i) m += (double)h[i];
    m /= 256;
    if (m < best) best = m;
  }
  const double bytes = (double)ksteps * 32768.0;
  printf("%-44s seg %4d B  depth %d  issuing waves %d: %8.0f cycles per block, %5.1f B/clk/CU\n", name, SEG, DEPTH, IW, best, bytes / best);
}

int main() {
  const int rows = 8192, ksteps = 112;  // K = 7168
  const int64_t ld = 7168;
  unsigned short* A;
  long long* dout;
  unsigned* sink;
  hipMalloc(&A, (size_t)rows * ld * 2);
  hipMemset(A, 1, (size_t)rows * ld * 2);
  hipMalloc(&dout, 256 * sizeof(long long));
  hipMalloc(&sink, 4);
  run<128, 0, 8>(A, ld, rows, ksteps, dout, sink, "buffer_load lds");
  run<128, 0, 4>(A, ld, rows, ksteps, dout, sink, "buffer_load lds");
  run<256, 0, 8>(A, ld, rows, ksteps, dout, sink, "buffer_load lds");
  run<512, 0, 8>(A, ld, rows, ksteps, dout, sink, "buffer_load lds");
  run<1024, 0, 8>(A, ld, rows, ksteps, dout, sink, "buffer_load lds");
  run<128, 0, 8, 4>(A, ld, rows, ksteps, dout, sink, "buffer_load lds");
  run<128, 0, 8, 2>(A, ld, rows, ksteps, dout, sink, "buffer_load lds");
  run<128, 0, 16, 2>(A, ld, rows, ksteps, dout, sink, "buffer_load lds");
  run<128, 0, 16, 1>(A, ld, rows, ksteps, dout, sink, "buffer_load lds");
  run<128, 0, 16, 4>(A, ld, rows, ksteps, dout, sink, "buffer_load lds");
  run<128, 0, 8, 1>(A, ld, rows, ksteps, dout, sink, "buffer_load lds");
  run<128, 2, 8>(A, ld, rows, ksteps, dout, sink, "global_load_lds");
  run<256, 2, 8>(A, ld, rows, ksteps, dout, sink, "global_load_lds");
  run<1024, 2, 8>(A, ld, rows, ksteps, dout, sink, "global_load_lds");
  run<128, 1, 8>(A, ld, rows, ksteps, dout, sink, "global_load_dwordx4 -> VGPR (compiler waits)");
  run<256, 1, 8>(A, ld, rows, ksteps, dout, sink, "global_load_dwordx4 -> VGPR (compiler waits)");
  run<1024, 1, 8>(A, ld, rows, ksteps, dout, sink, "global_load_dwordx4 -> VGPR (compiler waits)");
  return 0;
}
