// GPU micro-benchmark: how fast can a CU fill LDS -- by LDS-DMA (buffer_load_dwordx4 ... lds, 1 KiB per wave
// instruction, source resident in L2) and by ds_write_b128 from registers (the register-staged path)?
// Two blocks of 4 waves per CU, as the 64x64 GEMM tiles run.  usage: lds_fill [iters]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#ifndef DEPTH
#define DEPTH 8  // pieces a wave keeps in flight beyond the 4 it just issued
#endif
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) dma_fill(const float* src, int iters, float* sink) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned long long a = (unsigned long long)src;
  i32x4 rsrc = {(int)(unsigned)a, (int)((unsigned)(a >> 32) & 0xffffu), 1 << 20, 0x00020000};
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)smem);
  const int voff = lane * 16 + wave * 4096;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {  // 4 pieces = this wave's share of a 16 KiB tile step
      const unsigned dst = lds0 + (unsigned)(((it & 3) * 16 + wave * 4 + j) * 1024);
      const int soff = ((it * 4 + j) & 15) * 1024 * 16;
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                   : : "s"(dst), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(%0)" : : "n"(DEPTH) : "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
  if (sink) sink[blockIdx.x * 256 + threadIdx.x] = smem[threadIdx.x];
}

__global__ void __launch_bounds__(256) dsw_fill(const float* src, int iters, float* sink) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  f32x4 v = *reinterpret_cast<const f32x4*>(src + threadIdx.x * 4);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float* q = smem + (((it & 3) * 16 + (threadIdx.x >> 6) * 4 + j) * 256 + (threadIdx.x & 63) * 4);
      asm volatile("ds_write_b128 %0, %1" : : "v"((unsigned)(unsigned long long)q), "v"(v) : "memory");
      v.x += 1.f;
    }
  }
  __syncthreads();
  if (sink) sink[blockIdx.x * 256 + threadIdx.x] = smem[threadIdx.x];
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 4096;
  float *src, *sink;
  hipMalloc(&src, 1 << 20); hipMemset(src, 0, 1 << 20);
  hipMalloc(&sink, 512 * 256 * 4);
  hipFuncSetAttribute((const void*)dma_fill, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipFuncSetAttribute((const void*)dsw_fill, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int k = 0; k < 2; ++k) {
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0, 0);
      if (k == 0) hipLaunchKernelGGL(dma_fill, dim3(512), dim3(256), 65536, 0, src, iters, sink);
      else hipLaunchKernelGGL(dsw_fill, dim3(512), dim3(256), 65536, 0, src, iters, sink);
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double bytes_per_cu = 2.0 * 4 * 4 * 1024.0 * iters;  // 2 blocks x 4 waves x 4 KiB per iteration
      if (rep == 2)
        printf("%s: %8.3f ms  %6.1f B/clk/CU at 2.3 GHz  (%.1f TB/s chip)\n", k == 0 ? "LDS-DMA (L2 source)" : "ds_write_b128      ",
               ms, bytes_per_cu / (ms * 1e-3 * 2.3e9), bytes_per_cu * 256 / (ms * 1e-3) / 1e12);
    }
  }
  return 0;
}
