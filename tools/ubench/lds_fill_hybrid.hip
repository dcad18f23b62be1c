// GPU micro-benchmark (round 3): can a CU fill LDS faster by using BOTH paths at once -- LDS-DMA (buffer_load_dwordx4 ... lds,
// ~37 B/clk/CU alone) and the register path (buffer_load_dwordx4 -> VGPR -> ds_write_b128, LDS store limit ~73 B/clk/CU)?  If the
// two limits are independent, the bf16 GEMM's 1024-frame tiles (bound by the fill) could move one operand through registers.
// Source resident in L2 (1 MiB re-read by every CU).  Blocks of 4 waves; BLOCKS_PER_CU resident blocks per CU via the LDS request.
//   hipcc --offload-arch=gfx950 -O3 -o lds_fill_hybrid tools/ubench/lds_fill_hybrid.hip && ./lds_fill_hybrid
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// per iteration a wave moves ND KiB by LDS-DMA and NR KiB through registers
template <int ND, int NR>
__global__ void __launch_bounds__(256) fill(const float* src, int iters, float* sink) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned long long a = (unsigned long long)src;
  i32x4 rsrc = {(int)(unsigned)a, (int)((unsigned)(a >> 32) & 0xffffu), 1 << 20, 0x00020000};
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)smem);
  const int voff = lane * 16 + wave * 4096;
  // (inline asm throughout: the compiler must neither drop the LDS stores nor hoist the loads)
  f32x4 reg[NR > 0 ? NR : 1];
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < ND; ++j) {
      const unsigned dst = lds0 + (unsigned)(((it & 3) * 16 + wave * 4 + (j & 3)) * 1024);
      const int soff = ((it * ND + j) & 15) * 1024 * 16;
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                   : : "s"(dst), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
    }
#pragma unroll
    for (int j = 0; j < NR; ++j) {
      const int soff = 512 * 1024 + ((it * NR + j) & 15) * 16384;
      asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(reg[j]) : "v"(voff), "s"(rsrc), "s"(soff) : "memory");
    }
    // the register loads of this iteration are the youngest NR vector-memory operations: wait for them only
    if (NR > 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int j = 0; j < NR; ++j) {
        const unsigned q = lds0 + 65536u + (unsigned)((((it & 1) * 16 + wave * 4 + (j & 3)) * 256 + lane * 4) * 4);
        asm volatile("ds_write_b128 %0, %1" : : "v"(q), "v"(reg[j]) : "memory");
      }
    } else {
      asm volatile("s_waitcnt vmcnt(%0)" : : "n"(8) : "memory");
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
  if (sink) sink[blockIdx.x * 256 + threadIdx.x] = smem[threadIdx.x] + smem[16384 + threadIdx.x];
}

template <int ND, int NR>
void run(const float* src, float* sink, int iters, int blocks_per_cu) {
  const int lds = blocks_per_cu == 1 ? 128 * 1024 : 72 * 1024;  // ring images: 64 KiB DMA + up to 8 KiB register image (+ residency)
  hipFuncSetAttribute((const void*)fill<ND, NR>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((fill<ND, NR>), dim3(256 * blocks_per_cu), dim3(256), lds, 0, src, iters, sink);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (rep > 0 && ms < best) best = ms;
  }
  const double bytes_per_cu = (double)blocks_per_cu * 4 * (ND + NR) * 1024.0 * iters;
  printf("DMA %d KiB + registers %d KiB per wave-iteration, %d block(s)/CU: %8.3f ms  %6.1f B/clk/CU at 2.1 GHz  (DMA %5.1f + reg %5.1f)\n", ND, NR,
         blocks_per_cu, best, bytes_per_cu / (best * 1e-3 * 2.1e9), bytes_per_cu * ND / (ND + NR) / (best * 1e-3 * 2.1e9),
         bytes_per_cu * NR / (ND + NR) / (best * 1e-3 * 2.1e9));
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 4096;
  float *src, *sink;
  hipMalloc(&src, 1 << 20);
  hipMemset(src, 0, 1 << 20);
  hipMalloc(&sink, 1024 * 256 * 4);
  for (int bpc = 1; bpc <= 2; ++bpc) {
    run<4, 0>(src, sink, iters, bpc);
    run<0, 4>(src, sink, iters, bpc);
    run<0, 2>(src, sink, iters, bpc);
    run<4, 2>(src, sink, iters, bpc);
    run<4, 4>(src, sink, iters, bpc);
    run<2, 2>(src, sink, iters, bpc);
    run<2, 4>(src, sink, iters, bpc);
  }
  return 0;
}
