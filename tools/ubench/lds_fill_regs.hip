// GPU micro-benchmark (round 3): the REGISTER fill path with deep prefetch -- every wave keeps D KiB of buffer_load_dwordx4 in flight
// (a ring of D register quads), stores the oldest to LDS with ds_write_b128 and re-issues.  tools/ubench/lds_fill_hybrid.hip showed
// LDS-DMA capped at ~36-42 B/clk/CU while plain loads reached 66 B/clk/CU although that version waited for every load at once.
// Source resident in L2 (1 MiB).  Usage: lds_fill_regs [iters]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int D, bool STORE>
__global__ void __launch_bounds__(256) fill(const float* src, int iters, float* sink) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned long long a = (unsigned long long)src;
  i32x4 rsrc = {(int)(unsigned)a, (int)((unsigned)(a >> 32) & 0xffffu), 1 << 20, 0x00020000};
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)smem);
  const int voff = lane * 16 + wave * 4096;
  f32x4 reg[D];
#pragma unroll
  for (int j = 0; j < D; ++j) {
    const int soff = (j & 15) * 16384;
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(reg[j]) : "v"(voff), "s"(rsrc), "s"(soff) : "memory");
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < D; ++j) {
      asm volatile("s_waitcnt vmcnt(%0)" : : "n"(D - 1) : "memory");  // the oldest load has arrived
      if (STORE) {
        const unsigned q = lds0 + (unsigned)((((it & 3) * 16 + wave * 4 + (j & 3)) * 256 + lane * 4) * 4);
        asm volatile("ds_write_b128 %0, %1" : : "v"(q), "v"(reg[j]) : "memory");
      } else {
        asm volatile("" : : "v"(reg[j]));
      }
      const int soff = (((it + 1) * D + j) & 15) * 16384 + ((it & 3) << 8);
      asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(reg[j]) : "v"(voff), "s"(rsrc), "s"(soff) : "memory");
    }
  }
  asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  if (sink) sink[blockIdx.x * 256 + threadIdx.x] = smem[threadIdx.x] + reg[0].x;
}

template <int D, bool STORE>
void run(const float* src, float* sink, int iters, int blocks_per_cu) {
  const int lds = blocks_per_cu == 1 ? 128 * 1024 : blocks_per_cu == 2 ? 72 * 1024 : 36 * 1024;
  hipFuncSetAttribute((const void*)fill<D, STORE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL((fill<D, STORE>), dim3(256 * blocks_per_cu), dim3(256), lds, 0, src, iters, sink);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (rep > 0 && ms < best) best = ms;
  }
  const double bytes_per_cu = (double)blocks_per_cu * 4 * D * 1024.0 * iters;
  printf("%d KiB in flight per wave, %d block(s) of 4 waves per CU (%3d KiB in flight per CU), %s: %8.3f ms  %6.1f B/clk/CU at 2.1 GHz\n", D,
         blocks_per_cu, 4 * D * blocks_per_cu, STORE ? "load + ds_write_b128" : "load only          ", best,
         bytes_per_cu / (best * 1e-3 * 2.1e9));
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 1024;
  float *src, *sink;
  (void)hipMalloc(&src, 1 << 20);
  (void)hipMemset(src, 0, 1 << 20);
  (void)hipMalloc(&sink, 1024 * 256 * 4);
  for (int bpc = 1; bpc <= 4; bpc *= 2) {
    run<4, true>(src, sink, iters, bpc);
    run<8, true>(src, sink, iters, bpc);
    run<16, true>(src, sink, iters, bpc);
    run<8, false>(src, sink, iters, bpc);
    run<16, false>(src, sink, iters, bpc);
  }
  return 0;
}
