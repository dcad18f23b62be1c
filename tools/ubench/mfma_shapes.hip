// Microbenchmark: sustained chip rate of the two fp32 MFMA shapes of gfx950 -- v_mfma_f32_32x32x2_f32 (what gemm_f32.hip
// issues) and v_mfma_f32_16x16x4_f32 (what the vendor library's fp32 kernels issue) -- with nothing else in the loop.
// Same flops per cycle on paper (64 / clk / SIMD); the question is whether the chip holds the same clock under both
// (accumulator traffic per flop differs 2x).  One block per CU, 1 or 2 waves per SIMD, ~0.3 ms per launch, 10 launches.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_shapes.hip -o tools/bin/mfma_shapes && tools/bin/mfma_shapes
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE, int NACC>
__global__ void __launch_bounds__(512) k(float* out, int iters, unsigned seed) {
  const int tid = threadIdx.x;
  // operands with "random" mantissas (power depends on toggling): a few distinct registers, rotated
  float a[4], b[4];
  unsigned s = seed + tid * 2654435761u + blockIdx.x * 40503u;
  for (int j = 0; j < 4; ++j) {
    s = s * 1664525u + 1013904223u; a[j] = __uint_as_float(0x3f000000u | (s >> 9));
    s = s * 1664525u + 1013904223u; b[j] = __uint_as_float(0x3f000000u | (s >> 9)) - 0.75f;
  }
  float sum = 0.f;
  if (SHAPE == 32) {
    f32x16 acc[NACC];
    for (int q = 0; q < NACC; ++q) for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int m = 0; m < 16; ++m) acc[m % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m & 3], b[(m >> 2) & 3], acc[m % NACC], 0, 0, 0);
    }
    for (int q = 0; q < NACC; ++q) for (int r = 0; r < 16; ++r) sum += acc[q][r];
  } else {
    f32x4 acc[NACC];
    for (int q = 0; q < NACC; ++q) for (int r = 0; r < 4; ++r) acc[q][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int m = 0; m < 32; ++m) acc[m % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m & 3], b[(m >> 2) & 3], acc[m % NACC], 0, 0, 0);
    }
    for (int q = 0; q < NACC; ++q) for (int r = 0; r < 4; ++r) sum += acc[q][r];
  }
  out[blockIdx.x * blockDim.x + tid] = sum;
}

template <int SHAPE, int NACC>
void run(const char* name, int threads, float* out) {
  const int iters = 700;  // x 16 x 64 clk (or 32 x 32 clk) = 0.72 M clk per wave ~ 0.3 ms at one wave per SIMD
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9, sum = 0;
  const int reps = 10;
  for (int rep = 0; rep < reps + 2; ++rep) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<SHAPE, NACC>), dim3(256), dim3(threads), 0, 0, out, iters, 12345u + rep);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep >= 2) { sum += ms; best = ms < best ? ms : best; }
  }
  const double flops = 256.0 * (threads / 64) * iters * 16 * 4096.0;
  const double ms = sum / reps;
  printf("%-28s waves/SIMD %d: %7.1f us (best %7.1f)  %6.1f TF mean  %6.1f TF best  => clock %.0f MHz if the pipe never idles\n", name,
         threads / 256, ms * 1e3, best * 1e3, flops / (ms * 1e-3) / 1e12, flops / (best * 1e-3) / 1e12,
         flops / (ms * 1e-3) / (256.0 * 4 * 64) / 1e6);
}

int main() {
  float* out;
  hipMalloc(&out, 256 * 512 * 4);
  for (int round = 0; round < 2; ++round)
    for (int threads : {256, 512}) {
      run<32, 4>("32x32x2  4 accumulators", threads, out);
      run<16, 8>("16x16x4  8 accumulators", threads, out);
      run<32, 2>("32x32x2  2 accumulators", threads, out);
      run<16, 4>("16x16x4  4 accumulators", threads, out);
    }
  return 0;
}
