// Microbenchmark: what does a bf16 MFMA cost in ENERGY?  The socket runs into its power limit under dense bf16 matrix work
// (profiles/r05_gemm_*_power_smi.txt), so the sustained rate of a loop of nothing but MFMAs is a measure of joules per flop:
// shapes 32x32x16 and 16x16x32, operands with random significands / half of the A (or B) elements zero / all zeros.  Every
// variant runs back to back for ~3 s (the power controller needs that long to settle); the rate of the last 2 s is printed.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_bf16_energy.hip -o tools/bin/mfma_bf16_energy && tools/bin/mfma_bf16_energy
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// MODE: 0 random A and B, 1 half of A's elements zero, 2 half of B's elements zero, 3 zeros, 4 A = one plane pattern repeated
// (the same A register for every MFMA: an operand that stays put)
template <int SHAPE, int MODE, int BUBBLE = 0>
__global__ void __launch_bounds__(512) k(float* out, int iters, unsigned seed) {
  const int tid = threadIdx.x;
  bf16x8 a[8], b[8];
  unsigned s = seed + tid * 2654435761u + blockIdx.x * 40503u;
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      s = s * 1664525u + 1013904223u;
      float x = __uint_as_float(0x3f000000u | (s >> 9)) - 0.75f;  // [-0.25, 0.25): random sign and significand
      s = s * 1664525u + 1013904223u;
      float y = __uint_as_float(0x3f000000u | (s >> 9)) - 0.75f;
      const bool za = MODE == 3 || (MODE == 1 && (s >> 13 & 1)), zb = MODE == 3 || (MODE == 2 && (s >> 14 & 1));
      a[j][e] = (__bf16)(za ? 0.f : x);
      b[j][e] = (__bf16)(zb ? 0.f : y);
    }
  float sum = 0.f;
  if (SHAPE == 32) {
    f32x16 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int m = 0; m < 16; ++m) {
        acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[MODE == 4 ? 0 : (m & 7)], b[(m * 3 + 1) & 7], acc[m & 3], 0, 0, 0);
        // BUBBLE x 8 idle issue cycles behind every MFMA (two waves per SIMD take turns: the pipe idles when both sit in a bubble)
        if (BUBBLE >= 1) asm volatile("s_nop 7" ::: "memory");
        if (BUBBLE >= 2) asm volatile("s_nop 7" ::: "memory");
        if (BUBBLE >= 4) asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
        if (BUBBLE >= 8) asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) sum += acc[q][r];
  } else {
    f32x4 acc[8];
#pragma unroll
    for (int q = 0; q < 8; ++q)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[q][r] = 0.f;
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int m = 0; m < 32; ++m)
        acc[m & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[MODE == 4 ? 0 : (m & 7)], b[(m * 3 + 1) & 7], acc[m & 7], 0, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < 8; ++q)
#pragma unroll
      for (int r = 0; r < 4; ++r) sum += acc[q][r];
  }
  out[blockIdx.x * blockDim.x + tid] = sum;
}

template <int SHAPE, int MODE, int BUBBLE = 0>
void run(const char* name, float* out) {
  const int threads = 512;
  const int iters = 2000;  // x 16 MFMAs of 32768 flop (or 32 of 16384) per wave
  const double flops_per_launch = 256.0 * (threads / 64) * iters * 16 * 32768.0;
  using clk = std::chrono::steady_clock;
  const auto t0 = clk::now();
  long launches = 0, late = 0;
  clk::time_point t_mid;
  bool mid = false;
  while (true) {
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((k<SHAPE, MODE, BUBBLE>), dim3(256), dim3(threads), 0, 0, out, iters, 12345u + (unsigned)launches + i);
    (void)hipDeviceSynchronize();
    launches += 20;
    const double t = std::chrono::duration<double>(clk::now() - t0).count();
    if (!mid && t > 1.0) { mid = true; t_mid = clk::now(); late = launches; }
    if (t > 3.0) break;
  }
  const double dt = std::chrono::duration<double>(clk::now() - t_mid).count();
  const double tf = (launches - late) * flops_per_launch / dt / 1e12;
  printf("%-52s %7.1f TFLOP/s sustained = %.2f of 2500\n", name, tf, tf / 2500.0);
}

int main() {
  float* out;
  (void)hipMalloc(&out, 256 * 512 * sizeof(float));
  run<32, 0>("32x32x16  random A, random B", out);
  run<16, 0>("16x16x32  random A, random B", out);
  run<32, 1>("32x32x16  half of A zero", out);
  run<32, 2>("32x32x16  half of B zero", out);
  run<32, 4>("32x32x16  A stays in place (one register), random B", out);
  run<16, 4>("16x16x32  A stays in place (one register), random B", out);
  // is the limit POWER (the clock would rise when the pipe idles part of the time) or a clock cap that comes with matrix work?
  // bubbles behind every MFMA lower the pipe's duty; the zeros line of a pair gives that duty (x 0.98), the random line the clock
  run<32, 0, 4>("32x32x16  random, 32 idle issue cycles behind every MFMA", out);
  run<32, 3, 4>("32x32x16  zeros,  32 idle issue cycles behind every MFMA", out);
  run<32, 0, 8>("32x32x16  random, 64 idle issue cycles behind every MFMA", out);
  run<32, 3, 8>("32x32x16  zeros,  64 idle issue cycles behind every MFMA", out);
  run<32, 3>("32x32x16  zeros", out);
  run<16, 3>("16x16x32  zeros", out);
  return 0;
}
