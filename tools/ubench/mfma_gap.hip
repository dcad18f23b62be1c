// Microbenchmark: marginal cost of LDS / VMEM instructions issued in the gaps of an fp32 MFMA stream.
// One block of 256 threads per CU (1 wave per SIMD) or 512 (2 waves per SIMD); each wave runs ITER iterations
// of 16 x v_mfma_f32_32x32x2_f32 over NACC accumulators, with NW ds_write_b128 / NR ds_read_b128 / NL global
// dwordx4 loads spread one per MFMA gap.  Prints cycles per iteration (1024 = pure MFMA pace).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, int NW, int NR, int NL, int WIDTH>
__global__ void __launch_bounds__(512) k(const float* __restrict__ g, float* out, int iters, int stride) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x;
  f32x16 acc[NACC];
  for (int q = 0; q < NACC; ++q) for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
  float4 w = make_float4(tid, 1.f, 2.f, 3.f);
  float4 rd[NR > 0 ? NR : 1];
  float4 ld[NL > 0 ? NL : 1];
  for (int j = 0; j < (NL > 0 ? NL : 1); ++j) ld[j] = make_float4(0, 0, 0, 0);
  float a = 1.0f + tid * 1e-3f, b = 0.5f;
  const float* gp = g + (size_t)blockIdx.x * 4096 + tid * 4;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      acc[m % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[m % NACC], 0, 0, 0);
      if (m < NW) {
        if (WIDTH == 16) *reinterpret_cast<float4*>(lds + ((tid + m * 512) & 8191) * 4) = w;
        else if (WIDTH == 8) { *reinterpret_cast<float2*>(lds + ((tid + m * 512) & 8191) * 4) = make_float2(w.x, w.y); }
        else { lds[((tid + m * 512) & 8191) * 4] = w.x; }
      }
      if (m >= 4 && m - 4 < NR) rd[m - 4] = *reinterpret_cast<float4*>(lds + ((tid * 9 + (m - 4) * 64) & 8191) * 4);
      if (m >= 8 && m - 8 < NL) ld[m - 8] = *reinterpret_cast<const float4*>(gp + (size_t)((it * NL + (m - 8)) % 64) * stride);
      __builtin_amdgcn_sched_barrier(0);
    }
    for (int j = 0; j < NR; ++j) w.x += rd[j].x;
    for (int j = 0; j < NL; ++j) w.y += ld[j].y;
  }
  float s = w.x + w.y;
  for (int q = 0; q < NACC; ++q) for (int r = 0; r < 16; ++r) s += acc[q][r];
  out[blockIdx.x * blockDim.x + tid] = s;
}

template <int NACC, int NW, int NR, int NL, int WIDTH>
void run(const char* name, int threads, const float* g, float* out) {
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipFuncSetAttribute((const void*)k<NACC, NW, NR, NL, WIDTH>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<NACC, NW, NR, NL, WIDTH>), dim3(256), dim3(threads), 140 * 1024, 0, g, out, iters, 1 << 20);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
  }
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double wps = threads / 256.0;  // waves per SIMD
  printf("%-34s threads %3d: %8.1f us  %7.1f cyc/iter/wave-slot @2.29GHz (MFMA pace %4.0f)  %5.1f TF\n", name, threads,
         ms * 1e3, ms * 1e-3 * 2.29e9 / iters, 1024.0 * wps, 256.0 * (threads / 64) * iters * 16 * 4096.0 / (ms * 1e-3) / 1e12);
}

int main() {
  float *g, *out;
  hipMalloc(&g, (size_t)64 << 22); hipMemset(g, 0, (size_t)64 << 22);
  hipMalloc(&out, 256 * 512 * 4);
  for (int threads : {256, 512}) {
    run<4, 0, 0, 0, 16>("4acc mfma only", threads, g, out);
    run<1, 0, 0, 0, 16>("1acc mfma only", threads, g, out);
    run<4, 4, 0, 0, 16>("4acc + 4 ds_write_b128", threads, g, out);
    run<1, 4, 0, 0, 16>("1acc + 4 ds_write_b128", threads, g, out);
    run<4, 8, 0, 0, 16>("4acc + 8 ds_write_b128", threads, g, out);
    run<4, 4, 0, 0, 8>("4acc + 4 ds_write_b64", threads, g, out);
    run<4, 4, 0, 0, 4>("4acc + 4 ds_write_b32", threads, g, out);
    run<4, 0, 4, 0, 16>("4acc + 4 ds_read_b128", threads, g, out);
    run<1, 0, 4, 0, 16>("1acc + 4 ds_read_b128", threads, g, out);
    run<4, 0, 8, 0, 16>("4acc + 8 ds_read_b128", threads, g, out);
    run<4, 0, 0, 4, 16>("4acc + 4 global_load_dwordx4", threads, g, out);
    run<1, 0, 0, 4, 16>("1acc + 4 global_load_dwordx4", threads, g, out);
    run<4, 4, 4, 4, 16>("4acc + 4w + 4r + 4l", threads, g, out);
    run<1, 4, 4, 4, 16>("1acc + 4w + 4r + 4l", threads, g, out);
  }
  return 0;
}
