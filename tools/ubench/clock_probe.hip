// GPU tool: shader clock while the fp32 GEMM runs.  A one-wave probe kernel on a second stream samples
// s_memtime (shader cycles) against the constant 100 MHz wall clock while the GEMM loops on the first stream.
// Build with -DTFK_ABL=<mask> to probe the ablated loops as well.
#include "../../tfkaldi_amd/csrc/gemm_f32.hip"
#include <vector>
__global__ void probe(long long* out, int n, long long wall_ticks_per_sample) {
  if (threadIdx.x != 0) return;
  for (int i = 0; i < n; ++i) {
    const long long w0 = wall_clock64();
    const long long c0 = clock64();
    while (wall_clock64() - w0 < wall_ticks_per_sample) {}
    const long long c1 = clock64();
    const long long w1 = wall_clock64();
    out[2 * i] = c1 - c0;
    out[2 * i + 1] = w1 - w0;
  }
}
int main(int argc, char** argv) {
  int layout = argc > 1 ? atoi(argv[1]) : 0, M = 1024, N = 2048, K = 2048, cfg = argc > 2 ? atoi(argv[2]) : 3;
  if (layout == 2) { M = 2048; K = 1024; }
  size_t na = (size_t)4096 * 4100, nb = na;
  float *a, *b, *c;
  hipMalloc(&a, na * 4); hipMalloc(&b, nb * 4); hipMalloc(&c, na * 4);
  std::vector<float> h(na);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u >> 8) & 0xffff) / 65536.f - 0.5f;
  hipMemcpy(a, h.data(), na * 4, hipMemcpyHostToDevice); hipMemcpy(b, h.data(), nb * 4, hipMemcpyHostToDevice);
  tfk::GemmArgs g = {};
  g.A = a; g.B = b; g.C = c; g.M = M; g.N = N; g.K = K; g.epi = 0;
  g.lda = layout == 2 ? M : K; g.ldb = layout == 1 ? K : N; g.ldc = N;
  hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2);
  const int ns = 40;
  long long* d; hipMalloc(&d, ns * 16);
  std::vector<long long> r(2 * ns);
  for (int phase = 0; phase < 2; ++phase) {  // 0: idle GPU, 1: GEMM running
    if (phase == 1) for (int i = 0; i < 400; ++i) tfk::gemm_f32((tfk::GemmLayout)layout, g, cfg, s1);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, s2, d, ns, 50000LL);  // 0.5 ms per sample
    hipDeviceSynchronize();
    hipMemcpy(r.data(), d, ns * 16, hipMemcpyDeviceToHost);
    double lo = 1e9, hi = 0, sum = 0;
    for (int i = 5; i < ns; ++i) { double f = (double)r[2 * i] / (double)r[2 * i + 1] * 100.0; lo = f < lo ? f : lo; hi = f > hi ? f : hi; sum += f; }
    printf("ABL=%2d layout %d cfg %d %s: shader clock %.0f MHz (min %.0f max %.0f)\n", TFK_ABL, layout, cfg,
           phase ? "GEMM running" : "idle        ", sum / (ns - 5), lo, hi);
  }
  return 0;
}
