// GPU tool: time the fp32 MFMA GEMM (one layout/config/shape) from a plain HIP host program.
// Built with -DTFK_ABL=<mask> it times the ablated variants of the K loop (see gemm_f32.hip).
#include "../tfkaldi_amd/csrc/gemm_f32.hip"
#include <vector>
int main(int argc, char** argv) {
  int layout = argc > 1 ? atoi(argv[1]) : 0, M = argc > 2 ? atoi(argv[2]) : 1024, N = argc > 3 ? atoi(argv[3]) : 2048,
      K = argc > 4 ? atoi(argv[4]) : 2048, cfg = argc > 5 ? atoi(argv[5]) : 3, iters = argc > 6 ? atoi(argv[6]) : 50;
  const int pad = getenv("GEMM_PAD") ? atoi(getenv("GEMM_PAD")) : 0;  // extra floats per row (leading-dimension padding)
  size_t na = (size_t)(layout == 2 ? K : M) * ((layout == 2 ? M : K) + 3 + pad), nb = (size_t)(layout == 1 ? N : K) * ((layout == 1 ? K : N) + 3 + pad);
  float *a, *b, *c;
  hipMalloc(&a, na * 4); hipMalloc(&b, nb * 4); hipMalloc(&c, (size_t)M * (N + 3 + pad) * 4);
  std::vector<float> h(na > nb ? na : nb);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u >> 8) & 0xffff) / 65536.f - 0.5f;
  hipMemcpy(a, h.data(), na * 4, hipMemcpyHostToDevice); hipMemcpy(b, h.data(), nb * 4, hipMemcpyHostToDevice);
  tfk::GemmArgs g;
  g.A = a; g.B = b; g.C = c; g.bias = nullptr; g.stats = nullptr; g.M = M; g.N = N; g.K = K; g.epi = 0;
  g.lda = (((layout == 2 ? M : K) + 3) & ~3) + pad; g.ldb = (((layout == 1 ? K : N) + 3) & ~3) + pad; g.ldc = ((N + 3) & ~3) + pad;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    for (int i = 0; i < 10; ++i) tfk::gemm_f32((tfk::GemmLayout)layout, g, cfg, 0);
    hipEventRecord(e0, 0);
    for (int i = 0; i < iters; ++i) { int rc = tfk::gemm_f32((tfk::GemmLayout)layout, g, cfg, 0); if (rc) { printf("rc %d\n", rc); return 1; } }
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= iters;
    if (rep == 2) printf("pad=%3d ABL=%2d layout %d %dx%dx%d cfg %d: %7.1f us  %6.1f TF\n", pad, TFK_ABL, layout, M, N, K, cfg, ms * 1e3, 2.0 * M * N * K / ms / 1e9);
  }
  return 0;
}
