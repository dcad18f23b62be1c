// GPU tool: time the bf16 MFMA GEMM from a plain HIP host program; -DTFK_ABLB=<mask> times ablated K loops.
#include "../tfkaldi_amd/csrc/gemm_bf16.hip"
#include <stdio.h>
#include <vector>
int main(int argc, char** argv) {
  int layout = argc > 1 ? atoi(argv[1]) : 0, M = argc > 2 ? atoi(argv[2]) : 1024, N = argc > 3 ? atoi(argv[3]) : 2048,
      K = argc > 4 ? atoi(argv[4]) : 2048, iters = argc > 5 ? atoi(argv[5]) : 100;
  const int lda = ((layout == 2 ? M : K) + 7) & ~7, ldb = ((layout == 1 ? K : N) + 7) & ~7, ldc = (N + 3) & ~3;
  size_t na = (size_t)(layout == 2 ? K : M) * lda, nb = (size_t)(layout == 1 ? N : K) * ldb;
  uint16_t *a, *b; float* c;
  hipMalloc(&a, na * 2); hipMalloc(&b, nb * 2); hipMalloc(&c, (size_t)M * ldc * 4);
  std::vector<uint16_t> h(na > nb ? na : nb);
  for (size_t i = 0; i < h.size(); ++i) h[i] = 0x3c00 + (uint16_t)((i * 2654435761u >> 9) & 0x1ff);
  hipMemcpy(a, h.data(), na * 2, hipMemcpyHostToDevice); hipMemcpy(b, h.data(), nb * 2, hipMemcpyHostToDevice);
  tfk::GemmArgsB g = {};
  g.A = a; g.B = b; g.C = c; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    for (int i = 0; i < 10; ++i) tfk::gemm_bf16((tfk::GemmLayout)layout, g, 0);
    hipEventRecord(e0, 0);
    for (int i = 0; i < iters; ++i) { int rc = tfk::gemm_bf16((tfk::GemmLayout)layout, g, 0); if (rc) { printf("rc %d\n", rc); return 1; } }
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= iters;
    if (rep == 2) printf("ABLB=%2d layout %d %dx%dx%d: %7.1f us  %6.1f TF\n", TFK_ABLB, layout, M, N, K, ms * 1e3, 2.0 * M * N * K / ms / 1e9);
  }
  return 0;
}
