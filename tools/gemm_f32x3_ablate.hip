// GPU tool: where does the time of the fp32-emulating GEMM (gemm_bf16x3) go?  Compiles tfkaldi_amd/csrc/gemm_bf16.hip with pieces
// of the K loop removed (-DTFKB_ABL: 1 no MFMAs, 2 no LDS-DMA pieces, 4 no fragment reads; sums combine) and times a shape.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DTFKB_ABL=<n> -I tfkaldi_amd/csrc tools/gemm_f32x3_ablate.hip -o tools/bin/x3abl<n>
//   tools/bin/x3abl<n> <layout> <M> <N> <K>         (block geometry: env TFK_BF16X3_CFG)
#include "../tfkaldi_amd/csrc/gemm_bf16.hip"

#include <stdio.h>
#include <vector>

int main(int argc, char** argv) {
  if (argc < 5) return 1;
  const int layout = atoi(argv[1]), M = atoi(argv[2]), N = atoi(argv[3]), K = atoi(argv[4]);
  auto p8 = [](int n) { return (n + 7) & ~7; };
  const int a_rows = layout == 2 ? K : M, a_cols = layout == 2 ? M : K;
  const int b_rows = layout == 1 ? N : K, b_cols = layout == 1 ? K : N;
  const int lda = p8(a_cols), ldb = p8(b_cols), ldc = (N + 3) & ~3;
  const long pa = ((long)a_rows * lda + 127) & ~127L, pb = ((long)b_rows * ldb + 127) & ~127L;
  std::vector<uint16_t> ha((size_t)3 * pa), hb((size_t)3 * pb);
  unsigned s = 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (uint16_t)(0x3c00u + ((s >> 9) & 0x3ffu) + ((s >> 8) & 0x8000u)); };
  for (auto& x : ha) x = rnd();
  for (auto& x : hb) x = rnd();
  uint16_t *dA, *dB;
  float* dC;
  hipMalloc(&dA, ha.size() * 2); hipMalloc(&dB, hb.size() * 2); hipMalloc(&dC, (size_t)M * ldc * 4);
  hipMemcpy(dA, ha.data(), ha.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(dB, hb.data(), hb.size() * 2, hipMemcpyHostToDevice);
  tfk::GemmArgsB g = {};
  g.A = dA; g.B = dB; g.C = dC; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  g.a_plane = pa; g.b_plane = pb;
  // the split-K form where the shape is eligible (TFK_BF16X3_CFG=0 / 1: never)
  if (const size_t need = tfk::gemm_bf16x3_splitk_floats((tfk::GemmLayout)layout, M, N, K)) {
    hipMalloc(&g.splitk_ws, need * 4);
    hipMemset(g.splitk_ws, 0, need * 4);
    g.splitk_ws_floats = need;
  }
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 5; ++i) tfk::gemm_bf16x3((tfk::GemmLayout)layout, g, 0);
  const int iters = 30;
  hipEventRecord(e0, 0);
  for (int i = 0; i < iters; ++i) tfk::gemm_bf16x3((tfk::GemmLayout)layout, g, 0);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= iters;
  printf("ABL=%d %s layout %d %dx%dx%d: %7.1f us  %7.1f TF fp32-equivalent\n", TFKB_ABL, g.splitk_ws ? "split-K" : "       ", layout, M, N, K, ms * 1e3, 2.0 * M * N * K / ms / 1e9);
  return 0;
}
